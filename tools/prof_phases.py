"""Per-workgroup phase timeline of one Gridify call (index build + query kernels).

Needs the -DGG_PROF variant of the library (python -m grid_gcn_amd.build --prof): thread 0 of every
workgroup stamps the 100 MHz wall clock at the phase boundaries of its kernel into a buffer
(gridgcn_dev.h GG_STAMP).  Prints, per kernel: span of the whole kernel (first start -> last end),
and median / p95 / max duration of each phase over the workgroups.

    python tools/prof_phases.py --cfg seg80k|synth200k|seg8k|cls [--B 8] [--layer 0]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GG_HIP_LIB"] = os.path.join(ROOT, "grid_gcn_amd", "lib", "libgridgcn_hip_prof.so")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from grid_gcn_amd import _lib, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="seg80k")
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--layer", type=int, default=0)
a = ap.parse_args()
cfg = {"seg80k": synth.SEG_SCANNET_81920, "seg8k": synth.SEG_SCANNET_8192,
       "cls": synth.CLS_MODELNET40, "synth200k": synth.SYNTH_200K}[a.cfg]
dev = "cuda:0"
data, npn = synth.make_batch(a.B, cfg["num_points"], "planes" if a.cfg != "cls" else "ball")
d, n = torch.from_numpy(data).to(dev), torch.from_numpy(npn).to(dev)
for l in range(a.layer):
    out = ops.Gridify(d, n, **synth.gridify_kwargs(cfg, l))
    d, n = out[2], out[4]
kw = synth.gridify_kwargs(cfg, a.layer)
lib = _lib.load()
NK, NWG, NS = 4, 4096, 16
buf = torch.zeros(NK * NWG * NS, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.Gridify(d, n, **kw)
torch.cuda.synchronize()
for name in ("gridgcn_prof_set_index", "gridgcn_prof_set_query"):
    f = getattr(lib, name)
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p]
    assert f(ctypes.c_void_p(buf.data_ptr())) == 0
ops.Gridify(d, n, **kw)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NK, NWG, NS).astype(np.float64) * 0.01  # us
names = ["chunk_split", "slab_build", "centre_slots", "query"]
t0 = None
for k in range(NK):
    used = t[k][:, 0] > 0
    if not used.any():
        continue
    tk = t[k][used].copy()
    # slots 14 / 15: shader clock (s_memtime) at the first / latest stamp of the workgroup
    sclk = (tk[:, 15] - tk[:, 14]) / 0.01           # undo the 100 MHz scaling: raw cycles
    tk[:, 14:] = 0
    nst = int((tk > 0).all(axis=0).sum())
    start, end = tk[:, 0].min(), tk[:, :nst].max()
    if t0 is None:
        t0 = start
    print("%-13s WGs(stamped) %5d  kernel span %.2f us  [starts at +%.2f us, WG starts spread %.2f us]" % (
        names[k], used.sum(), end - start, start - t0, tk[:, 0].max() - start))
    for s in range(1, nst):
        dur = tk[:, s] - tk[:, s - 1]
        print("    phase %d->%d: median %6.2f  p95 %6.2f  max %6.2f us" % (
            s - 1, s, np.median(dur), np.percentile(dur, 95), dur.max()))
    tot = tk[:, nst - 1] - tk[:, 0]
    print("    WG total : median %6.2f  p95 %6.2f  max %6.2f us" % (
        np.median(tot), np.percentile(tot, 95), tot.max()))
    last = np.array([row[:14][row[:14] > 0].max() for row in tk])
    mhz = sclk / np.maximum(last - tk[:, 0], 1e-3)
    print("    shader clock over the workgroups' lives: median %.0f MHz (min %.0f, max %.0f)" % (
        np.median(mhz), mhz.min(), mhz.max()))
    if k == 1 and nst < NS and (tk[:, nst + 1] > 0).any():
        # centre pass: only the last slab workgroup of each cloud stamps nst (ticket drawn) and nst+1
        cp = tk[tk[:, nst + 1] > 0]
        d = cp[:, nst + 1] - cp[:, nst]
        print("    centre pass (%d workgroups): median %.2f  max %.2f us; starts %.2f .. %.2f us after the "
              "kernel's start, kernel ends at +%.2f us" % (len(cp), np.median(d), d.max(),
                                                           cp[:, nst].min() - start, cp[:, nst].max() - start,
                                                           cp[:, nst + 1].max() - start))
