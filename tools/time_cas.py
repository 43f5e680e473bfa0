"""Device time of Gridify_occaware (CAS) next to Gridify (RVS) on synthetic batches; also the
coverage (occupied voxels inside at least one centre's window) of both samples."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = "--prof" in sys.argv      # the -DGG_PROF library (python -m grid_gcn_amd.build --prof): the kernel prints
if PROF:                         # where the time of cloud 0 goes
    os.environ["GG_HIP_LIB"] = os.path.join(ROOT, "grid_gcn_amd", "lib", "libgridgcn_hip_prof.so")
import numpy as np
import torch
from grid_gcn_amd import ops, synth

def timeit(fn, it=10, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

def coverage(data, cent, centnum, kw, b=0):
    g = torch.tensor(kw["grid_size"], device=data.device)
    vs = torch.tensor(kw["voxel_size"], device=data.device)
    sh = torch.tensor(kw["coord_shift"], device=data.device)
    v = torch.floor((data[b, :, :3] + sh) / vs).long()
    ok = ((v >= 0) & (v < g)).all(1)
    occ = torch.zeros(tuple(kw["grid_size"][::-1]), dtype=torch.bool, device=data.device)
    v = v[ok]
    occ[v[:, 2], v[:, 1], v[:, 0]] = True
    n = int(centnum[b])
    cv = torch.floor((cent[b, :n, :3] + sh) / vs).long()
    cov = torch.zeros_like(occ)
    r = (kw["kernel_size"] - 1) // 2
    for dz in range(-r, r + 1):
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                w = cv + torch.tensor([dx, dy, dz], device=data.device)
                m = ((w >= 0) & (w < g)).all(1)
                w = w[m]
                cov[w[:, 2], w[:, 1], w[:, 0]] = True
    return int((cov & occ).sum()), int(occ.sum())

timeit(lambda: torch.zeros(8, device="cuda"), it=3, warm=1)   # first event pair: ~40 ms of one-time runtime setup
for name, cfg, B, N, kind in (("cfg4 down0 (8 x 81920, 40^3, O=1024)", synth.SEG_SCANNET_81920, 8, 81920, "planes"),
                              ("cfg5 layer0 (8 x 200000, 64^3, O=16384)", synth.SYNTH_200K, 8, 200000, "planes")):
    data, npn = synth.make_batch(B, N, kind)
    d, n = torch.from_numpy(data).cuda(), torch.from_numpy(npn).cuda()
    kw = synth.gridify_kwargs(cfg, 0)
    t_rvs = timeit(lambda: ops.Gridify(d, n, **kw))
    t_cas = timeit(lambda: ops.Gridify_occaware(d, n, beta=1.0, **kw), it=1 if PROF else 5, warm=0 if PROF else 2)
    a = ops.Gridify(d, n, **kw)
    c = ops.Gridify_occaware(d, n, beta=1.0, **kw)
    torch.cuda.synchronize()
    ca, no = coverage(d, a[2], a[4], kw)
    cc, _ = coverage(d, c[2], c[4], kw)
    print("%s: Gridify (RVS) %.3f ms, Gridify_occaware (CAS, beta=1) %.3f ms; cloud 0: %d occupied voxels, "
          "covered by RVS centres %d, by CAS centres %d" % (name, t_rvs, t_cas, no, ca, cc))
