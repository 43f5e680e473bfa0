"""Time the index operators at BASELINE configs[3] (ScanNet 81920-pt, B=8) with HIP events.
Usage: python tools/prof_index.py [--cfg seg80k|seg8k|cls|synth200k] [--B 8] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="seg80k")
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
cfg = {"seg80k": synth.SEG_SCANNET_81920, "seg8k": synth.SEG_SCANNET_8192,
       "cls": synth.CLS_MODELNET40, "synth200k": synth.SYNTH_200K}[a.cfg]
dev = "cuda:0"
data, npn = synth.make_batch(a.B, cfg["num_points"], "planes" if a.cfg != "cls" else "ball")
d, n = torch.from_numpy(data).to(dev), torch.from_numpy(npn).to(dev)


def timeit(fn, iters):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


levels = [(d, n)]
for l in range(len(cfg["down"])):
    kw = synth.gridify_kwargs(cfg, l)
    # device time of the whole call: back-to-back launches inside the library between two HIP
    # events (a Python loop would be host-bound at these durations)
    ms, out = ops.gridify_timed(levels[-1][0], levels[-1][1], a.iters, **kw)
    N = levels[-1][0].shape[1]
    byts = a.B * synth.gridify_algorithmic_bytes(N, kw["max_o_grid"], kw["max_p_grid"])
    print("gridify L%d N=%d O=%d P=%d k=%d: %.4f ms  alg %.2f MB -> %.1f GB/s = %.1f%% of 8 TB/s (mean nn %.1f, centres %.0f)" % (
        l, N, kw["max_o_grid"], kw["max_p_grid"], kw["kernel_size"], ms, byts / 1e6,
        byts / ms / 1e6, byts / ms / 1e6 / 80.0, float(out[1].sum(-1)[out[3] > 0].mean()),
        float(out[4].float().mean())))
    levels.append((out[2], out[4]))
if "up" in cfg:
    for u in range(3):
        down, dn = levels[3 - u]
        up, un = levels[2 - u]
        kw = synth.gridify_up_kwargs(cfg, u)
        r = kw["voxel_size"][0] * kw["kernel_size"] * 1.7 / 2
        ux, dx = up[..., :3].contiguous(), down[..., :3].contiguous()
        ms, _ = timeit(lambda: ops.GridifyUp(down, up, dn, un, **kw), a.iters)
        print("gridify_up up%d M=%d Nd=%d: %.3f ms" % (u, up.shape[1], down.shape[1], ms))
        ms, _ = timeit(lambda: ops.BallKNN(ux, dx, dn, un, k=5, radius=r), a.iters)
        print("ball_knn   up%d M=%d Nd=%d: %.3f ms" % (u, up.shape[1], down.shape[1], ms))
