"""Device time of one Gridify call (gridgcn_gridify_timed: back-to-back calls between two HIP events inside
the library) for every down layer of every BASELINE config, the small-cloud build on and off, and the
batch curve of layer 0 (what fraction of the HBM peak the four kernels reach once the launch latency of a
tiny batch is amortised: VERDICT r3 item 4e).

    python tools/time_gridify.py [--iters 200] [--curve]   ->  table on stdout
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import _lib, ops, synth  # noqa: E402

DEV = "cuda:0"
PEAK = 8.0e12


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def layers(cfg, B, kind):
    """(layer, data, np, kwargs) with layer l fed by layer l-1's centres, as the models chain them"""
    data, npn = synth.make_batch(B, cfg["num_points"], kind)
    d, n = T(data), T(npn)
    out = []
    for l in range(len(cfg["down"])):
        kw = synth.gridify_kwargs(cfg, l)
        out.append((l, d, n, kw))
        r = ops.Gridify(d, n, **kw)
        d, n = r[2], r[4]
    return out


def timed(d, n, kw, iters):
    ms, _ = ops.gridify_timed(d, n, iters, **kw)
    return ms * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--curve", action="store_true")
    a = ap.parse_args()
    lib = _lib.load()
    print("Gridify device time per call (us), %d back-to-back calls; alg. bytes = SURVEY 8(d): "
          "B * (16 N + 8 O P + 20 O + 8)" % a.iters)
    print("%-22s %3s %7s %6s %4s %5s %9s %9s %8s %7s" % ("config / layer", "B", "N", "O", "P", "k", "us",
                                                         "us(split)", "GB/s", "of 8TB/s"))
    for name, cfg, B, kind in (("cfg4 seg81920", synth.SEG_SCANNET_81920, 8, "planes"),
                               ("cfg3 seg8192", synth.SEG_SCANNET_8192, 16, "planes"),
                               ("cfg2 cls1024", synth.CLS_MODELNET40, 32, "ball"),
                               ("cfg1 cls1024 b1", synth.CLS_MODELNET40, 1, "ball"),
                               ("cfg5 synth200k", synth.SYNTH_200K, 8, "planes")):
        for l, d, n, kw in layers(cfg, B, kind):
            N = d.shape[1]
            us = timed(d, n, kw, a.iters)
            us0 = float("nan")
            if N <= 4096 and kw["max_o_grid"] <= 4096:
                _lib.check(lib.gridgcn_set_option(_lib.OPT_INDEX_SMALL, 0), "set_option")
                try:
                    us0 = timed(d, n, kw, a.iters)
                finally:
                    _lib.check(lib.gridgcn_set_option(_lib.OPT_INDEX_SMALL, 1), "set_option")
            by = B * synth.gridify_algorithmic_bytes(N, kw["max_o_grid"], kw["max_p_grid"])
            print("%-22s %3d %7d %6d %4d %5d %9.2f %9.2f %8.1f %7.4f" % (
                "%s L%d" % (name, l), B, N, kw["max_o_grid"], kw["max_p_grid"], kw["kernel_size"], us, us0,
                by / us / 1e3, by / (us * 1e-6) / PEAK))
    if a.curve:
        print("\nbatch curve, layer 0 (same clouds repeated in blocks of 8):")
        print("%-22s %4s %9s %9s %8s %8s" % ("config", "B", "us", "us/cloud", "GB/s", "of 8TB/s"))
        for name, cfg, Bs in (("cfg4 seg81920 L0", synth.SEG_SCANNET_81920, (1, 2, 4, 8, 16, 32, 64, 128)),
                              ("cfg5 synth200k L0", synth.SYNTH_200K, (1, 2, 4, 8, 16, 32))):
            base, bn = synth.make_batch(8, cfg["num_points"], "planes")
            kw = synth.gridify_kwargs(cfg, 0)
            for B in Bs:
                reps = (B + 7) // 8
                d = T(np.concatenate([base] * reps)[:B])
                n = T(np.concatenate([bn] * reps)[:B])
                us = timed(d, n, kw, max(20, a.iters // max(1, B // 8)))
                by = B * synth.gridify_algorithmic_bytes(cfg["num_points"], kw["max_o_grid"], kw["max_p_grid"])
                print("%-22s %4d %9.2f %9.2f %8.1f %8.4f" % (name, B, us, us / B, by / us / 1e3,
                                                             by / (us * 1e-6) / PEAK))
                del d, n
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
