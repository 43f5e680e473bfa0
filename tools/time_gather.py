"""batch_take_g (the reference's neighbour gather, utils/ops.py:78-93) forward and backward at the
cfg4 up2 shape: GB/s against SURVEY §8(d)'s algorithmic bytes.  usage: python tools/time_gather.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import ops  # noqa: E402

dev = "cuda:0"


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, B, N, C, O, P in (("cfg4 up2", 8, 1024, 128, 81920, 5), ("cfg4 down0", 8, 81920, 0, 1024, 128),
                            ("cfg4 down1", 8, 1024, 64, 256, 32)):
    g = torch.Generator(device=dev).manual_seed(0)
    data = torch.randn(B, N, 4 + C, device=dev, generator=g)
    idx = torch.randint(0, N, (B, O, P), device=dev, dtype=torch.int32, generator=g)
    alg = 4.0 * (4 + C) * N * B + 4.0 * O * P * B + 4.0 * (4 + C) * O * P * B
    ms_f = timed(lambda: ops.batch_take_g(data, idx))
    gout = torch.randn(B, O, P, 4 + C, device=dev, generator=g)
    ms_b = timed(lambda: ops.batch_take_g_backward(gout, idx, N, True))
    print("%-10s fwd %.3f ms = %6.0f GB/s (%4.1f %% of 8 TB/s) | bwd %.3f ms = %6.0f GB/s   alg %.1f MB" % (
        name, ms_f, alg / ms_f / 1e6, alg / ms_f / 1e6 / 80, ms_b, alg / ms_b / 1e6, alg / 1e6))
