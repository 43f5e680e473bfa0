"""Device time of the fast_rand build of Gridify next to the default build (cfg4 layer 0 shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grid_gcn_amd import ops, synth

def timeit(fn, it=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

timeit(lambda: torch.zeros(8, device="cuda"), 3, 1)
for name, cfg, N in (("cfg4 layer 0 (8 x 81920, P 128, O 1024)", synth.SEG_SCANNET_81920, 81920),
                     ("cfg3 layer 0 (16 x 8192, P 64, O 1024)", synth.SEG_SCANNET_8192, 8192)):
    B = 8 if N > 8192 else 16
    data, npn = synth.make_batch(B, N, "planes")
    d, n = torch.from_numpy(data).cuda(), torch.from_numpy(npn).cuda()
    kw = synth.gridify_kwargs(cfg, 0)
    print("%s: Gridify %.3f ms, Gridify_fast_rand %.3f ms" % (
        name, timeit(lambda: ops.Gridify(d, n, **kw)), timeit(lambda: ops.Gridify_fast_rand(d, n, **kw))))
