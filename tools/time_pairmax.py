"""gridgcn_pairmax_fwd on the down-layer shapes, neighbour split forced to 1 / 2 / 4 / 8 lanes (GRIDGCN_OPT_PAIRMAX_SPLIT).
usage: python tools/time_pairmax.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import _lib  # noqa: E402
from grid_gcn_amd.train import timers as ttimers

lib = _lib.load()
dev = "cuda:0"
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for ncent, P, C in [(8192, 128, 64), (2048, 32, 128), (192, 32, 256), (16384, 64, 64), (131072, 64, 64)]:
    E = ncent * P
    Zp, Za = torch.randn(E, C, device=dev), torch.randn(E, C, device=dev)
    sc = [torch.rand(C, device=dev) + 0.5 for _ in range(2)]
    sh = [torch.randn(C, device=dev) * 0.1 for _ in range(2)]
    agg = torch.empty(ncent, C, device=dev)
    amax = torch.empty(ncent, C, dtype=torch.uint8, device=dev)
    zsel = torch.empty(2, ncent, C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    row = []
    for ps in (0, 1, 2, 4, 8):
        if ps and P < 8 * (ps // 2 or 1) and ps > 1:
            pass
        assert lib.gridgcn_set_option(_lib.OPT_PAIRMAX_SPLIT, ps) == 0
        call = lambda: lib.gridgcn_pairmax_fwd(p(Zp), p(Za), p(sc[0]), p(sh[0]), p(sc[1]), p(sh[1]), ncent, P, C,  # noqa: E731
                                               p(agg), C, p(amax), p(zsel), st)
        assert call() == 0
        row.append("%s:%7.1f us" % (ps or "auto", ttimers.median_ms(call, 30, device=dev) * 1e3))
    lib.gridgcn_set_option(_lib.OPT_PAIRMAX_SPLIT, 0)
    print("ncent %7d P %3d C %3d (%.0f MB)  " % (ncent, P, C, 2 * E * C * 4 / 1e6) + "  ".join(row))
