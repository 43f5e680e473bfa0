"""A/B of the one-pass backward of a 128-output per-point layer (GRIDGCN_OPT_BWD_FUSED128) at the cfg4 head shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import _lib
from grid_gcn_amd.train import timers as ttimers
lib = _lib.load()
for cin in (128, 256):
    for rep in range(2):
        for v in (1, 0):
            lib.gridgcn_set_option(7, v)
            t = ttimers.time_linear_bwd(655360, 1, cin, 128, iters=20, prev_bn=True, dense=True)
            print("linear_bwd E 655360 %d -> 128 fused=%d: %.4f ms" % (cin, v, t))
lib.gridgcn_set_option(7, 1)
