"""A/B of the Z2-free attention backward's two tile loops at the cfg4 up2 shape (GRIDGCN_OPT_ATT_NZ_V2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import _lib
from grid_gcn_amd.train import timers as ttimers
lib = _lib.load()
for rep in range(2):
    for v2 in (1, 0):
        lib.gridgcn_set_option(6, v2)
        print("att_bwd_noz v2=%d  ncent 655360 P 5: %.4f ms" % (v2, ttimers.time_att_bwd_noz(655360, 5, 32, 128, iters=30)))
lib.gridgcn_set_option(6, 1)
