"""Backward of the 32 -> 128 attention conv of layer up2 (3276800 edges): the one-pass kernel
(gg_k_att_bwd_fused) against the separate dX / dW kernels (GRIDGCN_OPT_ATT_BWD_FUSED = 0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import _lib  # noqa: E402
from grid_gcn_amd.train import timers as ttimers

lib = _lib.load()
for fused in (1, 0):
    _lib.check(lib.gridgcn_set_option(_lib.OPT_ATT_BWD_FUSED, fused), "gridgcn_set_option")
    for prev in (True, False):
        ms = ttimers.time_linear_bwd(8 * 81920, 5, 32, 128, iters=20, ndx=32, prev_bn=prev)
        print("fused=%d prev_bn=%s: %.3f ms per call" % (fused, prev, ms))
lib.gridgcn_set_option(_lib.OPT_ATT_BWD_FUSED, 1)
