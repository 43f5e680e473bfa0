"""Backward of the 32 -> 128 attention conv of layer up2 (3276800 edges), fused kernel vs the
separate dX / dW kernels.  usage: [GG_NO_ATT_FUSED=1] python tools/time_att_bwd.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import train_ops  # noqa: E402

for prev in (True, False):
    ms = train_ops.time_linear_bwd(8 * 81920, 5, 32, 128, iters=20, ndx=32, prev_bn=prev)
    print("prev_bn=%s: %.3f ms per call (GG_NO_ATT_FUSED=%s)" % (prev, ms, os.environ.get("GG_NO_ATT_FUSED", "")))
