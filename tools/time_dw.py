"""Backward of the wide per-point layers (655 360 rows; 256->128 and 128->128) a few times, for a
rocprofv3 kernel trace:   rocprofv3 --kernel-trace --stats -d out -- python tools/time_dw.py [--dense]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd.train import timers as ttimers  # noqa: E402
dense = "--dense" in sys.argv
for cin, C in ((256, 128), (128, 128), (32, 128), (64, 64)):
    ms = ttimers.time_linear_bwd(131072, 5, cin, C, iters=5, device="cuda:0", prev_bn=True, dense=dense)
    print("bwd %d->%d%s: %.3f ms" % (cin, C, " dense" if dense else "", ms))
