#!/bin/bash
# time the fused attention backward at the up2 shape (3.28 M edges, 32 -> 128) and at cfg5's layer 0 (8.4 M edges, 16 -> 64)
python tools/time_att_bwd.py 2>&1 | grep prev_bn
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from grid_gcn_amd import train_ops
print("cfg5 L0 16->64: %.3f ms" % train_ops.time_linear_bwd(8 * 16384, 64, 16, 64, iters=10, ndx=16, prev_bn=True))
PY
