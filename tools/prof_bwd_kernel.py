"""rocprof helper: run the backward GEMM kernel(s) of one layer shape a few times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grid_gcn_amd.train import timers as ttimers
ncent, P, cin, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
print(ttimers.time_linear_bwd(ncent, P, cin, C, iters=3), "ms")
