#!/bin/bash
# A/B of two builds on the same box: alternate runs of bench.py with the default library and with
# grid_gcn_amd/lib/libgridgcn_hip_B.so (GG_HIP_LIB)
CFG=${1:-cfg4}
for i in 1 2 3; do
  for v in A B; do
    if [ $v = B ]; then export GG_HIP_LIB=$PWD/grid_gcn_amd/lib/libgridgcn_hip_B.so; else unset GG_HIP_LIB; fi
    python bench.py --config $CFG --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3))"
  done
done
