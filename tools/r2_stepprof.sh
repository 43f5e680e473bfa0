#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/step -o p -- python $R/bench.py --eager --steps 10 --warmup 3 --no-cpu-baseline > $R/$OUT/bench.log 2>&1
cd $R
python tools/step_stats.py $OUT/step 10 $OUT/step_stats.txt | head -70
