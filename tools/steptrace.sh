#!/bin/bash
# ordered kernel list of one eager training step: steptrace.sh <outdir> <cfg> [min_us] [extra bench args]
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1; CFG=$2; MINUS=${3:-40}; shift 3
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o p -- python $R/bench.py --config $CFG --steps 4 --warmup 2 --eager --no-cpu-baseline "$@" > $R/$OUT/bench.log 2>&1
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/step_trace.py $F $MINUS 70 narrow > $OUT/steptrace_$CFG.txt
cat $OUT/steptrace_$CFG.txt
rm -rf $OUT/trace
