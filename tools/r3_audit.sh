#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1; CFG=$2; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o p -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --eager --no-cpu-baseline > $R/$OUT/bench.log 2>&1
cd $R
python tools/occupancy_audit.py $OUT/trace 45 > $OUT/audit_$CFG.txt; cat $OUT/audit_$CFG.txt
rm -rf $OUT/trace
