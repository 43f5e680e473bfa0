#!/bin/bash
# kernel stats of an eager bench run: r2_prof_cfg.sh <outdir> <cfg> <steps> [extra bench args]
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1; CFG=$2; ST=$3; shift 3
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o p -- python $R/bench.py --config $CFG --steps $ST --warmup 0 --eager --no-cpu-baseline "$@" > $R/$OUT/bench.log 2>&1
cd $R
python tools/kstats.py $OUT/trace $ST > $OUT/kstats_$CFG.txt
head -45 $OUT/kstats_$CFG.txt
rm -rf $OUT/trace
