#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/step -o p -- python $R/bench.py --config cfg2 --eager --steps 6 --warmup 3 > $R/$OUT/bench.log 2>&1
cd $R
python - <<'PY'
import csv,glob,collections,sys
f=glob.glob('gpurun_out/%s/step/**/*kernel_trace.csv'%sys.argv[1] if len(sys.argv)>1 else '',recursive=True)
PY
python tools/kstats.py $OUT/step 9 | head -45
