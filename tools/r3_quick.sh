#!/bin/bash
# quick check of a change on the GPU box: r3_quick.sh <outdir> [pytest -k expression]
#   GPU tests (all, or the -k selection), one cfg4 bench line (no CPU baseline), eager step trace
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-quick}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
if [ -n "$2" ]; then
  timeout 900 python -m pytest tests -q -x -m gpu -k "$2" > $OUT/gpu_tests.log 2>&1
else
  timeout 1200 python -m pytest tests -q -x -m gpu > $OUT/gpu_tests.log 2>&1
fi
echo "pytest rc=$?"; tail -15 $OUT/gpu_tests.log
timeout 600 python bench.py --config cfg4 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
echo "bench rc=$?"; python -c "
import json
d=json.loads(open('$OUT/bench_cfg4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode') if k in d})"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o p -- python $R/bench.py --config cfg4 --steps 4 --warmup 2 --eager --no-cpu-baseline --no-micro > $R/$OUT/trace_bench.log 2>&1
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/step_trace.py $F 100 80 narrow > $OUT/steptrace_cfg4.txt
cat $OUT/steptrace_cfg4.txt
rm -rf $OUT/trace
