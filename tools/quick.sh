#!/bin/bash
# quick GPU check: quick.sh <outdir> "<pytest -k expr>" [bench args]
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-quick}; K="$2"; shift 2
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x --timeout 600 -k "$K" > $OUT/tests.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/tests.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --micro-iters 20 "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','ms_per_cagq_layer') if k in d})
for k in ('roofline','roofline_mfma'):
    if k in d: print(k, {x:d[k].get(x) for x in ('frac','frac_micro','ms_per_launch','ms_in_step')})
PY
