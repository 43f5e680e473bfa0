#!/bin/bash
# full GPU suite + default bench line; outputs under gpurun_out/$1
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r3_tests}
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/gpu_tests.log
timeout 900 python bench.py > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('$OUT/bench_cfg4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode') if k in d}, 'cagq', d.get('ms_per_cagq_layer'), 'roofline', d['roofline']['frac'], 'step', d.get('roofline_step',{}).get('frac'))
PY
