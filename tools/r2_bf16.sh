#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_train_ops.py -x -q -k "bf16 or mlp" > $OUT/t.log 2>&1; echo "rc=$?"; tail -6 $OUT/t.log
for cfg in cfg4; do
 for dt in f32 bf16; do
  st=30; [ $cfg = cfg5 ] && st=10
  timeout 600 python bench.py --config $cfg --dtype $dt --steps $st --warmup 5 --no-cpu-baseline > $OUT/bench_${cfg}_$dt.json 2> $OUT/bench_${cfg}_$dt.err; echo "== $cfg $dt rc=$?"; grep -v amdgpu.ids $OUT/bench_${cfg}_$dt.err | tail -2
  python -c "
import json
d=json.loads(open('$OUT/bench_${cfg}_$dt.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype','step_mode') if k in d}, d.get('roofline_step',{}).get('frac'))"
 done
done
