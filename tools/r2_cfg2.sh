#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 600 python bench.py --config cfg2 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
tail -3 $OUT/bench_cfg2.err | grep -v amdgpu.ids
python -c "
import json
d=json.loads(open('$OUT/bench_cfg2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','ms_per_step','host_enqueue_ms_per_step') if k in d})
"
bash tools/r2_prof_cfg.sh $1 cfg2 5
