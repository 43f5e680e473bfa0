#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/dbg_bf16.py 2>&1 | grep -v amdgpu | tail -4
for dt in f32 bf16; do timeout 300 python bench.py --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','dtype')})"; done
