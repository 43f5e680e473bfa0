#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gridconv.py tests/test_gpu_train_ops.py -x -q > $OUT/t.log 2>&1; echo "rc=$?"; tail -5 $OUT/t.log
for cfg in cfg4 cfg3 cfg2; do
  timeout 600 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; echo "== $cfg rc=$?"; grep -v amdgpu.ids $OUT/bench_$cfg.err | tail -3
  python -c "
import json
d=json.loads(open('$OUT/bench_$cfg.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step','step_mode') if k in d})"
done
GG_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --batch 4 > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "== N=2 gloo rc=$?"; tail -3 $OUT/bench_n2.err | grep -v amdgpu; python -c "
import json
d=json.loads(open('$OUT/bench_n2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus','step_mode') if k in d})"
