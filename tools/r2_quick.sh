#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/parity.log 2>&1
echo "parity rc=$?"; tail -2 $OUT/parity.log
for cfg in seg80k synth200k; do
  timeout 300 python tools/prof_index.py --cfg $cfg --iters 50 2>&1 | grep gridify | tee $OUT/time_${cfg}.log
  timeout 300 python tools/prof_phases.py --cfg $cfg 2>&1 | grep -v amdgpu.ids > $OUT/phases_${cfg}.log
  grep -E "kernel span" $OUT/phases_${cfg}.log
done
