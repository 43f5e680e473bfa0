#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -5 $OUT/gpu_tests.log
for cfg in seg80k synth200k; do
  timeout 300 python tools/prof_index.py --cfg $cfg --iters 50 2>&1 | grep gridify | tee $OUT/time_${cfg}.log
  timeout 300 python tools/prof_phases.py --cfg $cfg 2>&1 | grep -v amdgpu.ids > $OUT/phases_${cfg}.log
  grep -E "kernel span" $OUT/phases_${cfg}.log
done
