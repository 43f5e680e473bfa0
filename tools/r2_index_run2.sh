#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r2e
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/parity.log 2>&1
echo "parity rc=$?"; tail -3 $OUT/parity.log
for cfg in seg80k synth200k seg8k; do
  echo "== $cfg new"; timeout 300 python tools/prof_index.py --cfg $cfg --iters 50 2>&1 | grep gridify | tee $OUT/time_${cfg}_new.log
  echo "== $cfg phases"; timeout 300 python tools/prof_phases.py --cfg $cfg 2>&1 | grep -v amdgpu.ids | tee $OUT/phases_${cfg}.log
done
