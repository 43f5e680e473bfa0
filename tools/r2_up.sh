#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gridconv.py -x -q -k "up or parity" > $OUT/parity.log 2>&1
echo "parity rc=$?"; tail -3 $OUT/parity.log
for cfg in seg80k seg8k; do
  B=8; [ $cfg = seg8k ] && B=16
  timeout 300 python tools/prof_index.py --cfg $cfg --B $B --iters 50 2>&1 | grep "gridify_up\|ball" 
  GG_UP_WAVE=1 timeout 300 python tools/prof_index.py --cfg $cfg --B $B --iters 50 2>&1 | grep "gridify_up" | sed 's/^/wave-per-point: /'
done
