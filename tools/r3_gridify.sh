#!/bin/bash
# Gridify: parity tests, device time per call (all layers), per-workgroup phase timelines
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r3_gridify}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_cas.py tests/test_fast_rand.py -q -m gpu -x -k "not gridconv and not train" > $OUT/parity.log 2>&1; echo "parity rc=$?"; tail -3 $OUT/parity.log
for c in seg80k synth200k seg8k cls; do
  timeout 300 python tools/prof_index.py --cfg $c --iters 50 2>/dev/null | grep gridify | grep -v up > $OUT/index_$c.txt; cat $OUT/index_$c.txt
done
timeout 300 python tools/prof_phases.py --cfg seg80k > $OUT/phases_seg80k.txt 2>/dev/null; cat $OUT/phases_seg80k.txt
timeout 300 python tools/prof_phases.py --cfg synth200k > $OUT/phases_synth200k.txt 2>/dev/null; cat $OUT/phases_synth200k.txt
