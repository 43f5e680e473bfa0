#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_gridconv_golden.py tests/test_gpu_gridconv.py -q -k "gridify_up_variant" > $OUT/golden.log 2>&1
echo "rc=$?"; tail -40 $OUT/golden.log
