#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "rc=$?"; tail -12 $OUT/gpu_tests.log
