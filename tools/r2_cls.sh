#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_model_cls.py -x -q -m gpu 2>&1 | tail -30
