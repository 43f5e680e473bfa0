#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 600 python tools/graph_step_probe.py cfg4 2>&1 | grep -v amdgpu.ids | tee $OUT/graph_cfg4.log
timeout 600 python tools/graph_step_probe.py cfg3 2>&1 | grep -v amdgpu.ids | tee $OUT/graph_cfg3.log
