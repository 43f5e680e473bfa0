#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/dbg_bf16.py 2>&1 | grep -v amdgpu
