#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
run() { echo "-- $*"; env "$@" timeout 120 python tools/prof_index.py --cfg $CFG --iters 50 2>&1 | grep "gridify L0"; }
for CFG in seg80k synth200k; do
  echo "==== $CFG"
  run X=0
  run GG_TUNE_QNC=1
  run GG_TUNE_QNC=2
  run GG_TUNE_QNC=4
  run GG_TUNE_KB=-1
  run GG_TUNE_KB=1
  run GG_TUNE_CH=1024
  run GG_TUNE_CH=2048
  run GG_TUNE_CH=4096
done 2>&1 | tee $OUT/tune.log
