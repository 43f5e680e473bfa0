#!/bin/bash
# Gridify session: gridify_session.sh <outdir> [notests]
#   full -m gpu suite, device time of every Gridify layer of every config (small-cloud build on / off),
#   batch curve of layer 0, per-workgroup phase timelines
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-gridify}
mkdir -p $OUT
if [ -z "$2" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/gpu_tests.log
fi
timeout 600 python tools/time_gridify.py --curve > $OUT/gridify_times.txt 2> $OUT/gridify_times.err; cat $OUT/gridify_times.txt; tail -3 $OUT/gridify_times.err
timeout 300 python tools/prof_phases.py --cfg seg80k > $OUT/phases_seg80k.txt 2>/dev/null; cat $OUT/phases_seg80k.txt
