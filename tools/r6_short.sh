#!/bin/bash
# a late first GPU session (the pool opening near the end of a session): the driver's bench command, then the GPU tier
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6_short; mkdir -p $OUT
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg4_driver_cmd.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench_cfg4_driver_cmd.json
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -x > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_tests.log
