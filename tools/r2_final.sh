#!/bin/bash
# full GPU suite, then every bench line + PMC traffic, CAS timing
bash tools/r2_tests.sh $1
python tools/time_cas.py 2>&1 | grep -v amdgpu.ids > gpurun_out/$1/cas_timing.txt; cat gpurun_out/$1/cas_timing.txt
bash tools/r2_bench_all.sh $1
