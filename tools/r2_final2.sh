#!/bin/bash
# final session of the round: full GPU suite, every bench line (fp32 + bf16), kernel stats and ordered trace of cfg4
# (SKIP_PMC=1: the Gridify kernels did not change since the last PMC passes)
D=$1
bash tools/r2_tests.sh $D
bash tools/r2_bench_all.sh $D
for cfg in cfg4 cfg2 cfg5; do
  st=30; [ $cfg = cfg5 ] && st=10
  timeout 600 python bench.py --config $cfg --steps $st --warmup 5 --dtype bf16 --no-cpu-baseline > gpurun_out/$D/bench_${cfg}_bf16.json 2> gpurun_out/$D/bench_${cfg}_bf16.err
  python -c "
import json
d=json.loads(open('gpurun_out/$D/bench_${cfg}_bf16.json').read().strip().splitlines()[-1]); print('$cfg bf16', d['value'], d['ms_per_step'])"
done
bash tools/r2_prof_cfg.sh $D/k4 cfg4 9 > /dev/null 2>&1; head -3 gpurun_out/$D/k4/kstats_cfg4.txt
bash tools/r2_prof_cfg.sh $D/k2 cfg2 6 > /dev/null 2>&1; head -3 gpurun_out/$D/k2/kstats_cfg2.txt
bash tools/r2_prof_cfg.sh $D/k5 cfg5 4 > /dev/null 2>&1; head -3 gpurun_out/$D/k5/kstats_cfg5.txt
bash tools/r2_steptrace.sh $D/t4 cfg4 40 > /dev/null 2>&1; tail -2 gpurun_out/$D/t4/steptrace_cfg4.txt
