#!/bin/bash
# The evidence of a round in one GPU session: final.sh <outdir> <label> [skip-tests]
#   full -m gpu suite (with the float-parity report), whole-step PMC (profiles/traffic.json), per-kernel counter
#   tables, eager step trace, bench lines of every config (fp32 + the bf16 lines), kernel stats of the bench
#   command for cfg4 / cfg2 / cfg5, Gridify times.  Everything lands in gpurun_out/<outdir>/ under the names it
#   has in profiles/ (<label>_...): copy what is to be judged.
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-final}; L=${2:-rX}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
if [ -z "$3" ]; then
  rm -f $OUT/${L}_float_parity.txt
  GG_PARITY_REPORT=$R/$OUT/${L}_float_parity.txt timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $OUT/gpu_tests.log 2>&1
  echo "pytest rc=$?"; tail -3 $OUT/gpu_tests.log
fi
for cfg in cfg4 cfg1 cfg2 cfg3 cfg3up cfg5; do
  st=50; [ $cfg = cfg5 ] && st=10
  timeout 900 python bench.py --config $cfg --steps $st --warmup 5 > $OUT/${L}_bench_$cfg.json 2> $OUT/bench_$cfg.err
  echo "== $cfg rc=$?"; python -c "
import json
d=json.loads(open('$OUT/${L}_bench_$cfg.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','ms_per_cagq_layer') if k in d}, {k:d[k].get('frac') for k in d if k.startswith('roofline')})"
done
for cfg in cfg4 cfg3 cfg2 cfg5; do
  st=50; [ $cfg = cfg5 ] && st=10
  timeout 900 python bench.py --config $cfg --dtype bf16 --steps $st --warmup 5 --no-cpu-baseline > $OUT/${L}_bench_${cfg}_bf16.json 2> $OUT/bench_${cfg}_bf16.err
  echo "== $cfg bf16 rc=$?"; python -c "
import json
d=json.loads(open('$OUT/${L}_bench_${cfg}_bf16.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','dtype') if k in d})"
done
bash tools/steptrace.sh $1/trace cfg4 150 --no-micro > /dev/null 2>&1; cp $OUT/trace/steptrace_cfg4.txt $OUT/${L}_step_trace_cfg4.txt
bash tools/pmc_step.sh $1/pmc > $OUT/pmc_step.log 2>&1; tail -3 $OUT/pmc_step.log
cp $OUT/pmc/pmc_step.txt $OUT/${L}_pmc_step.txt; cp $OUT/pmc/traffic.json $OUT/traffic.json
bash tools/pmc_step_kernels.sh $1/pmck cfg4 > /dev/null 2>&1; cp $OUT/pmck/kernels_cfg4.txt $OUT/${L}_pmc_step_kernels_cfg4.txt
for cfg in cfg4 cfg2 cfg5; do
  bash tools/bench_kstats.sh $1/kst $cfg $L > /dev/null 2>&1; cp $OUT/kst/${L}_bench_kernel_stats_$cfg.txt $OUT/
done
timeout 600 python tools/time_gridify.py > $OUT/${L}_gridify_times.txt 2> $OUT/gridify_times.err; tail -12 $OUT/${L}_gridify_times.txt
# round 5: the bf16 step's own PMC pass (traffic.json: step_cfg4_bf16), the bf16 cfg4 line again with it, Gridify
# instruction counters, A/B of the two round-5 kernels
bash tools/pmc_step.sh $1/pmc bf16 > $OUT/pmc_step_bf16.log 2>&1; cp $OUT/pmc/pmc_step_bf16.txt $OUT/${L}_pmc_step_bf16.txt; cp $OUT/pmc/traffic.json $OUT/traffic.json
timeout 900 python bench.py --config cfg4 --dtype bf16 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${L}_bench_cfg4_bf16.json 2> $OUT/bench_cfg4_bf16b.err
bash tools/gridify_insts.sh $1/ginsts > /dev/null 2>&1; cp $OUT/ginsts/gridify_insts.txt $OUT/${L}_gridify_insts.txt
{ timeout 300 python tools/time_nz.py; timeout 300 python tools/time_bwdfused.py; timeout 300 python tools/time_attfwd.py; } > $OUT/${L}_ab_kernels.txt 2>&1; tail -12 $OUT/${L}_ab_kernels.txt
# the Z2-free attention forward, in the step (OPT.NOZ_ATT_FWD)
bash tools/ab_lib.sh $1/abfwd NOZ_ATT_FWD 2 > $OUT/${L}_ab_noz_att_fwd.txt 2>&1; cat $OUT/${L}_ab_noz_att_fwd.txt
timeout 300 python tools/graph_branch_probe.py > $OUT/${L}_graph_branch_probe.txt 2>&1; tail -2 $OUT/${L}_graph_branch_probe.txt
