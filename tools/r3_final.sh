#!/bin/bash
# Round-3 evidence in one GPU session: r3_final.sh <outdir> [skip-tests]
#   full -m gpu suite, whole-step PMC (traffic.json), per-kernel counter table, backward-GEMM counters,
#   eager step trace, the MFMA micro-benchmarks, bench lines of every config (fp32 + the bf16 lines)
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r3_final}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
if [ -z "$2" ]; then
  timeout 2400 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_tests.log
fi
bash tools/r3_pmc_step.sh $1/pmc > $OUT/pmc_step.log 2>&1; tail -3 $OUT/pmc_step.log
cp $OUT/pmc/pmc_step.txt $OUT/r3_pmc_step.txt
bash tools/pmc_step_kernels.sh $1/pmck cfg4 > /dev/null 2>&1; cp $OUT/pmck/kernels_cfg4.txt $OUT/r3_pmc_step_kernels_cfg4.txt
bash tools/pmc_step_kernels.sh $1/pmckb cfg4 --dtype bf16 > /dev/null 2>&1; cp $OUT/pmckb/kernels_cfg4.txt $OUT/r3_pmc_step_kernels_cfg4_bf16.txt
bash tools/pmc_gemm.sh $1/pmcg --dense > /dev/null 2>&1; cp $OUT/pmcg/summary.txt $OUT/r3_pmc_bwd_gemm.txt
bash tools/steptrace.sh $1/trace cfg4 150 --no-micro > /dev/null 2>&1; cp $OUT/trace/steptrace_cfg4.txt $OUT/r3_step_trace_cfg4.txt
{
  for m in mfma_peak mfma_valu mfma_mem; do
    hipcc --offload-arch=gfx950 -O3 tools/micro/$m.hip -o /tmp/$m 2>/dev/null
    echo "== tools/micro/$m.hip"; timeout 100 /tmp/$m
  done
} > $OUT/r3_micro_mfma.txt 2>&1
for cfg in cfg4 cfg1 cfg2 cfg3 cfg3up cfg5; do
  st=50; [ $cfg = cfg5 ] && st=10
  timeout 900 python bench.py --config $cfg --steps $st --warmup 5 > $OUT/r3_bench_$cfg.json 2> $OUT/bench_$cfg.err
  echo "== $cfg rc=$?"; python -c "
import json
d=json.loads(open('$OUT/r3_bench_$cfg.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','ms_per_cagq_layer') if k in d}, {k:d[k].get('frac') for k in d if k.startswith('roofline')})"
done
for cfg in cfg4 cfg3 cfg2 cfg5; do
  st=50; [ $cfg = cfg5 ] && st=10
  timeout 900 python bench.py --config $cfg --dtype bf16 --steps $st --warmup 5 --no-cpu-baseline > $OUT/r3_bench_${cfg}_bf16.json 2> $OUT/bench_${cfg}_bf16.err
  echo "== $cfg bf16 rc=$?"; python -c "
import json
d=json.loads(open('$OUT/r3_bench_${cfg}_bf16.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','dtype') if k in d})"
done
