#!/bin/bash
# GPU session: session.sh <outdir> [what...]   what = tests micro bench trace pmc configs (default: all but pmc/configs)
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GG_R6_UNVERIFIED=1   # tests/test_zz_r6_unverified.py: the round-6 opt-in paths
OUT=gpurun_out/${1:-session}; shift
WHAT="${@:-tests micro bench trace}"
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  rm -f $OUT/float_parity.txt
  GG_PARITY_REPORT=$R/$OUT/float_parity.txt timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $OUT/gpu_tests.log 2>&1
  echo "pytest rc=$?"; tail -12 $OUT/gpu_tests.log
fi
if has micro; then
  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mem.hip -o /tmp/mfma_mem 2>$OUT/micro_build.err
  timeout 600 /tmp/mfma_mem > $OUT/micro_overlap.txt 2>&1; echo "micro rc=$?"; cat $OUT/micro_overlap.txt
fi
if has bench; then
  timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "bench rc=$?"
  python - <<PY
import json
d=json.loads(open('$OUT/bench_cfg4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','ms_per_cagq_layer') if k in d})
for k in d:
    if k.startswith('roofline'): print(k, {x:d[k].get(x) for x in ('frac','frac_micro','ms_per_launch','ms_in_step','launches_per_step')})
print('inference', d.get('inference'))
PY
fi
if has trace; then
  bash tools/steptrace.sh ${OUT#gpurun_out/}/trace cfg4 150 --no-micro > /dev/null 2>&1
  python - <<PY
import glob
f='$OUT/trace/steptrace_cfg4.txt'
print(open(f).read())
PY
fi
if has configs; then
  for cfg in cfg1 cfg2 cfg3 cfg5; do
    st=50; [ $cfg = cfg5 ] && st=10
    timeout 900 python bench.py --config $cfg --steps $st --warmup 5 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
    echo "== $cfg rc=$?"; python -c "
import json
d=json.loads(open('$OUT/bench_$cfg.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','ms_per_cagq_layer') if k in d}, {k:d[k].get('frac') for k in d if k.startswith('roofline')})"
  done
fi
