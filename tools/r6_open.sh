#!/bin/bash
# First GPU session of round 6 (the pool was closed while the round's kernels were written):  r6_open.sh <outdir>
#   1. the GPU tier as shipped;  2. the round-6 opt-in tests (GG_R6_UNVERIFIED=1);  3. the driver's bench command;
#   4. A/B of the round-6 switches on the timed step (bench.py --switch NAME=1 against the shipped configuration).
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r6_open}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x > $OUT/gpu_tests.log 2>&1; echo "pytest(shipped) rc=$?"; tail -3 $OUT/gpu_tests.log
GG_R6_UNVERIFIED=1 GG_PARITY_REPORT=$R/$OUT/float_parity.txt timeout 600 python -m pytest tests/test_zz_r6_unverified.py -q -m gpu --timeout 300 > $OUT/gpu_tests_r6.log 2>&1; echo "pytest(r6 opt-in) rc=$?"; tail -5 $OUT/gpu_tests_r6.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg4_driver_cmd.json 2> $OUT/bench.err; echo "bench rc=$?"
run() { timeout 600 python bench.py --steps 50 --warmup 10 --no-micro --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'])"; }
{
  echo "# cfg4 step (ms): shipped configuration against each round-6 switch ON; bench.py --steps 50 --warmup 10 --no-micro; tools/r6_open.sh"
  printf "%-28s" "shipped"; for r in 1 2 3; do printf " %s" $(run); done; echo
  for sw in NOZ_BWD_MOMENTS INDEX_SIDE_STREAM; do
    printf "%-28s" "$sw=1"; for r in 1 2 3; do printf " %s" $(run --switch $sw=1); done; echo
  done
  printf "%-28s" "both"; for r in 1 2 3; do printf " %s" $(run --switch NOZ_BWD_MOMENTS=1 --switch INDEX_SIDE_STREAM=1); done; echo
  printf "%-28s" "shipped"; for r in 1 2 3; do printf " %s" $(run); done; echo
} | tee $OUT/ab_r6.txt
python - <<PY
import json
d=json.loads(open('$OUT/bench_cfg4_driver_cmd.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','ms_per_cagq_layer') if k in d})
PY
# the other BASELINE configs at HEAD (round 6 touched gg_k_ctx_max / gg_k_dz_segsum of the classifier and gg_k_query_up_lanes of
# the GridifyUp path: round-5 figures cfg2 16.44 ms, cfg3up 4.25 ms, cfg3 4.24 ms, cfg5 21.02 ms)
for cfg in cfg2 cfg3up cfg3 cfg5; do
  st=30; [ $cfg = cfg5 ] && st=10
  timeout 600 python bench.py --config $cfg --steps $st --warmup 5 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  echo "== $cfg rc=$?"; python -c "
import json
d=json.loads(open('$OUT/bench_$cfg.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode','ms_per_cagq_layer') if k in d})"
done
