#!/bin/bash
# A/B of every path switch of the training step on the timed cfg4 step, one session:
#   ab_all.sh <outdir> [reps]  ->  gpurun_out/<outdir>/ab_switches.txt
# Each row: the step with ONE switch off against the shipped configuration (all on), `reps` runs each,
# 50 replays of the captured step after 10 warm-up steps (bench.py --no-micro).
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-ab}; REPS=${2:-2}
mkdir -p $OUT
run() { timeout 600 python bench.py --steps 50 --warmup 10 --no-micro --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'])"; }
{
  echo "# cfg4 step (ms), one switch off at a time against the shipped configuration; bench.py --steps 50 --warmup 10 --no-micro; tools/ab_all.sh"
  printf "%-22s" "shipped (all on)"; for r in $(seq 1 $REPS); do printf " %s" $(run); done; echo
  for sw in COL_SPLIT NOZ_ATT_FWD NOZ_ATT_BWD ATT_NZ_V2 BWD_FUSED128 GLUE_KERNELS OWN_ADAM SRC_STATS FOLD_FINALIZE WGB_PREPACK FUSE_DROPOUT; do
    printf "%-22s" "$sw=0"; for r in $(seq 1 $REPS); do printf " %s" $(run --switch $sw=0); done; echo
  done
  printf "%-22s" "shipped (all on)"; for r in $(seq 1 $REPS); do printf " %s" $(run); done; echo
} | tee $OUT/ab_switches.txt
