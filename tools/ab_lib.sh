#!/bin/bash
# A/B of a library option (gridgcn_set_option name) on the timed cfg4 step: ab_lib.sh <outdir> NAME [reps] [extra bench args]
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-ablib}; NAME=$2; REPS=${3:-2}; shift 3
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for v in 1 0; do
    timeout 600 python bench.py --steps 50 --warmup 10 --no-micro --no-cpu-baseline --switch $NAME=$v "$@" > $OUT/b_${NAME}_${v}_$rep.json 2> $OUT/b_${NAME}_${v}_$rep.err
    python -c "
import json
d=json.loads(open('$OUT/b_${NAME}_${v}_$rep.json').read().strip().splitlines()[-1])
print('$NAME=$v rep $rep', round(d['ms_per_step'],4), 'ms', round(d['value'],1))"
  done
done
