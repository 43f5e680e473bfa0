#!/bin/bash
# A/B of a train/options.py switch on the timed cfg4 step: ab.sh <outdir> NAME [reps]
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-ab}; NAME=$2; REPS=${3:-2}
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for v in 1 0; do
    timeout 600 python bench.py --steps 50 --warmup 10 --no-micro --switch $NAME=$v > $OUT/b_${v}_$rep.json 2> $OUT/b_${v}_$rep.err
    python -c "
import json
d=json.loads(open('$OUT/b_${v}_$rep.json').read().strip().splitlines()[-1])
print('$NAME=$v rep $rep', round(d['ms_per_step'],4), 'ms', round(d['value'],1))"
  done
done
