#!/bin/bash
# three-way A/B of a library option: ab_lib3.sh <outdir> NAME "v1 v2 v3" [reps]
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-ablib}; NAME=$2; VALS="$3"; REPS=${4:-2}
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    timeout 600 python bench.py --steps 50 --warmup 10 --no-micro --no-cpu-baseline --switch $NAME=$v > $OUT/b_${NAME}_${v}_$rep.json 2> $OUT/b_${NAME}_${v}_$rep.err
    python -c "
import json
d=json.loads(open('$OUT/b_${NAME}_${v}_$rep.json').read().strip().splitlines()[-1])
print('$NAME=$v rep $rep', round(d['ms_per_step'],4), 'ms', round(d['value'],1))"
  done
done
