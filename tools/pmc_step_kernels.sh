#!/bin/bash
# Per-kernel VALU/MFMA instruction counts, MFMA-busy and wait shares over an eager training step:
#   pmc_step_kernels.sh <outdir> <cfg> [extra bench args]     -> $OUT/kernels_<cfg>.txt
OUT=gpurun_out/${1:-pmck}; CFG=${2:-cfg4}; shift 2
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/k$i -o p -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --eager --no-micro --no-cpu-baseline "$@" > $R/$OUT/k$i.log 2>&1 || echo "group '$grp' failed"
done
cd $R
python tools/pmc_step_kernels.py $OUT/k1 $OUT/k2 $OUT/k3 > $OUT/kernels_$CFG.txt
rm -rf $OUT/k1 $OUT/k2 $OUT/k3
head -50 $OUT/kernels_$CFG.txt
