#!/bin/bash
# gridify_insts.sh <outdir>: SQ instruction counters of the Gridify kernels (cfg4 layer 0, cfg5 layer 0) -> <outdir>/gridify_insts.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-ginsts}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in seg80k synth200k; do
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/$OUT/a_$cfg -o p -- python $R/tools/gridify_insts.py --run --cfg $cfg > $R/$OUT/a_$cfg.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $R/$OUT/b_$cfg -o p -- python $R/tools/gridify_insts.py --run --cfg $cfg > $R/$OUT/b_$cfg.log 2>&1
done
cd $R
{ python tools/gridify_insts.py --report $OUT/a_seg80k $OUT/b_seg80k --points $((8*81920)); echo; python tools/gridify_insts.py --report $OUT/a_synth200k $OUT/b_synth200k --points $((8*200000)); } > $OUT/gridify_insts.txt 2>&1
cat $OUT/gridify_insts.txt
