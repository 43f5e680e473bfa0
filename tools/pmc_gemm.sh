#!/bin/bash
# Counter passes over tools/time_dw.py (backward GEMM kernels, 655 360 rows): pmc_gemm.sh <outdir> [--dense]
# One rocprofv3 --pmc pass per counter group (kernel-trace only), summarised per kernel by tools/pmc_gemm.py.
OUT=gpurun_out/${1:-pmcg}; shift
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/g$i -o p -- python $R/tools/time_dw.py "$@" > $R/$OUT/g$i.log 2>&1 || echo "group '$grp' failed"
done
cd $R
python tools/pmc_gemm.py $OUT/g* | tee $OUT/summary.txt
rm -rf $OUT/g*/
