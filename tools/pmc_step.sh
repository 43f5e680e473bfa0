#!/bin/bash
# PMC passes over bench.py (cfg4): whole-step HBM bytes + MFMA-pipe utilisation, and the per-launch traffic of
# the micro-benchmarked kernels -> profiles/traffic.json, <outdir>/pmc_step.txt
# (separate --pmc passes, --kernel-trace only: MI355X_MICROARCH.md, rocprofv3 section)
export HSA_ENABLE_IPC_MODE_LEGACY=0
# pmc_step.sh <outdir> [bf16]: with bf16 only the whole-step passes, key step_cfg4_bf16 (bench.py --dtype bf16)
OUT=gpurun_out/${1:-pmc}
mkdir -p $OUT
if [ "$2" = "bf16" ]; then
  R=$GRAFT_REPO_ROOT
  cp profiles/traffic.json $OUT/traffic.json
  cd /tmp && export TMPDIR=/tmp
  ARGS="--steps 2 --warmup 1 --eager --no-cpu-baseline --no-micro --dtype bf16"
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/fetch16 -o p -- python $R/bench.py $ARGS > $R/$OUT/fetch16.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/write16 -o p -- python $R/bench.py $ARGS > $R/$OUT/write16.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/busy16 -o p -- python $R/bench.py $ARGS > $R/$OUT/busy16.log 2>&1
  cd $R
  python tools/pmc_step.py --fetch $OUT/fetch16 --write $OUT/write16 --busy $OUT/busy16 --out $OUT/traffic.json --key step_cfg4_bf16 > $OUT/pmc_step_bf16.txt
  head -12 $OUT/pmc_step_bf16.txt
  cp $OUT/traffic.json profiles/traffic.json
  exit 0
fi
R=$GRAFT_REPO_ROOT
cp profiles/traffic.json $OUT/traffic.json
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --eager --no-cpu-baseline --micro-iters 3"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/fetch -o p -- python $R/bench.py $ARGS > $R/$OUT/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/write -o p -- python $R/bench.py $ARGS > $R/$OUT/write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/busy -o p -- python $R/bench.py $ARGS --no-micro > $R/$OUT/busy.log 2>&1
cd $R
python tools/pmc_step.py --fetch $OUT/fetch --write $OUT/write --busy $OUT/busy --out $OUT/traffic.json > $OUT/pmc_step.txt
head -60 $OUT/pmc_step.txt
# per-launch traffic of the micro-benchmarked kernels: bench.py runs them after the steps, so the LAST launches
# of each kernel name are the benchmarked shape (5 warm-up + 3 timed)
python tools/pmc_traffic.py --fetch $OUT/fetch --write $OUT/write --key att_bwd_noz_E3276800_32to128 \
    --kernels "gg_k_att_bwd_nz,gg_k_att_nz_reduce,gg_k_att_nz_finish" --wide "gg_k_att_bwd_nz" --largest --out $OUT/traffic.json
python tools/pmc_traffic.py --fetch $OUT/fetch --write $OUT/write --key linear_fwd_E655360_256to128 \
    --kernels "gg_k_linear_fwd_direct<4" --wide "gg_k_linear_fwd_direct<4" --last 8 --out $OUT/traffic.json
python tools/pmc_traffic.py --fetch $OUT/fetch --write $OUT/write --key batch_take_up2_E3276800 \
    --kernels "gg_k_take<" --wide "" --last 8 --out $OUT/traffic.json
python tools/pmc_traffic.py --fetch $OUT/fetch --write $OUT/write --key gridconv_up2_E3276800 \
    --kernels gg_k_gridconv --wide "" --last 8 --out $OUT/traffic.json
python tools/pmc_traffic.py --fetch $OUT/fetch --write $OUT/write --key gridify_N81920_B8 \
    --kernels gg_k_chunk_split:327680,gg_k_slab_build:524288,gg_k_centre_slots:20480,gg_k_query_gridify:524288 --out $OUT/traffic.json
python -c "
import json; d=json.load(open('$OUT/traffic.json')); print({k:(round(v/1e6,1) if isinstance(v,float) and v>10 else v) for k,v in d.items() if not k.endswith('_detail')})"
cp $OUT/traffic.json profiles/traffic.json
