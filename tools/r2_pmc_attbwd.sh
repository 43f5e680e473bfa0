#!/bin/bash
# HBM traffic of bench.py's `roofline` kernel -> profiles/traffic.json (key att_bwd_fused_E3276800_32to128)
R=$PWD; O=$R/gpurun_out/pmc_attbwd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python $R/tools/prof_att_bwd.py > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python $R/tools/prof_att_bwd.py > $O/w.log 2>&1
cd $R
python tools/prof_att_bwd.py --summarise $O/fetch $O/write | tee $O/summary.txt
python tools/pmc_traffic.py --fetch $O/fetch --write $O/write --key att_bwd_fused_E3276800_32to128 \
    --kernels gg_k_att_bwd_fused,gg_k_att_dw_reduce --wide gg_k_att_bwd_fused
# (float4 streaming reads: FETCH_SIZE x2 -- gg_k_linear_fwd_direct in the same run counts 328 MB for the 671 MB it reads)
