#!/bin/bash
# one GPU session: bench lines of every BASELINE config + PMC traffic of the gridify call
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cp profiles/traffic.json $OUT/traffic.json        # other keys (att_bwd_fused: tools/r2_pmc_attbwd.sh) stay
if [ -z "$SKIP_PMC" ]; then
cd /tmp && export TMPDIR=/tmp
for cfg in seg80k synth200k; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_${cfg}_$c -o p -- python $R/tools/prof_index.py --cfg $cfg --iters 5 > $R/$OUT/pmc_${cfg}_$c.log 2>&1
  done
done
cd $R
python tools/pmc_traffic.py --fetch $OUT/pmc_seg80k_FETCH_SIZE --write $OUT/pmc_seg80k_WRITE_SIZE --key gridify_N81920_B8 --kernels gg_k_chunk_split:327680,gg_k_slab_build:524288,gg_k_centre_slots:163840,gg_k_query_gridify:524288 --out $OUT/traffic.json
python tools/pmc_traffic.py --fetch $OUT/pmc_synth200k_FETCH_SIZE --write $OUT/pmc_synth200k_WRITE_SIZE --key gridify_N200000_B8 --kernels gg_k_chunk_split:401408,gg_k_slab_build:524288,gg_k_centre_slots:401408,gg_k_query_gridify:2097152 --out $OUT/traffic.json
cp $OUT/traffic.json profiles/traffic.json
fi
cd $R
for cfg in cfg4 cfg1 cfg2 cfg3 cfg3up cfg5; do
  st=50; [ $cfg = cfg5 ] && st=10
  timeout 900 python bench.py --config $cfg --steps $st --warmup 5 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  echo "== $cfg rc=$?"; tail -c 600 $OUT/bench_$cfg.err | grep -v amdgpu.ids; python -c "
import json,sys
try:
    d=json.loads(open('$OUT/bench_$cfg.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('metric','value','ms_per_step','host_enqueue_ms_per_step','ms_per_cagq_layer') if k in d})
    for k in d:
        if k.startswith('roofline'): print(' ',k, {x:d[k].get(x) for x in ('achieved','unit','frac','traffic')})
    if 'cpu_baseline' in d: print('  cpu', d['cpu_baseline'])
except Exception as e: print('parse failed', e)
"
done
