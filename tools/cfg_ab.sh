export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/s6
for cfg in cfg2 cfg5 cfg3; do
  st=30; [ $cfg = cfg5 ] && st=10
  for v in 1 0; do
    timeout 600 python bench.py --config $cfg --steps $st --warmup 5 --no-cpu-baseline --no-micro --switch BWD_FUSED128=$v > gpurun_out/s6/b_${cfg}_$v.json 2> gpurun_out/s6/b_${cfg}_$v.err
    python -c "
import json
d=json.loads(open('gpurun_out/s6/b_${cfg}_$v.json').read().strip().splitlines()[-1])
print('$cfg BWD_FUSED128=$v', round(d['ms_per_step'],3), 'ms', round(d['value'],1))"
  done
done
