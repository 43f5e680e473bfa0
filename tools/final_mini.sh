#!/bin/bash
# After tools/final.sh has refreshed profiles/traffic.json: the full GPU suite once more and the headline lines again, so that
# the committed cfg4 lines carry the PMC figures of THIS code (bench.py looks `traffic` up in profiles/traffic.json).
#   final_mini.sh <outdir> <label>
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-mini}; L=${2:-rX}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
rm -f $OUT/${L}_float_parity.txt
GG_PARITY_REPORT=$R/$OUT/${L}_float_parity.txt timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $OUT/gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/gpu_tests.log
timeout 900 python bench.py --steps 50 --warmup 5 > $OUT/${L}_bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "cfg4 rc=$?"
timeout 900 python bench.py --dtype bf16 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${L}_bench_cfg4_bf16.json 2> $OUT/bench_cfg4_bf16.err; echo "cfg4 bf16 rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${L}_bench_cfg4_driver_cmd.json 2> $OUT/bench_driver.err; echo "driver cmd rc=$?"
python - <<PY
import json
for f in ("${L}_bench_cfg4.json", "${L}_bench_cfg4_bf16.json", "${L}_bench_cfg4_driver_cmd.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, {k: d[k] for k in ("value", "ms_per_step", "step_mode", "ms_per_cagq_layer") if k in d},
          {k: (d[k].get("frac"), d[k].get("traffic")) for k in d if k.startswith("roofline")})
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
