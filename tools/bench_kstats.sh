#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench command, summarised: bench_kstats.sh <outdir> <cfg> <label> [bench args]
#   per-kernel totals per step (tools/kstats.py), the kernels that are NOT this library's (ATen, rocBLAS "Cijk_*",
#   runtime copies) listed separately, and -- cfg4 -- the micro-benchmarked launches of the `roofline` kernels
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-kst}; CFG=${2:-cfg4}; LABEL=${3:-r4}; shift 3
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
ST=10; [ $CFG = cfg5 ] && ST=4
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/t_$CFG -o p -- python $R/bench.py --config $CFG --steps $ST --warmup 3 --no-cpu-baseline "$@" > $R/$OUT/bench_$CFG.json 2> $R/$OUT/bench_$CFG.err
cd $R
F=$OUT/${LABEL}_bench_kernel_stats_$CFG.txt
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $CFG --steps $ST --warmup 3 --no-cpu-baseline $@   (hipGraph replays + the micro-benchmarks / inference of the line; tools/bench_kstats.sh)"
  echo "# bench line of this run: $(python -c "
import json
d=json.loads(open('$OUT/bench_$CFG.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','step_mode') if k in d}, {k:{x:d[k].get(x) for x in ('ms_per_launch','ms_in_step','frac')} for k in d if k.startswith('roofline') and isinstance(d[k], dict)})")"
  python tools/kstats.py $OUT/t_$CFG $((ST + 3))
  python - "$OUT/t_$CFG" <<'PY'
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if "gg_k_" in n:
        continue
    acc[n.split("(")[0].replace("void ", "")[:80]][0] += 1
    acc[n.split("(")[0].replace("void ", "")[:80]][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("# kernels of the whole trace that are not this library's (name, launches, total us):")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("#   %-80s %6d %10.1f" % (k, v[0], v[1]))
cijk = [k for k in acc if "Cijk" in k or "rocblas" in k.lower() or "gemm" in k.lower() and "gg_k" not in k]
print("# library GEMM kernels (rocBLAS / hipBLASLt) in the trace: %s" % (cijk if cijk else "NONE"))
PY
} > $F
rm -rf $OUT/t_$CFG
head -8 $F; grep "library GEMM" $F
