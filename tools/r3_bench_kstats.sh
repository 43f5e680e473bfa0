#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (cfg4), summarised:
#   per-kernel totals per step (tools/kstats.py) + the micro-benchmarked launches of the `roofline` kernel
#   (the last 50 launches of gg_k_att_bwd_fused<4>: what bench.py's roofline.ms_per_launch is the median of)
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-kst}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/t -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/$OUT/bench.json 2> $R/$OUT/bench.err
cd $R
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline   (cfg4, hipGraph replays; tools/r3_bench_kstats.sh)"
  echo "# bench line of this run: $(python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print({k:d[k] for k in ('value','ms_per_step','step_mode')}, 'roofline', {k:d['roofline'][k] for k in ('ms_per_launch','frac')}, 'roofline_mfma', {k:d['roofline_mfma'][k] for k in ('ms_per_launch','frac')})")"
  python tools/kstats.py $OUT/t 13
  python - "$OUT/t" <<'PY'
import csv, glob, os, sys, statistics
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for name, label in (("gg_k_att_bwd_fused<4", "roofline"), ("gg_k_linear_fwd_direct<4, true, true, false>", "roofline_mfma")):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    last = d[-50:]
    print("# %s kernel %s: last 50 launches (the micro-benchmark) mean %.1f us, median %.1f us, min %.1f, max %.1f" % (
        label, name, sum(last) / len(last), statistics.median(last), min(last), max(last)))
PY
} > $OUT/r3_bench_kernel_stats_cfg4.txt
S=$(find $OUT/t -name '*kernel_stats.csv' | head -1); [ -n "$S" ] && head -25 "$S" > $OUT/r3_bench_kernel_stats_cfg4_rocprof.csv
rm -rf $OUT/t
head -12 $OUT/r3_bench_kernel_stats_cfg4.txt; tail -3 $OUT/r3_bench_kernel_stats_cfg4.txt
