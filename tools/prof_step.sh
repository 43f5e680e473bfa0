#!/bin/bash
# One rocprofv3 kernel trace of the bench workload, summarised per training step.
# usage (on the GPU box): bash tools/prof_step.sh [outdir]   -> <outdir>/step.txt, kernel_stats.csv
out=${1:-gpurun_out/prof}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ggprof
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ggprof -o run -- \
    python "$repo/bench.py" --steps 6 --warmup 3 > /tmp/ggprof_bench.log 2>&1
tail -1 /tmp/ggprof_bench.log > "$repo/$out/bench.json"
f=$(find /tmp/ggprof -name "*kernel_trace.csv" | head -1)
s=$(find /tmp/ggprof -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp "$s" "$repo/$out/kernel_stats.csv"
python "$repo/tools/step_trace.py" "$f" 150 ${2:-45} > "$repo/$out/step.txt" 2>&1
tail -60 "$repo/$out/step.txt"
