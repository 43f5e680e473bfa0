#!/bin/bash
# per-kernel average durations of the backward GEMM kernels (tools/time_dw.py, dense and sparse upstream
# gradient): prof_bwd_gemm.sh <outdir>
OUT=gpurun_out/${1:-bwd}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in dense sparse; do
  a=""; [ $m = dense ] && a="--dense"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/$m -o p -- python $R/tools/time_dw.py $a > $R/$OUT/$m.log 2>&1
  F=$(find $R/$OUT/$m -name '*kernel_stats.csv' | head -1)
  echo "== $m"; python - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"])>20000: print("%-70s n=%3s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
  rm -rf $R/$OUT/$m
done
