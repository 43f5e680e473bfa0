#!/bin/bash
# dW kernel durations under the variant libraries built by tools/micro/mkvar.sh (edit the list below):
#   rocprofv3 kernel stats of tools/time_dw.py --dense per variant
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" _d6 _d5 _d4; do
  export GG_HIP_LIB=$R/grid_gcn_amd/lib/libgridgcn_hip$v.so
  rm -rf /tmp/o$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/o$v -o p -- python $R/tools/time_dw.py --dense > /tmp/log$v 2>&1
  F=$(find /tmp/o$v -name '*kernel_stats.csv' | head -1)
  echo "== variant '$v'"; python - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dw_direct" in r["Name"]: print("   %-60s avg %8.1f us" % (r["Name"][:60], float(r["AverageNs"])/1e3))
PY
done
