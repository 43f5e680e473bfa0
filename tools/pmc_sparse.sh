#!/bin/bash
# HBM fetch / write bytes per launch of gg_k_edge_lin0_bwd_sparse at the cfg4 up2 shape (tools/time_sparse.py)
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ps_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/ps_$c -o p -- python $R/tools/time_sparse.py > /tmp/ps_$c.log 2>&1
  python - /tmp/ps_$c $c <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
     if "edge_lin0_bwd_sparse" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print("%s: %d launches, mean %.1f MB per launch (counter in KB, no correction)" % (sys.argv[2], len(v), sum(v) / len(v) / 1024))
PY
done
