export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/s12
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_af -o af -- python $GRAFT_REPO_ROOT/tools/time_attfwd.py > $GRAFT_REPO_ROOT/gpurun_out/s12/time.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_af -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/s12/kernels.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %5s avg %9.1f us" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
