cd /tmp && export TMPDIR=/tmp
GG_DW_SHARED=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dws -o p -- python /root/repo/tools/time_dw.py 2>/dev/null | grep bwd
python - <<PY
import csv,glob
f=glob.glob('/tmp/dws/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
