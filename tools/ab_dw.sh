cd /tmp && export TMPDIR=/tmp
for v in A B; do
  if [ $v = B ]; then export GG_HIP_LIB=/root/repo/grid_gcn_amd/lib/libgridgcn_hip_B.so; else unset GG_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dw_$v -o p -- python /root/repo/tools/time_dw.py 2>/dev/null | grep bwd
  python - <<PY
import csv,glob
f=glob.glob('/tmp/dw_$v/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print('$v', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
