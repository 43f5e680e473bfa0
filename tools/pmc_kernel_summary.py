"""One row per (kernel, grid size) of rocprofv3 PMC passes: dispatches, mean duration, mean counter.
usage: python tools/pmc_kernel_summary.py <label>=<dir of a --pmc pass> ...  > profiles/..._kernel_stats.txt"""
import csv, glob, os, sys
from collections import defaultdict
print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python tools/prof_index.py --cfg <cfg> --iters 5")
print("# (tools/final.sh; MI355X).  One row per (kernel, grid size): dispatches, mean duration in THIS")
print("# (counter-collecting, hence slower) pass, mean counter value in KB.  The un-profiled device time of a whole")
print("# Gridify call is bench.py's ms_per_cagq_layer; traffic per call (profiles/traffic.json) = tools/pmc_traffic.py.")
for arg in sys.argv[1:]:
    label, _, d = arg.partition("=")
    acc = defaultdict(lambda: [0, 0.0, 0.0, ""])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if "gg_k" not in name:
                continue
            k = (name, int(r["Grid_Size"]))
            a = acc[k]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a[2] += float(r["Counter_Value"])
            a[3] = r["Counter_Name"]
    print("\n== %s" % label)
    for (name, grid), a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
        print("%-40s grid=%-9d n=%-4d avg %8.2f us   %s avg %10.1f KB" % (name[:40], grid, a[0], a[1] / a[0], a[3], a[2] / a[0]))
