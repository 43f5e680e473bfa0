"""Total time per kernel over a whole rocprofv3 kernel trace (csv), divided by `nsteps`."""
import csv, glob, os, sys
from collections import defaultdict
d, nsteps = sys.argv[1], float(sys.argv[2])
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
acc = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]
    acc[name][0] += 1
    acc[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
tot = sum(v[1] for v in acc.values())
print("# all launches of the trace / %g steps: %.3f ms/step, %d launches/step" % (nsteps, tot / nsteps / 1e3, sum(v[0] for v in acc.values()) / nsteps))
for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-90s n/step %6.1f us/launch %9.2f ms/step %7.3f %5.1f%%" % (name, n / nsteps, t / n, t / nsteps / 1e3, 100 * t / tot))
