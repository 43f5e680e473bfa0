"""Per-kernel time of the training step from a rocprofv3 kernel trace (csv).
usage: python tools/step_stats.py <dir with *kernel_trace.csv> <steps traced> [out.txt]
Rows: kernel name (+ grid), launches per step, us per launch, ms per step, share."""
import csv
import glob
import os
import sys
from collections import defaultdict

d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# one step = from one softmax cross-entropy forward (exactly one per step) to the next: the window
# covers `steps` steps starting at the 4th loss evaluation (warm-up excluded)
ce = [i for i, r in enumerate(rows) if "gg_k_ce_fwd" in r["Kernel_Name"]]
acc = defaultdict(lambda: [0, 0.0])
t_first, t_last = None, None
start, end = ce[3], ce[3 + steps]
for r in rows[start:end]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    acc[name][0] += 1
    acc[name][1] += dur
    t_first = int(r["Start_Timestamp"]) if t_first is None else t_first
    t_last = int(r["End_Timestamp"])
tot = sum(v[1] for v in acc.values())
lines = ["# %d steps; kernels per step %.1f; sum of kernel time %.3f ms/step; wall span %.3f ms/step" % (
    steps, sum(v[0] for v in acc.values()) / steps, tot / steps / 1000.0, (t_last - t_first) / 1e6 / steps)]
for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    lines.append("%-70s n/step %6.1f  us/launch %9.2f  ms/step %7.3f  %5.1f%%" % (
        name[:70], n / steps, t / n, t / steps / 1000.0, 100.0 * t / tot))
out = "\n".join(lines)
print(out[:6000])
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
