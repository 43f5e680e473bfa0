"""Summarise one training step from a rocprofv3 kernel trace CSV (bench.py run).
usage: python tools/step_trace.py <kernel_trace.csv> [min_us] [top_n] [narrow]
narrow: also list every launch of fewer than 256 workgroups that ran longer than 8 us"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "gg_k_ce_fwd" in r["Kernel_Name"]] or \
      [i for i, r in enumerate(rows) if "nll_loss_forward_reduce" in r["Kernel_Name"]]
seq = rows[idx[-2]:idx[-1]]
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in seq:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    k = r["Kernel_Name"][:60]
    agg[k][0] += 1
    agg[k][1] += d
    if d > minus:
        print("%9.1f us %-46s grid=%-9s wg=%s lds=%s" % (d, k[:46], r["Grid_Size_X"], r["Workgroup_Size_X"],
                                                      r.get("LDS_Block_Size", "")))
print("---- by kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("%-62s n=%4d tot=%9.1f us" % (k, v[0], v[1]))
if len(sys.argv) > 4:
    print("---- fewer than 256 workgroups, longer than 8 us")
    for r in seq:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wgs = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * \
              (int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))) * \
              (int(r.get("Grid_Size_Z", 1)) // max(1, int(r.get("Workgroup_Size_Z", 1))))
        if wgs < 256 and d > 8:
            print("%9.1f us wgs=%5d %s" % (d, wgs, r["Kernel_Name"][:70]))
print("sum %.1f ms, kernels %d, wall %.1f ms" % (
    tot / 1e3, len(seq), (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e6))
