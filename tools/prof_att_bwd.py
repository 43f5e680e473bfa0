"""Run the kernels bench.py's `roofline` object times (backward of the 32->128 attention conv of
layer up2: gg_k_att_bwd_fused + gg_k_att_dw_reduce) a few times, for rocprofv3:

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out1 -- python tools/prof_att_bwd.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out2 -- python tools/prof_att_bwd.py
  python tools/prof_att_bwd.py --summarise out1 out2       # KB per launch and kernel
"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"][:48], r["Counter_Name"])
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
            for (k, c), v in sorted(agg.items()):
                if "gg_k" in k:
                    print("%-50s %-11s n=%3d avg=%12.1f KB" % (k, c, v[0], v[1] / v[0]))
    sys.exit(0)

from grid_gcn_amd.train import timers as ttimers  # noqa: E402

ms = ttimers.time_linear_bwd(655360, 5, 32, 128, iters=3, device="cuda:0", ndx=32, prev_bn=True)   # bench.py's call
print("ms per call", ms)
ms = ttimers.time_linear_fwd(655360, 256, 128, iters=3, device="cuda:0")
print("fwd ms per call", ms)
