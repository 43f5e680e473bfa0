"""Instructions per point of the four Gridify kernels ("the batch curve says the instruction
stream is the roof -- report it").  Two modes:

    python tools/gridify_insts.py --run [--cfg seg80k|synth200k]    # 20 Gridify calls of layer 0 (the profiled command)
    python tools/gridify_insts.py --report DIR [DIR ...] --points N  # rocprofv3 --pmc csv directories -> table

Counters (per WAVE instruction, as the SQ counts them): SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS, SQ_WAVES in one
pass, SQ_INSTS_VMEM_RD, SQ_INSTS_VMEM_WR, SQ_INSTS_SMEM in a second (tools/gridify_insts.sh).  "per point" = the
launch's counter / (B * N points of the call); a wave instruction covers up to 64 lanes."""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("--run", action="store_true")
ap.add_argument("--cfg", default="seg80k")
ap.add_argument("--report", nargs="*")
ap.add_argument("--points", type=int, default=8 * 81920)
a = ap.parse_args()

if a.run:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import torch
    from grid_gcn_amd import ops, synth
    cfg, B = (synth.SEG_SCANNET_81920, 8) if a.cfg == "seg80k" else (synth.SYNTH_200K, 8)
    data, npn = synth.make_batch(B, cfg["num_points"], "planes")
    d = torch.from_numpy(np.ascontiguousarray(data)).to("cuda:0")
    n = torch.from_numpy(np.ascontiguousarray(npn)).to("cuda:0")
    kw = synth.gridify_kwargs(cfg, 0)
    for _ in range(20):
        ops.Gridify(d, n, **kw)
    torch.cuda.synchronize()
    print("ran 20 Gridify calls: B %d N %d" % (B, cfg["num_points"]))
    sys.exit(0)

per = defaultdict(lambda: defaultdict(list))
for d in a.report:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("gg_k_"):
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
         "SQ_INSTS_SMEM"]
print("# Gridify layer 0, %d points per call; median over the profiled calls; wave instructions PER POINT" % a.points)
print("%-34s %8s %8s %8s %8s %8s %8s %8s %10s" % ("kernel", "waves", "VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR", "SMEM",
                                                  "insts/wave"))
tot = defaultdict(float)
for k in sorted(per, key=lambda k: ("chunk" not in k, "slab" not in k, "centre" not in k, k)):
    v = per[k]
    med = {c: sorted(v[c])[len(v[c]) // 2] if v.get(c) else float("nan") for c in names}
    ins = sum(med[c] for c in names[1:] if med[c] == med[c])
    print("%-34s %8.0f %8.2f %8.2f %8.2f %8.3f %8.3f %8.3f %10.0f" % (
        k[:34], med["SQ_WAVES"], *[med[c] / a.points for c in names[1:]], ins / med["SQ_WAVES"]))
    for c in names[1:]:
        if med[c] == med[c]:
            tot[c] += med[c]
print("%-34s %8s %8.2f %8.2f %8.2f %8.3f %8.3f %8.3f" % ("all four kernels", "", *[tot[c] / a.points for c in names[1:]]))
