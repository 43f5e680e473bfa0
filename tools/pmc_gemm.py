"""Per-kernel means of every counter found in the rocprofv3 --pmc pass directories given (tools/pmc_gemm.sh)."""
import csv, glob, os, sys
from collections import defaultdict
val = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "gg_k_linear_d" not in name and "gg_k_att_bwd" not in name and "gg_k_linear_fwd" not in name:
                continue
            a = val[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            t = dur[name]
            t[0] += 1
            t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name in sorted(val, key=lambda n: -dur[n][1] / dur[n][0]):
    c = {k: v[1] / v[0] for k, v in val[name].items()}
    print("== %s   (avg %.1f us in the counter passes)" % (name, dur[name][1] / dur[name][0]))
    g = c.get
    if g("SQ_BUSY_CU_CYCLES"):
        print("   MFMA busy            %5.1f %%" % (100 * g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / g("SQ_BUSY_CU_CYCLES")))
    if g("SQ_WAVE_CYCLES"):
        print("   wave waiting (inst)  %5.1f %%" % (100 * g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")))
    if g("SQ_INSTS_MFMA"):
        print("   VALU / MFMA          %5.2f    (VALU %d, MFMA %d, SALU %d, VMEM rd %d)" % (
            (g("SQ_INSTS_VALU", 0) - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU", 0), g("SQ_INSTS_MFMA"),
            g("SQ_INSTS_SALU", 0), g("SQ_INSTS_VMEM_RD", 0)))
    for k in sorted(c):
        print("   %-34s %14.0f" % (k, c[k]))
