"""One row per kernel of an eager training step from three rocprofv3 --pmc passes (tools/pmc_step_kernels.sh):
time share, MFMA instructions, VALU instructions per MFMA, MFMA-pipe busy share, share of wave time spent waiting.
On gfx950 a VALU instruction costs the SIMD ~5 cycles that the MFMA pipe cannot use (tools/micro/mfma_valu.hip):
expected MFMA-rate ceiling of a kernel = 64 / (64 + 5.2 * VALU/MFMA)."""
import csv, glob, os, sys
from collections import defaultdict
val = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(float)
nd = defaultdict(int)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            val[name][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[name][r["Counter_Name"]] += 1
            dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            nd[name] += 1
tot = sum(dur.values())
print("%-58s %6s %7s %9s %9s %8s %8s %8s" % ("kernel", "calls", "time %", "avg us", "MFMA/call", "VALU/MFMA", "MFMAbusy", "wait"))
for name in sorted(dur, key=lambda n: -dur[n])[:45]:
    c = {k: v / cnt[name][k] for k, v in val[name].items()}
    g = c.get
    mf = g("SQ_INSTS_MFMA", 0)
    v = (g("SQ_INSTS_VALU", 0) - mf) / mf if mf else float("nan")
    busy = 100 * g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / g("SQ_BUSY_CU_CYCLES") if g("SQ_BUSY_CU_CYCLES") else float("nan")
    wait = 100 * g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else float("nan")
    ncalls = max(cnt[name].values())
    print("%-58s %6d %6.1f%% %9.1f %9.0f %8.2f %7.1f%% %7.1f%%" % (name[:58], ncalls, 100 * dur[name] / tot, dur[name] / nd[name], mf, v, busy, wait))
