"""Whole-step HBM traffic and MFMA-pipe utilisation from rocprofv3 PMC passes over bench.py
(tools/pmc_step.sh) -> profiles/traffic.json + a per-kernel table.

    python tools/pmc_step.py --fetch DIR --write DIR --busy DIR [--out profiles/traffic.json] > table

One STEP = the dispatches from one gg_k_ce_fwd launch (the loss of step i) up to the next one
(zero_grad, backward of step i, Adam, forward of step i+1: every kernel of a step exactly once).
FETCH_SIZE / WRITE_SIZE are in KB.  gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE
reports HALF of the bytes of a wide (16 B per lane) coalesced streaming read; it is applied (x2) to the
kernels in WIDE -- the ones whose bulk reads are dwordx4 row streams -- and to nothing else; both sums
are printed.  MFMA-pipe utilisation of the step = sum(SQ_VALU_MFMA_BUSY_CYCLES) / 4 SIMDs per CU over
sum(SQ_BUSY_CU_CYCLES) (the same formula as profiles/r2_pmc_bwd_gemm.txt).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

WIDE = ("gg_k_linear_fwd_direct", "gg_k_linear_dx_direct", "gg_k_att_bwd_fused", "gg_k_bn_apply",
        "gg_k_bn_bwd_reduce", "gg_k_pairmax", "gg_k_edge_lin0", "gg_k_chunk_split",
        "gg_k_ce_", "gg_k_colsum", "gg_k_linear_fwd<", "multi_tensor_apply", "gg_k_dw_reduce",
        "gg_k_att_dw_reduce", "gg_k_att_max_eval", "gg_k_att_bwd_nz", "gg_k_adam")

ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--busy", default="")
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                              "profiles", "traffic.json"))
ap.add_argument("--key", default="step_cfg4")
a = ap.parse_args()


def rows_of(d):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        out += list(csv.DictReader(open(f)))
    out.sort(key=lambda r: (int(r["Start_Timestamp"]), r["Counter_Name"]))
    return out


def one_step(rows):
    """dispatch ids of the LAST complete step (between the last two gg_k_ce_fwd launches)"""
    seen, marks = set(), []
    for r in rows:
        did = r["Dispatch_Id"]
        if did in seen:
            continue
        seen.add(did)
        if "gg_k_ce_fwd" in r["Kernel_Name"]:
            marks.append(int(r["Start_Timestamp"]))
    assert len(marks) >= 2, "need two steps in the pass"
    lo, hi = marks[-2], marks[-1]
    return [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]


def short(name):
    return name.split("(")[0].replace("void ", "")[:44]


# kernels whose reads are PARTLY dwordx4 row streams: factor = 1 + (share of the fetched bytes that are wide)
# gg_k_linear_bwd_fused128: Z and dY as 16-byte row reads (2/3 of its HBM reads), X as coalesced dword rows
# gg_k_att_pairmax: Z1 rows and edge records as 16-byte reads (630 of its ~730 MB), the gathered source rows and the
#   neighbour indices as dword reads: (630 + 97) / (315 + 97)
PARTLY = {"gg_k_linear_bwd_fused128": 1.5, "gg_k_att_pairmax": 1.75}


def is_wide(name):
    return any(w in name for w in WIDE)


def fetch_factor(name):
    for k, f in PARTLY.items():
        if k in name:
            return f
    return 2.0 if is_wide(name) else 1.0


per = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum over the step
cnt = defaultdict(int)
tot = defaultdict(float)
for d, want in ((a.fetch, ("FETCH_SIZE",)), (a.write, ("WRITE_SIZE",)),
                (a.busy, ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"))):
    if not d:
        continue
    st = one_step(rows_of(d))
    seen = set()
    for r in st:
        c = r["Counter_Name"]
        if c not in want:
            continue
        k = short(r["Kernel_Name"])
        v = float(r["Counter_Value"])
        per[k][c] += v
        tot[c] += v
        if c == "FETCH_SIZE":
            per[k]["FETCH_x2"] += v * fetch_factor(r["Kernel_Name"])
            tot["FETCH_x2"] += v * fetch_factor(r["Kernel_Name"])
            if r["Dispatch_Id"] not in seen:
                cnt[k] += 1
                seen.add(r["Dispatch_Id"])

raw = (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
cor = (tot["FETCH_x2"] + tot["WRITE_SIZE"]) * 1024.0
print("# one training step of bench.py (cfg4, B = 8 x 81920 points, eager) under rocprofv3 --pmc, MI355X")
print("# HBM bytes of the step: %.3f GB with the counters as they are; %.3f GB with FETCH_SIZE x2 on the "
      "dwordx4 row-streaming kernels (MI355X_MICROARCH HBM section) -- the figure in traffic.json" % (raw / 1e9, cor / 1e9))
busy = None
if tot.get("SQ_BUSY_CU_CYCLES"):
    busy = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 4.0 / tot["SQ_BUSY_CU_CYCLES"]
    print("# MFMA pipe busy over the step: %.1f %% of the busy CU cycles (sum MFMA_BUSY / 4 SIMDs / sum BUSY_CU)"
          % (100 * busy))
print("%-46s %4s %10s %10s %10s %8s" % ("kernel", "n", "fetch MB", "fetchx2 MB", "write MB", "MFMAbusy"))
for k, v in sorted(per.items(), key=lambda kv: -(kv[1]["FETCH_x2"] + kv[1]["WRITE_SIZE"]))[:45]:
    mb = ""
    if v.get("SQ_BUSY_CU_CYCLES"):
        mb = "%5.1f%%" % (100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / 4.0 / v["SQ_BUSY_CU_CYCLES"])
    print("%-46s %4d %10.1f %10.1f %10.1f %8s" % (k, cnt[k], v["FETCH_SIZE"] / 1024, v["FETCH_x2"] / 1024,
                                                  v["WRITE_SIZE"] / 1024, mb))
try:
    cur = json.load(open(a.out))
except (OSError, ValueError):
    cur = {}
cur[a.key] = cor
cur[a.key + "_raw_counters"] = raw
if busy is not None:
    cur[a.key + "_mfma_busy"] = busy
json.dump(cur, open(a.out, "w"), indent=1, sort_keys=True)
