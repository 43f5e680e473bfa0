"""HBM traffic per launch from rocprofv3 PMC passes -> profiles/traffic.json (read by bench.py).

Recipe (MI355X_MICROARCH.md, HBM section: separate --pmc passes, --kernel-trace only):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python tools/prof_index.py --cfg seg80k --iters 5
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python tools/prof_index.py --cfg seg80k --iters 5
    python tools/pmc_traffic.py --fetch $OUT/pmc_fetch --write $OUT/pmc_write --key gridify_N81920_B8 \
        --kernels gg_k_chunk_split:327680,gg_k_slab_build:524288,gg_k_centre_slots:163840,gg_k_query_gridify:524288

FETCH_SIZE / WRITE_SIZE are in KB.  gfx950 correction of the guide: FETCH_SIZE reports half of the
bytes of a wide (16 B per lane) coalesced streaming read -- applied (x2) ONLY to the kernels listed
in --wide (here gg_k_chunk_split, the float4 point stream); the other kernels gather 4-16 B pieces
and their counter value is taken as is (uncalibrated, as the guide says).  One "launch" of the key =
one call of the operator = the sum over its kernels of the per-dispatch average.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--key", required=True)
ap.add_argument("--kernels", required=True,
                help="comma separated name[:grid] of the kernels of one call (grid = total threads of "
                     "the dispatch, to tell the layers of a network apart)")
ap.add_argument("--wide", default="gg_k_chunk_split", help="kernels whose FETCH_SIZE gets the x2")
ap.add_argument("--grid-min", type=int, default=0, help="only dispatches with at least this grid size")
ap.add_argument("--last", type=int, default=0,
                help="per kernel name keep only the last N dispatches of the pass (bench.py runs its "
                     "micro-benchmarks after the training steps: their launches are the last ones)")
ap.add_argument("--largest", action="store_true",
                help="per kernel name keep only the dispatches with the LARGEST grid (the micro-benchmarked "
                     "shape when the same kernel also runs at smaller shapes in the step)")
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                              "profiles", "traffic.json"))
a = ap.parse_args()
kernels, grids = [], {}
for item in a.kernels.split(","):
    name, _, g = item.partition(":")
    kernels.append(name)
    if g:
        grids[name] = int(g)
wide = set(a.wide.split(",")) if a.wide else set()


def per_kernel(dirname, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            grid = int(row.get("Grid_Size", 0) or 0)
            if grid < a.grid_min:
                continue
            for k in kernels:
                if k in name and (k not in grids or grids[k] == grid):
                    acc[k].append((grid, float(row["Counter_Value"]), int(row["Start_Timestamp"])))
    out = {}
    for k, v in acc.items():
        if a.last:
            v = sorted(v, key=lambda x: x[2])[-a.last:]
        if a.largest:
            g = max(x[0] for x in v)
            v = [x for x in v if x[0] == g]
        out[k] = sum(x[1] for x in v) / len(v)
    return out


fe, wr = per_kernel(a.fetch, "FETCH_SIZE"), per_kernel(a.write, "WRITE_SIZE")
total = 0.0
detail = {}
for k in kernels:
    f = fe.get(k, 0.0) * 1024.0 * (2.0 if k in wide else 1.0)
    w = wr.get(k, 0.0) * 1024.0
    detail[k] = {"fetch_bytes": f, "write_bytes": w}
    total += f + w
try:
    cur = json.load(open(a.out))
except (OSError, ValueError):
    cur = {}
cur[a.key] = total
cur[a.key + "_detail"] = detail
json.dump(cur, open(a.out, "w"), indent=1, sort_keys=True)
print(a.key, "%.2f MB per launch" % (total / 1e6), json.dumps(detail))
