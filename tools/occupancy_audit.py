"""Per kernel of a rocprofv3 kernel trace (csv): launches, mean time, workgroup size, number of
workgroups, registers -> waves per SIMD the registers allow, workgroups resident per CU (registers
only; the LDS column is the STATIC part alone), and how many ROUNDS of workgroups the grid needs on 256 CUs.  A persistent
kernel whose grid is not a whole number of rounds wastes the difference."""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
acc = defaultdict(lambda: [0, 0.0, None])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:52]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    key = (name, wg, grid // wg)
    a = acc[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a[2] = (int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"]))
print("%-52s %5s %9s %5s %7s %5s %5s %7s %6s %7s %7s" % ("kernel", "n", "us", "wg", "nwg", "vgpr", "agpr", "lds", "w/SIMD", "wg/CU", "rounds"))
for (name, wg, nwg), (n, t, res) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    v, a, lds, scr = res
    # (rocprofv3's VGPR_Count on gfx950 is HALF the allocation of the code object's metadata --
    #  gg_k_att_bwd_fused<4>: 124 here, .vgpr_count 247 there -- so it is doubled)
    regs = ((2 * (v + a) + 7) // 8) * 8
    wps = min(8, 512 // max(regs, 8))
    waves = wg // 64
    per_cu = max(1, (wps * 4) // max(waves, 1)) if waves <= wps * 4 else 0
    if lds:
        per_cu = min(per_cu, max(1, (160 * 1024) // lds))
    rounds = nwg / (256.0 * per_cu) if per_cu else float("nan")
    print("%-52s %5d %9.1f %5d %7d %5d %5d %7d %6d %7d %7.2f%s" % (name, n, t / n, wg, nwg, v, a, lds, wps, per_cu, rounds, "  scratch %d" % scr if scr else ""))
