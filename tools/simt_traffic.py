"""Memory traffic of the kernels, COUNTED on the emulator (no GPU needed):

    python tools/simt_traffic.py gridify [cfg4|cfg5|cfg3] [--batch B] [--lines N]   ->  profiles/r6_emulated_traffic.txt
    python tools/simt_traffic.py att_bwd [--ncent N]                               (the Z2-free attention backward, per edge)
    python tools/simt_traffic.py step                                              ->  profiles/r6_emulated_step_lines.txt

The race build of the emulated library (tests/simt/simt_race.cpp) sees every global load and store of every work-item
with its address.  With the race checks off and `simt_traffic_enable(1)` it keeps, per launch, which 128-byte lines each
XCD touched and which 32-byte sectors were stored to, and reports per kernel
    requested   bytes asked for by the work-items
    fetched     lines x 128 an XCD reads without having touched them earlier in the launch, summed over 8 XCDs
    written     distinct sectors x 32 stored to
(definitions and what they bracket: the header of simt_race.cpp).  For the configurations whose counter tables were
taken on the GPU in round 5 (profiles/traffic.json: FETCH_SIZE x 2 / WRITE_SIZE per kernel) the two are printed side by
side; --lines adds the source lines that fetch / write the most.  TEST INFRASTRUCTURE: nothing here is a measurement of
the machine -- it is the traffic the SOURCE asks for, which is what a change to the source can be judged by while no GPU
is at hand."""
import argparse
import collections
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GG_SIMT_RACE"] = "1"
os.environ.pop("GG_SIMT_RACE_REPORT", None)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from simt import sim, emu  # noqa: E402
from grid_gcn_amd import synth  # noqa: E402


def lib():
    L = emu.library()
    L.simt_traffic_report.restype = ctypes.c_int
    L.simt_traffic_report.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    return L


def start():
    L = lib()
    L.simt_race_enable(0)
    L.simt_traffic_reset()
    L.simt_traffic_enable(1)


def report(per_pc=False):
    """({kernel: dict(launches, wgs, req_ld, req_st, fetched, written)}, {kernel: [(file:line, req_ld, req_st, fetched,
    written)]})"""
    L = lib()
    n = L.simt_traffic_report(None, 0, 1 if per_pc else 0)
    buf = ctypes.create_string_buffer(n + 16)
    L.simt_traffic_report(buf, len(buf), 1 if per_pc else 0)
    ker, pcs = {}, collections.defaultdict(list)
    rows = [ln.split("\t") for ln in buf.value.decode().splitlines() if ln]
    for r in rows:
        if r[0] == "K":
            ker[r[1]] = dict(zip(("launches", "wgs", "req_ld", "req_st", "fetched", "written"), map(int, r[2:8])))
    prow = [r for r in rows if r[0] == "P"]
    if prow:
        so = next((r[2] for r in prow if r[2] != "?"), "?")
        sym = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + so, "--functions=linkage",
                              "--no-inlines", "--no-demangle"] + [hex(int(r[3], 16) - 1) for r in prow],
                             capture_output=True, text=True).stdout
        pairs = [ln.strip() for ln in sym.splitlines() if ln.strip()]
        funcs, locs = pairs[0::2], pairs[1::2]
        for r, fn, l in zip(prow, funcs, locs):
            f, _, rest = os.path.basename(l).partition(":")
            ln = rest.split(":")[0]
            if f.endswith(".simt.cpp") and ln.isdigit():
                f, ln = f.replace(".simt.cpp", ".hip"), str(int(ln) - 1)
            pcs[r[1]].append(("%s:%s" % (f, ln),) + tuple(map(int, r[4:8])))
    return ker, pcs


def short(k):
    return k.split("<")[0].strip()


def table(ker, pcs, pmc=None, top=0, out=sys.stdout):
    agg = collections.OrderedDict()
    for k, v in ker.items():
        a = agg.setdefault(short(k), collections.Counter())
        a.update(v)
    w = out.write
    w("%-26s %8s %8s | %10s %10s | %10s %10s" % ("kernel", "launches", "wgs", "req ld MB", "req st MB", "fetched MB",
                                                   "written MB"))
    w(" | %10s %10s\n" % ("PMC fetch", "PMC write") if pmc else "\n")
    tot = collections.Counter()
    for k, a in agg.items():
        w("%-26s %8d %8d | %10.3f %10.3f | %10.3f %10.3f" % (k, a["launches"], a["wgs"], a["req_ld"] / 1e6,
                                                              a["req_st"] / 1e6, a["fetched"] / 1e6, a["written"] / 1e6))
        tot.update({x: a[x] for x in ("req_ld", "req_st", "fetched", "written")})
        if pmc:
            m = next((v for kk, v in pmc.items() if kk.rstrip("<") == k), None)
            w(" | %10.3f %10.3f\n" % (m["fetch_bytes"] / 1e6, m["write_bytes"] / 1e6) if m else " |\n")
        else:
            w("\n")
    w("%-26s %8s %8s | %10.3f %10.3f | %10.3f %10.3f" % ("total", "", "", tot["req_ld"] / 1e6, tot["req_st"] / 1e6,
                                                          tot["fetched"] / 1e6, tot["written"] / 1e6))
    if pmc:
        w(" | %10.3f %10.3f" % (sum(v["fetch_bytes"] for v in pmc.values()) / 1e6,
                                  sum(v["write_bytes"] for v in pmc.values()) / 1e6))
    w("\n")
    if top:
        for k in agg:
            rows = collections.Counter()
            for kk, pl in pcs.items():
                if short(kk) == k:
                    for loc, rl, rs, fe, wr in pl:
                        rows[loc] += fe + wr
            w("  %s: lines by fetched + written bytes\n" % k)
            det = collections.defaultdict(lambda: [0, 0, 0, 0])
            for kk, pl in pcs.items():
                if short(kk) == k:
                    for loc, rl, rs, fe, wr in pl:
                        d = det[loc]
                        d[0] += rl; d[1] += rs; d[2] += fe; d[3] += wr
            for loc, _ in rows.most_common(top):
                d = det[loc]
                w("    %-28s req ld %9.3f st %9.3f | fetched %9.3f written %9.3f MB\n" % (
                    loc, d[0] / 1e6, d[1] / 1e6, d[2] / 1e6, d[3] / 1e6))
    return tot


def att_bwd(ncent, top):
    """the Z2-free attention backward of an up layer (32 -> 128 over 5 edges per centre) in its shipped form and in the
    round-6 form that takes S1 / S2 from the forward's moments.  Emulated at TWO sizes: bytes = fixed + per-edge x E
    (the fixed part: per-workgroup partial tiles, their reduction, the operand staging), the per-edge slope next to the
    round-5 counters of the shipped form at cfg4's 3 276 800 edges"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_simt_train as T
    pmc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["att_bwd_noz_E3276800_32to128_detail"]
    sizes = (ncent, 2 * ncent)
    res = {}
    for nc in sizes:
        d = T._inputs(nc, 5, 11)
        args = [d[k] for k in ("Z1", "ps", "psh", "pm", "pr", "W2", "b2", "sc", "mu", "rs", "bsums", "amax", "gval")]
        gamma, beta = np.ones(128, np.float32), np.zeros(128, np.float32)
        _, _, mom = sim.att_bn2_moments(d["Z1"], d["ps"], d["psh"], d["W2"], d["b2"], gamma, beta)
        for name, kw in (("shipped: gridgcn_att_bwd_noz", {}), ("round 6, OPT.NOZ_BWD_MOMENTS: gridgcn_att_bwd_noz_mom", {"mom": mom})):
            start()
            sim.att_bwd_noz(*args, 5, **kw)
            ker, _ = report(per_pc=True)
            res[(name, nc)] = {short(k): v for k, v in ker.items()}
    print("# Z2-free attention backward 32 -> 128, 5 edges per centre, emulated at %d and %d edges: bytes = fixed + slope x E."
          % (5 * sizes[0], 5 * sizes[1]))
    print("# Algorithmic: 384 B per edge (Z1 128 read, dX 128 written, arg max + gradient 128 x 5 per 5 edges)."
          "  PMC = round-5 counters of the shipped form at 3 276 800 edges, per edge (profiles/traffic.json).")
    E0, E1 = 5 * sizes[0], 5 * sizes[1]
    for name in ("shipped: gridgcn_att_bwd_noz", "round 6, OPT.NOZ_BWD_MOMENTS: gridgcn_att_bwd_noz_mom"):
        print("## " + name)
        print("%-24s %5s | %14s %14s | %14s %14s | %9s %9s" % ("kernel", "wgs", "fetched B/edge", "fixed MB", "written B/edge",
                                                            "fixed MB", "PMC fetch", "PMC write"))
        tf = tw = ff = fw = 0.0
        for k in res[(name, sizes[0])]:
            a, b = res[(name, sizes[0])][k], res[(name, sizes[1])][k]
            sf = (b["fetched"] - a["fetched"]) / (E1 - E0)
            sw = (b["written"] - a["written"]) / (E1 - E0)
            xf = (a["fetched"] - sf * E0) / 1e6
            xw = (a["written"] - sw * E0) / 1e6
            m = next((v for kk, v in pmc.items() if kk == k or (k.startswith(kk) and kk == "gg_k_att_bwd_nz")), None)
            print("%-24s %5d | %14.1f %14.3f | %14.1f %14.3f | %9s %9s" % (
                k, b["wgs"], sf, xf, sw, xw, "%.1f" % (m["fetch_bytes"] / 3276800) if m else "",
                "%.1f" % (m["write_bytes"] / 3276800) if m else ""))
            tf += sf; tw += sw; ff += xf; fw += xw
        print("%-24s %5s | %14.1f %14.3f | %14.1f %14.3f |" % ("total", "", tf, ff, tw, fw))


def step():
    """one training step (forward + backward) of the segmentation net's 81 920-point layer family on the reduced grid of
    tests/test_simt_product.py (2 x 1024 points): every kernel of the step with the bytes it requests, fetches and
    writes.  At this size the per-launch constants (operand staging, partial tiles) dominate, so the table says nothing
    about the step's traffic at BASELINE size; what it is for is LINE UTILISATION -- a kernel that fetches much more than
    it requests reads 4-byte items out of 128-byte lines or re-reads the same lines from several XCDs"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import test_simt_product as P
    from grid_gcn_amd import model
    torch.manual_seed(0)
    cfg = P._tiny_seg_cfg(model.SEG_81920)
    data, npn = synth.make_batch(2, 1024, "planes")
    with emu.emulated_gpu(poison=False):
        net = model.GGCNSeg(cfg, fixed_seed=True).train()
        x, n = torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn)
        lab = torch.randint(1, 21, (2, 1024))
        start()
        model.seg_loss(net(x, n), lab).backward()
        ker, _ = report()
    agg = collections.OrderedDict()
    for k, v in ker.items():
        agg.setdefault(short(k), collections.Counter()).update(v)
    print("# one training step of GGCNSeg (81 920-point layer family) at 2 x 1024 points on the emulator: MB per step, per kernel;")
    print("# f/r = fetched / requested loads, w/r = written / requested stores.  Line utilisation only -- see tools/simt_traffic.py: step")
    print("%-30s %5s | %9s %9s | %9s %9s | %6s %6s" % ("kernel", "n", "req ld", "req st", "fetched", "written", "f/r", "w/r"))
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["fetched"] + kv[1]["written"])):
        print("%-30s %5d | %9.3f %9.3f | %9.3f %9.3f | %6.2f %6.2f" % (
            k, a["launches"], a["req_ld"] / 1e6, a["req_st"] / 1e6, a["fetched"] / 1e6, a["written"] / 1e6,
            a["fetched"] / max(a["req_ld"], 1), a["written"] / max(a["req_st"], 1)))


def linear_fwd(E):
    """the 256 -> 128 forward conv of the per-point update layer (gridgcn_linear_fwd_direct: the BatchNorm + ReLU of the
    layer in front applied while loading, statistics epilogue) at E and 2 E rows: bytes per ROW next to the round-5
    counters at cfg4's 655 360 rows (profiles/traffic.json: linear_fwd_E655360_256to128).  Algorithmic: 1024 B read
    (256 fp32 inputs) + 512 B written (128 outputs) per row."""
    import torch
    from grid_gcn_amd.train import timers as ttimers
    pmc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["linear_fwd_E655360_256to128_detail"]
    res = {}
    with emu.emulated_gpu(poison=False):
        torch.cuda.Event = type("E", (emu._Event,), {"elapsed_time": lambda self, o: 1.0})
        for rows in (E, 2 * E):
            start()
            ttimers.time_linear_fwd(rows, 256, 128, iters=1, device="cpu")
            ker, _ = report()
            k = next(v for kk, v in ker.items() if kk.startswith("gg_k_linear_fwd_direct"))
            res[rows] = k
    a, b = res[E], res[2 * E]
    print("# gridgcn_linear_fwd_direct 256 -> 128, emulated at %d and %d rows (%d / %d launches of the kernel: warm-up + timed):"
          " bytes per ROW and launch" % (E, 2 * E, a["launches"], b["launches"]))
    for name, key in (("fetched", "fetched"), ("written", "written"), ("requested loads", "req_ld")):
        pa, pb = a[key] / a["launches"], b[key] / b["launches"]
        slope = (pb - pa) / E
        print("%-16s %8.1f B per row + %.3f MB per launch" % (name, slope, (pa - slope * E) / 1e6))
    m = next(iter(pmc.values()))
    print("round-5 PMC at 655 360 rows: fetch %.1f B per row, write %.1f B per row" % (m["fetch_bytes"] / 655360, m["write_bytes"] / 655360))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gridify", "att_bwd", "step", "linear_fwd"])
    ap.add_argument("cfg", nargs="?", default="cfg4")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--layer", type=int, default=0)
    ap.add_argument("--lines", type=int, default=0, help="source lines per kernel")
    ap.add_argument("--option", action="append", default=[], help="library option NAME=VALUE (gridgcn_set_option)")
    ap.add_argument("--ncent", type=int, default=12000)
    a = ap.parse_args()
    if a.what == "att_bwd":
        return att_bwd(a.ncent, a.lines)
    if a.what == "step":
        return step()
    if a.what == "linear_fwd":
        return linear_fwd(a.ncent)
    grids = {"cfg4": (synth.SEG_SCANNET_81920, 81920, 8, "planes", "gridify_N81920_B8"),
             "cfg3": (synth.SEG_SCANNET_8192, 8192, 16, "planes", None),
             "cfg5": (synth.SYNTH_200K, 200000, 8, "planes", "gridify_N200000_B8")}
    g, N, B, kind, key = grids[a.cfg]
    B = a.batch or B
    from grid_gcn_amd import _lib
    for o in a.option:
        name, val = o.split("=")
        sim.set_option(getattr(_lib, "OPT_" + name), int(val))
    data, npn = synth.make_batch(B, N, kind)
    data, npn = np.asarray(data), np.asarray(npn)
    kw = synth.gridify_kwargs(g, a.layer, 0)
    t0 = time.time()
    start()
    sim.Gridify(data, npn, **kw)
    ker, pcs = report(per_pc=a.lines > 0)
    alg = B * synth.gridify_algorithmic_bytes(N, kw["max_o_grid"], kw["max_p_grid"])
    pmc = None
    if key and a.layer == 0 and B == grids[a.cfg][2]:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key + "_detail")
    print("# Gridify %s layer %d: B = %d, N = %d, O = %d, P = %d, grid %s; algorithmic bytes (SURVEY 8d) %.3f MB; emulated in %.0f s"
          % (a.cfg, a.layer, B, N, kw["max_o_grid"], kw["max_p_grid"], tuple(kw["grid_size"]), alg / 1e6, time.time() - t0))
    if a.option:
        print("# options: " + " ".join(a.option))
    tot = table(ker, pcs, pmc, a.lines)
    print("# fetched + written = %.3f MB = %.2f x algorithmic" % ((tot["fetched"] + tot["written"]) / 1e6,
                                                                   (tot["fetched"] + tot["written"]) / alg))


if __name__ == "__main__":
    main()
