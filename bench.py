"""bench.py -- BASELINE.json metric on MI355X: point-clouds/s fwd+bwd of the ScanNet 81920-pt
segmentation network (configs[3]: batch 8 per GPU, data parallel), plus ms per CAGQ layer
(= one Gridify call) with its HBM roofline, and the CPU baseline timed beside it.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config cfg1|cfg2|cfg3|cfg3up|cfg5      (the other BASELINE.json configs)

A step = zero_grad, forward (3 Gridify + 3 BallKNN + 6 gathers + 6 GridConv layers + head),
loss, backward, one flat RCCL all-reduce of the gradients (N > 1), Adam update.  Inputs are
synthetic clouds already resident in HBM (no dataset offline); weights are Xavier random.

Every number in the JSON line is measured in this run, except `traffic` (HBM bytes per launch from
the PMC counters): that one is read from profiles/traffic.json, written by tools/pmc_traffic.py from
separate rocprofv3 --pmc passes of the same kernels at the same shapes; null if no entry matches.
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL's peer-to-peer setup fails
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from grid_gcn_amd import dp, graph, model, model_cls, ops, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32, dense


def load_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def cpu_baseline(cfg, B, points, kind, sample_clouds):
    """Reference-side numbers on the host cores.  The S0 oracle (C, OpenMP
    across clouds) for the index ops, PyTorch-CPU for gather/GridConv.  The CAGQ layer is timed
    on the SAME batch as the GPU; the whole fwd+bwd step on a bounded sample of that batch."""
    from oracle import oracle as orc
    from oracle.torch_index_ops import OracleIndexOps
    data, npn = synth.make_batch(B, points, kind)
    kw = synth.gridify_kwargs(cfg["grid"], 0)
    orc.gridify(data, npn, **kw)                              # builds / loads the oracle, starts the team
    t1 = time.perf_counter()
    for _ in range(3):
        orc.gridify(data, npn, **kw)
    dt_g = (time.perf_counter() - t1) / 3
    cagq_threads = orc.threads_for(B)
    torch.manual_seed(0)
    m = model.GGCNSeg(cfg, index_ops=OracleIndexOps)
    m.train()
    S = min(B, sample_clouds)
    x = torch.from_numpy(data[:S, :, :3].copy())
    n = torch.from_numpy(npn[:S])
    lab = torch.randint(0, cfg["num_classes"], (S, points))
    dts = []
    for _ in range(2):            # the first pass pays thread-pool start-up, page faults, oneDNN set-up
        for prm in m.parameters():
            prm.grad = None
        t0 = time.perf_counter()
        loss = model.seg_loss(m(x, n), lab)
        loss.backward()
        dts.append(time.perf_counter() - t0)
    dt = dts[1]
    return {"value": S / dt, "unit": "point-clouds/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d of the %d clouds x %d pts, fwd+bwd twice, the SECOND pass reported (first: "
                      "%.2f s): S0 oracle (C, OpenMP across clouds) for Gridify/BallKNN + PyTorch-CPU "
                      "(%d threads) for gather/GridConv" % (S, B, points, dts[0],
                                                            torch.get_num_threads()),
            "ms_per_cagq_layer": dt_g * 1e3,
            "cagq_threads": cagq_threads,
            "cagq_sample": "Gridify down layer 0 on the same %d-cloud batch, OpenMP across clouds "
                           "(%d threads), mean of 3 calls" % (B, cagq_threads)}


_PROBE = r"""
import os, sys, torch, torch.distributed as dist
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world)
x = torch.full((1 << 16,), float(rank + 1), device="cuda")
y = torch.zeros_like(x)
dist.all_reduce(y)                                   # communicator set up outside the capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y.copy_(x)
    dist.all_reduce(y)
    y.div_(world)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
want = sum(range(1, world + 1)) / world
ok = bool((y == want).all())
dist.barrier()
dist.destroy_process_group()
print("PROBE_OK" if ok else "PROBE_WRONG")
"""


def rccl_capture_probe(world, rank, local, dev):
    """Can an RCCL all-reduce be captured into a hipGraph and replayed on this node?  Asked in a
    throw-away process per rank (its own process group on port MASTER_PORT + 17): a failed stream
    capture poisons the HIP context of the process it happens in, so the real step must only try
    what is known to work.  The ranks agree on the answer (MIN over ranks)."""
    import subprocess
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local),
               MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
               MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 17))
    for k in list(env):
        if k.startswith("TORCHELASTIC") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "GROUP_WORLD_SIZE",
                                                  "ROLE_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING"):
            env.pop(k)
    try:
        r = subprocess.run([sys.executable, "-c", _PROBE], env=env, capture_output=True, text=True,
                           timeout=240)
        ok = r.returncode == 0 and "PROBE_OK" in r.stdout
        if not ok and rank == 0:
            sys.stderr.write("RCCL capture probe failed (rc %d): %s\n" % (r.returncode, r.stderr[-400:]))
    except Exception as e:  # noqa: BLE001
        ok = False
        if rank == 0:
            sys.stderr.write("RCCL capture probe failed (%s)\n" % e)
    if world > 1 and dist.is_initialized():
        t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(int(t.item()))
    return ok


def make_opt(net, eager):
    """Adam(lr 1e-3, wd 1e-5): the one-launch kernel of grid_gcn_amd.optim (torch.optim.Adam's update to
    within rounding; tests/test_gpu_glue.py), or -- --switch OWN_ADAM=0 -- the framework's multi-tensor one"""
    from grid_gcn_amd import optim
    from grid_gcn_amd.train.options import OPT
    if OPT.OWN_ADAM:
        return optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    return torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, fused=True, capturable=not eager)


def make_step(net, opt, sync, loss_fn, inputs, target, use_graph):
    """The timed step.  Default: one captured hipGraph (for N > 1 with the flat RCCL all-reduce
    inside, when rccl_capture_probe says a captured collective replays correctly here) --
    grid_gcn_amd/graph.py -- so that the result does not depend on how fast the host enqueues ~350
    launches; the GPU work of a replay is that of the eager step, launch for launch, with the
    random draws still fresh per step (device-side seed).  Falls back to the eager step when the
    capture fails; `step_mode` in the JSON line says which one was timed."""
    one = torch.ones((), dtype=torch.float32, device=inputs[0].device)

    def eager():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(net(*inputs), target)
        loss.backward(one)                    # (the default would launch a fill for the same scalar)
        sync()
        opt.step()
        return loss

    make_step.probe = None
    if use_graph and sync is not None and sync.world > 1:
        # N > 1: the graph holds the RCCL all-reduce (graph.py); only attempted where a throw-away
        # process has shown that such a capture replays correctly
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if dist.get_backend() != "nccl":
            make_step.probe = "not asked (backend %s)" % dist.get_backend()
            use_graph = False
        else:
            ok = rccl_capture_probe(sync.world, dist.get_rank(), local, inputs[0].device)
            make_step.probe = "ok" if ok else "failed"
            use_graph = ok
        sys.stderr.write("[rank %d] RCCL capture probe: %s -> %s step\n" % (
            dist.get_rank(), make_step.probe, "hipgraph" if use_graph else "eager"))
    if use_graph:
        try:
            return graph.GraphedTrainStep(net, opt, loss_fn, inputs, target, sync), "hipgraph"
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            sys.stderr.write("graph capture failed (%s: %s); timing the eager step\n" % (type(e).__name__, e))
            net.seed_dev = None
            if (sync is None or sync.world == 1) and "--eager" not in sys.argv:
                # a failed stream capture leaves the HIP context of this process unusable
                # ("operation failed due to a previous error during capture"): start over, eager
                sys.stderr.flush()
                os.execv(sys.executable, [sys.executable] + sys.argv + ["--eager"])
    return eager, "eager"


def ranks_agree(what, value, world, dev):
    """N > 1: every rank must time the SAME kind of step.  One rank silently on the eager fallback (its capture
    failed, its probe disagreed) would read as a scaling loss in SCALE_r*.json; this makes it fatal on all ranks
    instead.  `value` is a short string; compared through its bytes' MIN and MAX over the ranks."""
    if world <= 1 or not dist.is_initialized():
        return
    b = (value or "").encode()[:32].ljust(32, b"\0")
    t = torch.tensor(list(b), dtype=torch.int32, device=dev)
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not bool((lo == hi).all()):
        raise RuntimeError("[rank %d] %s differs across ranks (this rank: %r): refusing to time a mixed "
                           "hipgraph / eager job" % (dist.get_rank(), what, value))


def time_training(step, steps, warmup, world, dev):
    """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks.
    Also returns the host time needed to ENQUEUE the K steps (before the final synchronize)."""
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss).item()
    return dt, t_enq


def time_allreduce(sync, world, dev, iters=20):
    """The gradient collective alone: median ms of one all-reduce of the flat bucket (what the step adds per
    rank count), with what the process group itself reports -- so that a scaling record can be audited."""
    out = {"allreduce_ms": None, "flat_bucket_bytes": int(sync.flat.numel() * sync.flat.element_size()),
           "dist_world_size": dist.get_world_size() if dist.is_initialized() else 1,
           "dist_backend": dist.get_backend() if dist.is_initialized() else None}
    if world > 1:
        buf = torch.zeros_like(sync.flat)
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for e0, e1 in ev:
            e0.record()
            dist.all_reduce(buf)
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        t = torch.tensor([ts[len(ts) // 2]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["allreduce_ms"] = float(t.item())
    return out


def param_sync_spread(net, world, dev):
    """max - min over the ranks of sum(parameters) after the timed steps: 0.0 when every rank applied the
    same all-reduced gradients to the same broadcast start (what data parallelism promises)."""
    with torch.no_grad():
        s = torch.zeros(1, dtype=torch.float64, device=dev)
        for prm in net.parameters():
            s += prm.double().sum()
    if world > 1:
        hi, lo = s.clone(), s.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        return float((hi - lo).item())
    return 0.0


def in_step_ms(net, loss_fn, inputs, target, keys, steps=4):
    """Device time of selected library calls inside EAGER training steps (forward + loss + backward; the
    kernels behind their real predecessors, real tensors, cold L2): tcommon.LaunchTimers.  The first step
    is dropped."""
    from grid_gcn_amd.train import common as tcommon
    from grid_gcn_amd.train.options import OPT
    OPT.TIMERS = tcommon.LaunchTimers(keys)
    try:
        for _ in range(steps):
            for prm in net.parameters():
                prm.grad = None
            loss_fn(net(*inputs), target).backward()
        torch.cuda.synchronize()
        per_step = {k: len(v) // steps for k, v in OPT.TIMERS.ev.items()}
        return {k: OPT.TIMERS.median(k, skip=per_step[k]) for k in keys}, per_step
    finally:
        OPT.TIMERS = None


def cagq_roofline(d4, n, kw, B, N, traffic, key, iters=100):
    ms, _ = ops.gridify_timed(d4, n, iters, **kw)
    alg = B * synth.gridify_algorithmic_bytes(N, kw["max_o_grid"], kw["max_p_grid"])
    ach = alg / (ms * 1e-3) / 1e9
    # (clouds of <= 4096 points take the one-launch build: csrc/gridgcn_index.hip, GRIDGCN_OPT_INDEX_SMALL)
    from grid_gcn_amd import _lib as _glib
    small = N <= 4096 and kw["max_o_grid"] <= 4096 and _glib.load().gridgcn_get_option(_glib.OPT_INDEX_SMALL) != 0
    build = "gg_k_small_build" if small else "gg_k_chunk_split + gg_k_slab_build + gg_k_centre_slots"
    return ms, {"bound": "hbm",
                "kernel": "gridgcn_gridify (%s + gg_k_query_gridify; %d back-to-back calls between two HIP events on "
                          "the launch stream)" % (build, iters),
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic.get(key), "traffic_key": key,
                "algorithmic_bytes_per_launch": alg, "ms_per_launch": ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg4",
                    choices=["cfg1", "cfg2", "cfg3", "cfg3up", "cfg4", "cfg5"],
                    help="BASELINE.json configs[0..4]; cfg4 (configs[3]) carries the headline metric")
    ap.add_argument("--batch", type=int, default=0, help="clouds per GPU (0: the config's own)")
    ap.add_argument("--points", type=int, default=0, help="points per cloud (0: the config's own)")
    ap.add_argument("--cpu-sample", type=int, default=1, help="clouds of the CPU fwd+bwd sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--micro-iters", type=int, default=50,
                    help="launches per micro-benchmark (median reported); the PMC passes use 3")
    ap.add_argument("--no-micro", action="store_true",
                    help="only the timed step: no per-kernel rooflines, no inference, no CPU baseline "
                         "(what the PMC passes of tools/pmc_step.sh run)")
    ap.add_argument("--eager", action="store_true", help="time the eager step instead of the hipGraph replay")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="contraction precision of the training GEMM kernels: f32 = exact fp32 MFMA "
                         "(parity path, the headline), bf16 = bf16 MFMA operands with fp32 storage, "
                         "accumulation and statistics (BASELINE configs[2] 'bf16 MLP / fp32 indices')")
    ap.add_argument("--switch", action="append", default=[],
                    help="NAME=0|1: a path switch of grid_gcn_amd.train.options.OPT (e.g. NOZ_ATT_BWD=0) or a "
                         "gridgcn_set_option name (COL_SPLIT, INDEX_SMALL, ATT_BWD_FUSED) for A/B measurements; "
                         "the defaults are what is shipped and reported")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (GG_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than
    #  ranks -- several ranks then share a device; never used for reported numbers)
    backend = os.environ.get("GG_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local if backend == "nccl" else local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    traffic = load_traffic()
    from grid_gcn_amd.train import common as tcommon, evalpath as teval, timers as ttimers
    from grid_gcn_amd.train.options import OPT
    tcommon.set_mlp_precision("bf16" if a.dtype == "bf16" else "fp32")
    for sw in a.switch:
        name, val = sw.split("=")
        from grid_gcn_amd import _lib as _glib
        if hasattr(_glib, "OPT_" + name):       # a kernel-selection option of the library (gridgcn_set_option)
            _glib.check(_glib.load().gridgcn_set_option(getattr(_glib, "OPT_" + name), int(val)), "set_option")
            continue
        OPT.set(name, val)               # (KeyError on an unknown name)

    if a.config != "cfg4":
        import bench_configs
        out = bench_configs.run(a, world, rank, dev, traffic, time_training, cagq_roofline, make_step,
                                time_allreduce, param_sync_spread, make_opt)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            import gc
            gc.collect()
            torch.cuda.synchronize()
            dist.barrier()
            dist.destroy_process_group()
        return

    points = a.points or 81920
    B = a.batch or 8
    cfg = model.SEG_81920 if points > 8192 else model.SEG_8192
    kind = "planes"
    torch.manual_seed(0)
    net = model.GGCNSeg(cfg, seed=rank).to(dev)
    net.train()
    opt = make_opt(net, a.eager)
    sync = dp.FlatGradAllReduce(net)
    sync.broadcast_parameters()
    data, npn = synth.make_batch(B, points, kind, first_id=rank * B)   # a different shard per rank
    x = torch.from_numpy(data[..., :3].copy()).to(dev)
    n = torch.from_numpy(npn).to(dev)
    lab = torch.randint(0, cfg["num_classes"], (B, points), device=dev)
    step, step_mode = make_step(net, opt, sync, model.seg_loss, (x, n), lab, not a.eager)
    ranks_agree("step_mode", step_mode, world, dev)
    ranks_agree("rccl_capture_probe", getattr(make_step, "probe", None), world, dev)

    dt, t_enq = time_training(step, a.steps, a.warmup, world, dev)
    ms_step = dt / a.steps * 1e3
    allreduce = time_allreduce(sync, world, dev)
    allreduce["param_sync_spread"] = param_sync_spread(net, world, dev)
    fe, fr = model.seg_forward_flops(net, B, points)
    step_flops = 3.0 * (fe + fr)
    tf_step = step_flops / (ms_step * 1e-3) / 1e12

    out = {
        "metric": "point-clouds/sec fwd+bwd (ScanNet 81920-pt)", "value": world * B * a.steps / dt,
        "unit": "point-clouds/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic", "step_mode": step_mode,
        "rccl_capture_probe": getattr(make_step, "probe", None),
        "config": {"workload": "BASELINE configs[3]: ScanNet %d-pt segmentation, batch %d per GPU, "
                               "3 Gridify down + 3 BallKNN up layers, Adam, fp32" % (points, B),
                   "global_batch": world * B, "points_per_cloud": points,
                   "parallelism": "dp%d" % world,
                   "kernels": "hand-written HIP for Gridify/BallKNN, edge inputs (gather+geo) and their "
                              "sorted backward, all conv+BatchNorm+ReLU stacks fwd+bwd (fp32 MFMA), "
                              "att product + max, fc1 + dropout + class scores (one op) + softmax "
                              "cross-entropy, the small GEMMs on the source points, concat + centre mask of "
                              "the layer boundaries, Adam (one launch); PyTorch-ROCm for device memory, "
                              "streams, autograd bookkeeping and the RCCL call"},
        # host side of the timed region: time to ENQUEUE the K steps (Python + ctypes + launches);
        # the step is GPU-bound while this stays below ms_per_step
        "host_enqueue_ms_per_step": t_enq / a.steps * 1e3,
        # the gradient collective alone and what the process group reports (N > 1; null on one rank)
        "allreduce_ms": allreduce["allreduce_ms"], "flat_bucket_bytes": allreduce["flat_bucket_bytes"],
        "dist_world_size": allreduce["dist_world_size"], "dist_backend": allreduce["dist_backend"],
        "param_sync_spread": allreduce["param_sync_spread"],
        # whole step against the fp32 matrix peak: SURVEY section 8(d) algorithmic flops of the step
        # (3 x forward: per-edge MLPs + per-point MLPs + head) / ms_per_step
        "roofline_step": {"bound": "mfma", "kernel": "whole training step (all kernels)",
                          "achieved": tf_step, "peak": MFMA_F32_PEAK_TF if a.dtype == "f32" else 2500.0,
                          "unit": "TFLOP/s",
                          "frac": tf_step / (MFMA_F32_PEAK_TF if a.dtype == "f32" else 2500.0),
                          "traffic": None, "mfma_busy": None,
                          "algorithmic_flops_per_step": step_flops,
                          "edge_flops_fwd": fe, "per_point_flops_fwd": fr},
    }

    # whole-step HBM bytes and MFMA-pipe utilisation from the PMC passes over one eager step
    # (tools/pmc_step.sh -> profiles/traffic.json), f32 only
    if points == 81920 and B == 8:
        # (fp32 and bf16 steps have PMC passes of their own: tools/pmc_step.sh [bf16])
        skey = "step_cfg4" if a.dtype == "f32" else "step_cfg4_bf16"
        out["roofline_step"]["traffic"] = traffic.get(skey)
        out["roofline_step"]["mfma_busy"] = traffic.get(skey + "_mfma_busy")
        out["roofline_step"]["traffic_key"] = skey
        # the step's three roofs in one place: `frac` = algorithmic flops of the REFERENCE's graph
        # / time / matrix peak (a speed figure, not a utilisation), `mfma_busy` = the counter, `hbm_frac` = looked-up
        # PMC bytes of one step / THIS run's ms_per_step / 8 TB/s
        if traffic.get(skey):
            out["roofline_step"]["hbm_frac"] = traffic[skey] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline_step"]["hbm_frac_note"] = ("traffic is a look-up in profiles/traffic.json (PMC pass of "
                                                     "another session), the time is this run's")
    if rank == 0 and world == 1 and not a.no_micro:
        from grid_gcn_amd.train.timers import median_ms
        # ---- ms per CAGQ layer: Gridify of down layer 0 on the same batch ----
        kw = synth.gridify_kwargs(cfg["grid"], 0)
        d4 = torch.from_numpy(data).to(dev)
        mi = a.micro_iters
        ms, rc = cagq_roofline(d4, n, kw, B, points, traffic, "gridify_N%d_B%d" % (points, B),
                               iters=2 * mi)
        out["ms_per_cagq_layer"] = ms
        out["roofline_cagq"] = rc
        # ---- inference forward through the fused GridConv kernels (the reference's own speed
        #      recipe times inference: train_gpu_speed_profiling.py:105-118) + the dominant
        #      hand-written kernel of the path: gg_k_gridconv of up layer 2 (fp32 MFMA bound) ----
        net.eval()
        with torch.no_grad():
            net.jobs = []
            net(x, n)
            jobs, net.jobs = net.jobs, None
            ms_inf = median_ms(lambda: net(x, n), max(3, mi // 2), 3, dev)
            name, layer, cent_, src_, idx_ = max(jobs, key=lambda j: j[4].numel() * j[1].cin)
            pt, att = layer.packed_layers()
            src_ = src_.contiguous()
            call = lambda: ops.gridconv_forward(src_, idx_, cent_, pt, att,  # noqa: E731
                                                has_feats=layer.has_feats,
                                                localfdim=layer.localfdim)
            ms_k = median_ms(call, mi, 5, dev)
            # the path evaluation actually takes for this layer (one point conv: source-side)
            ms_src = None
            from grid_gcn_amd.train import common as tcommon, evalpath as teval, timers as ttimers
            from grid_gcn_amd.train.options import OPT
            attl, ptl = [layer.att1[0], layer.att2[0]], list(layer.pt_mlp)
            if teval.edge_block_src_eval_supported(ptl, attl, src_, layer.has_feats):
                call2 = lambda: teval.edge_block_src_eval(src_, idx_, cent_.contiguous(), ptl[0],  # noqa: E731
                                                        attl, layer.localfdim)
                ms_src = median_ms(call2, mi, 5, dev)
        macs = sum(l.lin.in_features * l.lin.out_features
                   for seq in (layer.pt_mlp, layer.att1, layer.att2) for l in seq)
        flops = 2.0 * idx_.numel() * macs
        tf = flops / (ms_k * 1e-3) / 1e12
        out["roofline_inference"] = {
            "bound": "mfma", "kernel": "gg_k_gridconv (GridConv %s: gather + per-edge MLPs + att "
            "product + max, one launch, inference-mode BatchNorm)" % name,
            "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
            "frac": tf / MFMA_F32_PEAK_TF,
            "traffic": traffic.get("gridconv_%s_E%d" % (name, idx_.numel())),
            "traffic_key": "gridconv_%s_E%d" % (name, idx_.numel()),
            "algorithmic_flops_per_launch": flops, "ms_per_launch": ms_k,
            "dtype": "f32 (v_mfma_f32_32x32x2_f32)",
            # evaluation runs this layer through the source-side kernels instead (first conv once
            # per source point, gathered by the max kernel): same result, fewer executed flops
            "ms_source_side_path": ms_src,
            "note": "kernel-level figure for gg_k_gridconv on this layer's shape; the evaluation "
                    "forward itself runs up layers through ms_source_side_path (source-side conv + "
                    "gridgcn_att_max_eval) and uses gg_k_gridconv for the down layers"}
        # ---- dominant kernel of the TIMED training step.  The point conv of this layer runs on the source
        #      points (gridgcn_edgelin.hip), so the largest per-edge GEMMs left are those of the attention
        #      MLP.  Backward of its C/4 -> C conv WITHOUT the [E, C] pre-activation (gridgcn_att_bwd_noz,
        #      csrc/gridgcn_attbwd_nz.hip: the arg-max term by MFMA from the sparse gradient, the dense
        #      BatchNorm term through W2^T diag(bz) W2 and the moments of the C/4-wide activation; dA1,
        #      the BatchNorm-backward sums of the layer in front and dW2 from one read of Z1) is the longest
        #      single launch of the step.  Its algorithmic traffic (Z1 in, dA1 out, the [ncent, C] sparse
        #      gradient) is 1.3 GB -- 0.2 ms at HBM peak -- so the roof that binds it is the fp32 MFMA pipe,
        #      which its VALU work shares. ----
        from grid_gcn_amd.train import common as tcommon, evalpath as teval, timers as ttimers
        from grid_gcn_amd.train.options import OPT
        cin_b = layer.att2[0].lin.in_features
        c_b = layer.att2[0].lin.out_features
        ncent_b, p_b = idx_.shape[0] * idx_.shape[1], idx_.shape[2]
        e_b = float(ncent_b * p_b)
        # (bf16 mode takes the same fp32 pair of Z2-free kernels for this layer: OPT.NOZ_IN_BF16)
        noz = (OPT.NOZ_ATT_BWD and cin_b == 32 and c_b == 128 and p_b == 5
               and (a.dtype == "f32" or (OPT.NOZ_IN_BF16 and OPT.NOZ_ATT_FWD)))
        if noz:
            ms_b = ttimers.time_att_bwd_noz(ncent_b, p_b, cin_b, c_b, iters=mi, device=dev)
            # read Z1 [E,cin], the sparse upstream gradient (one-byte amax + fp32 value) [ncent,C]; write dA1
            bytes_b = 4.0 * e_b * 2 * cin_b + 5.0 * ncent_b * c_b
        else:
            ms_b = ttimers.time_linear_bwd(ncent_b, p_b, cin_b, c_b, iters=mi, device=dev,
                                             ndx=cin_b, prev_bn=True)
            # read Z [E,C] once, the sparse upstream gradient [ncent,C], the previous layer's raw output
            # [E,cin]; write dX [E,cin].  The micro-benchmark reads an fp32 Z; inside a bf16-mode step this
            # tensor is STORED as bf16 (OPT.Z16_STORAGE): the in-step figure counts it at that width (VERDICT r4:
            # counted at 4 bytes the line claimed 6.9 TB/s, above what the part delivers)
            bytes_b = 4.0 * e_b * (c_b + 2 * cin_b) + 5.0 * ncent_b * c_b
            z16 = a.dtype == "bf16" and OPT.Z16_STORAGE and c_b in (64, 128)
            bytes_b_step = bytes_b - (2.0 * e_b * c_b if z16 else 0.0)
        flops_b = 4.0 * e_b * cin_b * c_b            # dX + dW products
        # ... and the same calls timed INSIDE eager training steps (their real predecessors and tensors):
        # `frac` is what the step pays, `frac_micro` the back-to-back micro-benchmark
        e_f, cin_f, c_f = B * points, 256, 128
        kb_, kf_ = ("linear_bwd", int(e_b), cin_b, c_b), ("linear_fwd", e_f, cin_f, c_f)
        net.train()
        instep, per_step = in_step_ms(net, model.seg_loss, (x, n), lab, [kb_, kf_])
        net.eval()
        ms_b_step = instep[kb_] or ms_b
        if noz:
            tf_b = flops_b / (ms_b_step * 1e-3) / 1e12
            key_b = "att_bwd_noz_E%d_%dto%d" % (int(e_b), cin_b, c_b)
            out["roofline"] = {"bound": "mfma", "kernel": "gg_k_att_bwd_nz + gg_k_att_nz_reduce + gg_k_att_nz_finish "
                               "(backward of the %d->%d attention conv of GridConv %s over %d edges without "
                               "its [E,%d] pre-activation: dX, BN-backward sums of the layer in front and dW "
                               "from one read of the %d-wide activation)" % (cin_b, c_b, name, ncent_b * p_b,
                                                                            c_b, cin_b),
                               "achieved": tf_b, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                               "frac": tf_b / MFMA_F32_PEAK_TF,
                               "traffic": traffic.get(key_b), "traffic_key": key_b,
                               "algorithmic_flops_per_launch": flops_b, "algorithmic_bytes_per_launch": bytes_b,
                               "hbm_frac": bytes_b / (ms_b_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "ms_per_launch": ms_b, "ms_in_step": instep[kb_],
                               "launches_per_step": per_step[kb_],
                               "frac_micro": flops_b / (ms_b * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                               "timing": "frac / achieved from ms_in_step (HIP events around the call inside "
                                         "eager training steps, median); ms_per_launch = back-to-back "
                                         "micro-benchmark",
                               "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}
        else:
            gbs_b = (bytes_b_step if instep[kb_] else bytes_b) / (ms_b_step * 1e-3) / 1e9
            key_b = "att_bwd_fused_E%d_%dto%d" % (int(e_b), cin_b, c_b)
            out["roofline"] = {"bound": "hbm", "kernel": "gg_k_att_bwd_fused + gg_k_att_dw_reduce "
                               "(fused backward of the %d->%d attention conv of GridConv %s over %d "
                               "edges: BN/ReLU backward formed in registers from the sparse arg-max "
                               "gradient; dX, BN-backward sums of the layer in front and dW in one pass "
                               "over Z)" % (cin_b, c_b, name, ncent_b * p_b),
                               "achieved": gbs_b, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": gbs_b / HBM_PEAK_GBS,
                               "traffic": traffic.get(key_b), "traffic_key": key_b,
                               "algorithmic_bytes_per_launch": bytes_b_step if instep[kb_] else bytes_b,
                               "algorithmic_bytes_micro": bytes_b, "ms_per_launch": ms_b,
                               "ms_in_step": instep[kb_], "launches_per_step": per_step[kb_],
                               "frac_micro": bytes_b / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "timing": "frac / achieved from ms_in_step (HIP events around the call inside eager "
                                         "training steps, median); ms_per_launch = back-to-back micro-benchmark",
                               "algorithmic_flops_per_launch": flops_b,
                               "dtype": "f32 (v_mfma_f32_32x32x2_f32)" if a.dtype == "f32" else "bf16 MFMA operands"}
        # the largest MFMA-bound kernel of the step: forward of the 256->128 update conv over
        # all B*N points (previous BatchNorm+ReLU applied while loading, statistics epilogue)
        ms_f = ttimers.time_linear_fwd(e_f, cin_f, c_f, iters=mi, device=dev)
        ms_f_step = instep[kf_] or ms_f
        tf_f = 2.0 * e_f * cin_f * c_f / (ms_f_step * 1e-3) / 1e12
        # (bf16 mode: this layer sits behind a BatchNorm + ReLU and runs on v_mfma_f32_32x32x16_bf16 -- priced against
        #  THAT pipe's dense peak; against the fp32 peak the line read 1.16)
        peak_f = MFMA_F32_PEAK_TF if a.dtype == "f32" else 2500.0
        out["roofline_mfma"] = {"bound": "mfma", "kernel": "gg_k_linear_fwd_direct (%d->%d conv + "
                                "BN/ReLU prologue + statistics over %d rows)" % (cin_f, c_f, e_f),
                                "achieved": tf_f, "peak": peak_f, "unit": "TFLOP/s",
                                "frac": tf_f / peak_f,
                                "traffic": traffic.get("linear_fwd_E%d_%dto%d" % (e_f, cin_f, c_f)),
                                "traffic_key": "linear_fwd_E%d_%dto%d" % (e_f, cin_f, c_f),
                                "algorithmic_flops_per_launch": 2.0 * e_f * cin_f * c_f,
                                "ms_per_launch": ms_f, "ms_in_step": instep[kf_],
                                "launches_per_step": per_step[kf_],
                                "frac_micro": 2.0 * e_f * cin_f * c_f / (ms_f * 1e-3) / 1e12 / peak_f,
                                "dtype": "f32 (v_mfma_f32_32x32x2_f32)" if a.dtype == "f32"
                                else "bf16 operands, fp32 storage (v_mfma_f32_32x32x16_bf16; HBM bound: 1.0 GB per launch)"}
        # ---- the materialising neighbour gather as an operator (SURVEY §8(d) algorithmic bytes):
        #      batch_take_g forward + its sorted backward at the shape of layer up2 ----
        with torch.no_grad():
            gsrc = src_.contiguous()
            tk = lambda: ops.batch_take_g(gsrc, idx_, neighbour_index=True)  # noqa: E731
            ms_g = median_ms(tk, mi, 5, dev)
            gout = tk()
            tb = lambda: ops.batch_take_g_backward(gout, idx_, gsrc.shape[1], True)  # noqa: E731
            ms_gb = median_ms(tb, mi, 5, dev)
        alg_g = 4.0 * gsrc.numel() + 4.0 * idx_.numel() + 4.0 * idx_.numel() * gsrc.shape[2]
        out["roofline_gather"] = {
            "bound": "hbm", "kernel": "gridgcn_batch_take -- OPERATOR-BOUNDARY MICRO-BENCHMARK, not launched by the "
            "timed step (the step's edge kernels gather inside themselves): the reference's batch_take_g of GridConv "
            "%s, src %s, index %s" % (name, list(gsrc.shape), list(idx_.shape)),
            "achieved": alg_g / (ms_g * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_g / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic.get("batch_take_%s_E%d" % (name, idx_.numel())),
            "traffic_key": "batch_take_%s_E%d" % (name, idx_.numel()),
            "algorithmic_bytes_per_launch": alg_g, "ms_per_launch": ms_g,
            "backward_sorted_ms": ms_gb,
            "backward_sorted_GBps": alg_g / (ms_gb * 1e-3) / 1e9}
        del gout
        out["inference"] = {"value": B / (ms_inf * 1e-3), "unit": "point-clouds/s",
                            "ms_per_batch": ms_inf, "path": "HIP index ops + gg_k_gridconv (down layers) / source-side "
                                    "conv + fused attention-max kernel (up layers) + MFMA eval MLPs"}
        net.train()
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, B, points, kind, a.cpu_sample)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        # teardown order: the step's graph (it holds the captured all-reduce) before the communicator
        del step
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
