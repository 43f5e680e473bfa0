"""bench.py -- BASELINE.json metric on MI355X: point-clouds/s fwd+bwd of the ScanNet 81920-pt
segmentation network (configs[3]: batch 8 per GPU, data parallel), plus ms per CAGQ layer
(= one Gridify call) with its HBM roofline, and the CPU baseline timed beside it.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = zero_grad, forward (3 Gridify + 3 BallKNN + 6 gathers + 6 GridConv layers + head),
loss, backward, one flat RCCL all-reduce of the gradients (N > 1), Adam update.  Inputs are
synthetic clouds already resident in HBM (no dataset offline); weights are Xavier random.
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
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from grid_gcn_amd import dp, model, ops, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s achievable)


def cpu_baseline(cfg, points, kind):
    """Reference-side number: the S0 oracle (C, scalar) for the index ops + PyTorch CPU for
    gather/GridConv, fwd+bwd of ONE cloud of the same workload (bounded sample)."""
    from oracle.torch_index_ops import OracleIndexOps
    torch.manual_seed(0)
    m = model.GGCNSeg(cfg, index_ops=OracleIndexOps)
    m.train()
    data, npn = synth.make_batch(1, points, kind)
    x = torch.from_numpy(data[..., :3].copy())
    n = torch.from_numpy(npn)
    lab = torch.randint(0, cfg["num_classes"], (1, points))
    t0 = time.time()
    loss = model.seg_loss(m(x, n), lab)
    loss.backward()
    dt = time.time() - t0
    # the CAGQ layer alone on the CPU (oracle, 1 core)
    from oracle import oracle as orc
    kw = synth.gridify_kwargs(cfg["grid"], 0)
    t1 = time.time()
    orc.gridify(data, npn, **kw)
    dt_g = time.time() - t1
    return {"value": 1.0 / dt, "unit": "point-clouds/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "1 cloud x %d pts, fwd+bwd: S0 oracle (C, 1 thread) for Gridify/BallKNN + "
                      "PyTorch-CPU (%d threads) for gather/GridConv" % (points, torch.get_num_threads()),
            "ms_per_cagq_layer_1core": dt_g * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=81920)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    cfg = model.SEG_81920 if a.points > 8192 else model.SEG_8192
    if a.points not in (8192, 81920):       # off-config sizes keep the layer tables
        cfg = dict(cfg)
    kind = "planes"
    torch.manual_seed(0)
    net = model.GGCNSeg(cfg).to(dev)
    net.train()
    # one multi-tensor kernel for the whole update instead of ~10 tiny launches per parameter
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
    sync = dp.FlatGradAllReduce(net)
    sync.broadcast_parameters()
    B = a.batch
    data, npn = synth.make_batch(B, a.points, kind, first_id=rank * B)   # a different shard per rank
    x = torch.from_numpy(data[..., :3].copy()).to(dev)
    n = torch.from_numpy(npn).to(dev)
    lab = torch.randint(0, cfg["num_classes"], (B, a.points), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.seg_loss(net(x, n), lab)
        loss.backward()
        sync()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss).item()

    out = {
        "metric": "point-clouds/sec fwd+bwd (ScanNet 81920-pt)", "value": world * B * a.steps / dt,
        "unit": "point-clouds/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: ScanNet %d-pt segmentation, batch %d per GPU, "
                               "3 Gridify down + 3 BallKNN up layers, Adam, fp32" % (a.points, B),
                   "global_batch": world * B, "points_per_cloud": a.points,
                   "parallelism": "dp%d" % world,
                   "kernels": "hand-written HIP for Gridify/BallKNN, edge inputs (gather+geo) and their "
                              "sorted backward, all conv+BatchNorm+ReLU stacks fwd+bwd (fp32 MFMA), "
                              "att product + max, fc1 + dropout + class scores (one op) + softmax "
                              "cross-entropy; PyTorch-ROCm for the small GEMMs on source points, "
                              "concat/mask on [B,O,C] and fused Adam"},
    }

    if rank == 0 and world == 1:
        # ---- ms per CAGQ layer: Gridify of down layer 0 on the same batch, HIP events on the
        #      stream the kernels are launched on (torch's current stream) ----
        kw = synth.gridify_kwargs(cfg["grid"], 0)
        d4 = torch.from_numpy(data).to(dev)
        for _ in range(5):
            ops.Gridify(d4, n, **kw)
        torch.cuda.synchronize()
        iters = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.Gridify(d4, n, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        alg = B * synth.gridify_algorithmic_bytes(a.points, kw["max_o_grid"], kw["max_p_grid"])
        ach = alg / (ms * 1e-3) / 1e9
        out["ms_per_cagq_layer"] = ms
        out["roofline_cagq"] = {"bound": "hbm", "kernel": "gridgcn_gridify (memset + 6 launches, "
                                "down layer 0)", "achieved": ach, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "traffic": 134e6,  # FETCH+WRITE_SIZE, profiles/r1_pmc_index.txt
                                "algorithmic_bytes_per_launch": alg}
        # ---- inference forward through the fused GridConv kernels (the reference's own speed
        #      recipe times inference: train_gpu_speed_profiling.py:105-118) + the dominant
        #      hand-written kernel of the path: gg_k_gridconv of up layer 2 (fp32 MFMA bound) ----
        net.eval()
        with torch.no_grad():
            net.jobs = []
            net(x, n)
            jobs, net.jobs = net.jobs, None
            for _ in range(2):
                net(x, n)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                net(x, n)
            e1.record()
            torch.cuda.synchronize()
            ms_inf = e0.elapsed_time(e1) / 10
            name, layer, cent_, src_, idx_ = max(jobs, key=lambda j: j[4].numel() * j[1].cin)
            pt, att = layer.packed_layers()
            src_ = src_.contiguous()
            call = lambda: ops.gridconv_forward(src_, idx_, cent_, pt, att,  # noqa: E731
                                                has_feats=layer.has_feats,
                                                localfdim=layer.localfdim)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms_k = e0.elapsed_time(e1) / 20
            # the path evaluation actually takes for this layer (one point conv: source-side)
            ms_src = None
            from grid_gcn_amd import train_ops as _to
            attl, ptl = [layer.att1[0], layer.att2[0]], list(layer.pt_mlp)
            if _to.edge_block_src_eval_supported(ptl, attl, src_, layer.has_feats):
                call2 = lambda: _to.edge_block_src_eval(src_, idx_, cent_.contiguous(), ptl[0],  # noqa: E731
                                                        attl, layer.localfdim)
                for _ in range(3):
                    call2()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    call2()
                e1.record()
                torch.cuda.synchronize()
                ms_src = e0.elapsed_time(e1) / 20
        macs = sum(l.lin.in_features * l.lin.out_features
                   for seq in (layer.pt_mlp, layer.att1, layer.att2) for l in seq)
        flops = 2.0 * idx_.numel() * macs
        tf = flops / (ms_k * 1e-3) / 1e12
        out["roofline_inference"] = {
            "bound": "mfma", "kernel": "gg_k_gridconv (GridConv %s: gather + per-edge MLPs + att "
            "product + max, one launch, inference-mode BatchNorm)" % name,
            "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "traffic": None,
            "algorithmic_flops_per_launch": flops, "ms_per_launch": ms_k,
            "dtype": "f32 (v_mfma_f32_32x32x2_f32)",
            # evaluation runs this layer through the source-side kernels instead (first conv once
            # per source point, gathered by the max kernel): same result, fewer executed flops
            "ms_source_side_path": ms_src,
            "note": "kernel-level figure for gg_k_gridconv on this layer's shape; the evaluation "
                    "forward itself runs up layers through ms_source_side_path (source-side conv + "
                    "gridgcn_att_max_eval) and uses gg_k_gridconv for the down layers"}
        # ---- dominant kernels of the TIMED training step.  The point conv of this layer runs on
        #      the source points (gridgcn_edgelin.hip), so the largest per-edge GEMMs left are those
        #      of the attention MLP: backward of its C/4 -> C conv = gg_k_att_bwd_fused (dZ formed
        #      in registers from the sparse arg-max gradient; dX, the BatchNorm-backward sums of
        #      the layer in front and dW in ONE pass over Z) + its small reduce.  With K = C/4 it is
        #      HBM bound. ----
        from grid_gcn_amd import train_ops
        cin_b = layer.att2[0].lin.in_features
        c_b = layer.att2[0].lin.out_features
        ncent_b, p_b = idx_.shape[0] * idx_.shape[1], idx_.shape[2]
        ms_b = train_ops.time_linear_bwd(ncent_b, p_b, cin_b, c_b, iters=10, device=dev,
                                         ndx=cin_b, prev_bn=True)
        e_b = float(ncent_b * p_b)
        # algorithmic bytes of the operation: read Z [E,C] once, the sparse upstream gradient
        # (amax, gval) [ncent,C], the previous layer's raw output [E,cin]; write dX [E,cin]
        bytes_b = 4.0 * e_b * (c_b + 2 * cin_b) + 8.0 * ncent_b * c_b
        gbs_b = bytes_b / (ms_b * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "gg_k_att_bwd_fused + gg_k_att_dw_reduce "
                           "(fused backward of the %d->%d attention conv of GridConv %s over %d "
                           "edges: BN/ReLU backward formed in registers from the sparse arg-max "
                           "gradient; dX, BN-backward sums of the layer in front and dW in one pass "
                           "over Z)" % (cin_b, c_b, name, ncent_b * p_b),
                           "achieved": gbs_b, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": gbs_b / HBM_PEAK_GBS,
                           # FETCH_SIZE (x2: 16-byte streaming reads) + WRITE_SIZE of the kernel at
                           # this shape, profiles/r1_pmc_summary.txt
                           "traffic": 3.23e9 if (a.points == 81920 and B == 8) else None,
                           "algorithmic_bytes_per_launch": bytes_b, "ms_per_launch": ms_b,
                           "algorithmic_flops_per_launch": 4.0 * e_b * cin_b * c_b,
                           "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}
        # the largest MFMA-bound kernel of the step: forward of the 256->128 update conv over
        # all B*N points (previous BatchNorm+ReLU applied while loading, statistics epilogue)
        e_f, cin_f, c_f = B * a.points, 256, 128
        ms_f = train_ops.time_linear_fwd(e_f, cin_f, c_f, iters=10, device=dev)
        tf_f = 2.0 * e_f * cin_f * c_f / (ms_f * 1e-3) / 1e12
        out["roofline_mfma"] = {"bound": "mfma", "kernel": "gg_k_linear_fwd_direct (%d->%d conv + "
                                "BN/ReLU prologue + statistics over %d rows)" % (cin_f, c_f, e_f),
                                "achieved": tf_f, "peak": 157.3, "unit": "TFLOP/s",
                                "frac": tf_f / 157.3, "traffic": None,
                                "algorithmic_flops_per_launch": 2.0 * e_f * cin_f * c_f,
                                "ms_per_launch": ms_f, "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}
        # ---- the materialising neighbour gather as an operator (SURVEY §8(d) algorithmic bytes):
        #      batch_take_g forward + its sorted backward at the shape of layer up2 ----
        with torch.no_grad():
            gsrc = src_.contiguous()
            tk = lambda: ops.batch_take_g(gsrc, idx_, neighbour_index=True)  # noqa: E731
            for _ in range(3):
                tk()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                tk()
            e1.record()
            torch.cuda.synchronize()
            ms_g = e0.elapsed_time(e1) / 20
            gout = tk()
            tb = lambda: ops.batch_take_g_backward(gout, idx_, gsrc.shape[1], True)  # noqa: E731
            for _ in range(3):
                tb()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                tb()
            e1.record()
            torch.cuda.synchronize()
            ms_gb = e0.elapsed_time(e1) / 20
        alg_g = 4.0 * gsrc.numel() + 4.0 * idx_.numel() + 4.0 * idx_.numel() * gsrc.shape[2]
        out["roofline_gather"] = {
            "bound": "hbm", "kernel": "gridgcn_batch_take (batch_take_g of GridConv %s: src %s, "
            "index %s)" % (name, list(gsrc.shape), list(idx_.shape)),
            "achieved": alg_g / (ms_g * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_g / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": alg_g, "ms_per_launch": ms_g,
            "backward_sorted_ms": ms_gb,
            "backward_sorted_GBps": alg_g / (ms_gb * 1e-3) / 1e9}
        del gout
        out["inference"] = {"value": B / (ms_inf * 1e-3), "unit": "point-clouds/s",
                            "ms_per_batch": ms_inf, "path": "HIP index ops + gg_k_gridconv (down layers) / source-side "
                                    "conv + fused attention-max kernel (up layers) + MFMA eval MLPs"}
        net.train()
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.points, kind)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
