"""The BASELINE.json configs that do not carry the headline metric, as bench.py lines of the same
schema (python bench.py --config cfg1|cfg2|cfg3|cfg3up|cfg5).  Same timing contract as bench.py.

  cfg1   ModelNet40 1024-pt, batch 1, ONE Gridify layer: ms per CAGQ layer (latency case)
  cfg2   ModelNet40 1024-pt classifier, batch 32, fp32: clouds/s fwd+bwd
  cfg3   ScanNet 8192-pt segmentation, batch 16, up path BallKNN (the shipped yaml)
  cfg3up the same with GridifyUp as the up path (up_neigh_fetch: False)
  cfg5   synthetic 200k-pt clouds, P = 64, 4-layer GridConv (builder-defined HBM stress)
"""
import torch

from grid_gcn_amd import dp, model, model_cls, model_synth, ops, synth

MFMA_F32_PEAK_TF = 157.3


def _line(metric, value, unit, a, world, ms_step, workload, extra):
    out = {"metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": unit != "ms",
           "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": workload, "parallelism": "dp%d" % world}}
    cx = extra.pop("config_extra", None)
    if cx:
        out["config"].update(cx)
    out.update(extra)
    return out


def run(a, world, rank, dev, traffic, time_training, cagq_roofline, make_step, time_allreduce=None,
        param_sync_spread=None, make_opt=None):
    torch.manual_seed(0)
    if a.config == "cfg1":
        # latency of one CAGQ layer on one cloud: the reference's CPU-runnable case
        B, N = a.batch or 1, a.points or 1024
        data, npn = synth.make_batch(B, N, "ball", first_id=rank * B)
        kw = synth.gridify_kwargs(synth.CLS_MODELNET40, 0)
        d4, n = torch.from_numpy(data).to(dev), torch.from_numpy(npn).to(dev)
        ms, rc = cagq_roofline(d4, n, kw, B, N, traffic, "gridify_N%d_B%d" % (N, B))
        return _line("ms per CAGQ layer (ModelNet40 1024-pt, batch 1)", ms, "ms", a, world, ms,
                     "BASELINE configs[0]: ModelNet40 %d-pt, batch %d, Gridify layer 0 (grid 40^3, "
                     "k 7, P 64, O 1024)" % (N, B), {"ms_per_cagq_layer": ms, "roofline": rc})

    if a.config == "cfg2":
        B, N = a.batch or 32, a.points or 1024
        net = model_cls.GGCNCls(seed=rank).to(dev).train()
        data, npn = synth.make_batch(B, N, "ball", first_id=rank * B)
        lab = torch.randint(0, 40, (B,), device=dev)
        loss_fn = model_cls.cls_loss
        workload = ("BASELINE configs[1]: ModelNet40 %d-pt classifier (3 Gridify + GridConv layers, "
                    "FC head), batch %d per GPU, Adam, fp32" % (N, B))
        metric = "point-clouds/sec fwd+bwd (ModelNet40 1024-pt classifier)"
        flops = 3.0 * model_cls.cls_forward_flops(net, B)
    elif a.config in ("cfg3", "cfg3up"):
        B, N = a.batch or 16, a.points or 8192
        cfg = dict(model.SEG_8192, up_neigh_fetch=(a.config == "cfg3"))
        net = model.GGCNSeg(cfg, seed=rank).to(dev).train()
        data, npn = synth.make_batch(B, N, "planes", first_id=rank * B)
        lab = torch.randint(0, cfg["num_classes"], (B, N), device=dev)
        loss_fn = model.seg_loss
        workload = ("BASELINE configs[2]: ScanNet %d-pt segmentation, batch %d per GPU, up path %s, "
                    "Adam, fp32" % (N, B, "BallKNN" if a.config == "cfg3" else "GridifyUp"))
        metric = "point-clouds/sec fwd+bwd (ScanNet 8192-pt, %s up path)" % (
            "BallKNN" if a.config == "cfg3" else "GridifyUp")
        fe, fr = model.seg_forward_flops(net, B, N)
        flops = 3.0 * (fe + fr)
    else:  # cfg5
        B, N = a.batch or 8, a.points or 200000
        net = model_synth.GGCNSynth(seed=rank).to(dev).train()
        data, npn = synth.make_batch(B, N, "planes", first_id=rank * B)
        lab = torch.randint(0, 40, (B,), device=dev)
        loss_fn = model_synth.synth_loss
        workload = ("BASELINE configs[4]: synthetic %d-pt clouds, P = 64, 4-layer GridConv "
                    "(64^3/32^3/16^3/8^3, O 16384/4096/1024/256, C 64/128/256/512), batch %d per "
                    "GPU, Adam, fp32" % (N, B))
        metric = "point-clouds/sec fwd+bwd (synthetic 200k-pt, 4-layer GridConv)"
        flops = 3.0 * model_synth.forward_flops(net, B)

    opt = make_opt(net, a.eager)
    sync = dp.FlatGradAllReduce(net)
    sync.broadcast_parameters()
    x = torch.from_numpy(data[..., :3].copy()).to(dev)
    n = torch.from_numpy(npn).to(dev)
    step, step_mode = make_step(net, opt, sync, loss_fn, (x, n), lab, not a.eager)

    dt, t_enq = time_training(step, a.steps, a.warmup, world, dev)
    ms_step = dt / a.steps * 1e3
    extra = {"host_enqueue_ms_per_step": t_enq / a.steps * 1e3, "step_mode": step_mode,
             "rccl_capture_probe": getattr(make_step, "probe", None)}
    extra["config_extra"] = {"global_batch": world * B, "points_per_cloud": N}
    if time_allreduce is not None:
        extra.update(time_allreduce(sync, world, dev))
        extra["param_sync_spread"] = param_sync_spread(net, world, dev)
    if flops is not None:
        tf = flops / (ms_step * 1e-3) / 1e12
        peak = MFMA_F32_PEAK_TF if a.dtype == "f32" else 2500.0
        extra["roofline_step"] = {"bound": "mfma", "kernel": "whole training step (all kernels)",
                                  "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                                  "frac": tf / peak, "traffic": None,
                                  "algorithmic_flops_per_step": flops}
    if rank == 0 and world == 1:
        grid = net.cfg["grid"]
        kw = synth.gridify_kwargs(grid, 0)
        d4 = torch.from_numpy(data).to(dev)
        ms, rc = cagq_roofline(d4, n, kw, B, N, traffic, "gridify_N%d_B%d" % (N, B))
        extra["ms_per_cagq_layer"] = ms
        extra["roofline"] = rc
        if a.config == "cfg5":
            # the materialising neighbour gather at the shape of layer 1 (64 feature channels,
            # 4096 x 64 neighbours per cloud out of 16384 centres)
            with torch.no_grad():
                out0 = ops.Gridify(d4, n, **kw)
                feat = torch.randn((B, kw["max_o_grid"], 64), device=dev)
                src = torch.cat([out0[2], feat], dim=2).contiguous()
                idx1 = ops.Gridify(out0[2], out0[4], **synth.gridify_kwargs(grid, 1))[0]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    ops.batch_take_g(src, idx1, neighbour_index=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    ops.batch_take_g(src, idx1, neighbour_index=True)
                e1.record()
                torch.cuda.synchronize()
                ms_g = e0.elapsed_time(e1) / 20
            alg_g = 4.0 * src.numel() + 4.0 * idx1.numel() + 4.0 * idx1.numel() * src.shape[2]
            extra["roofline_gather"] = {
                "bound": "hbm", "kernel": "gridgcn_batch_take (layer 1: src %s, index %s)" % (
                    list(src.shape), list(idx1.shape)),
                "achieved": alg_g / (ms_g * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": alg_g / (ms_g * 1e-3) / 1e9 / 8000.0, "traffic": None,
                "algorithmic_bytes_per_launch": alg_g, "ms_per_launch": ms_g}
    return _line(metric, world * B * a.steps / dt, "point-clouds/s", a, world, ms_step, workload,
                 extra)
