"""The glue kernels of a training step (csrc/gridgcn_optim.hip, gridgcn_ballgrid.hip strided entry) against
the stock PyTorch ops they replace."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _params(sizes, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in sizes]


SIZES = [(128, 131), (128,), (1,), (21, 128), (21,), (257, 9), (3,), (2048 + 5,), (64, 64), (1024,), (1023,),
         (1025,)]


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_matches_torch(wd):
    """five steps with fresh gradients: parameters and both moments within rounding of torch.optim.Adam"""
    from grid_gcn_amd import optim
    p1 = _params(SIZES, 0)
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in p1]
    o1 = torch.optim.Adam(p1, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    o2 = optim.Adam(p2, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    g = torch.Generator().manual_seed(1)
    for it in range(5):
        for a, b in zip(p1, p2):
            gr = torch.randn(a.shape, generator=g).to(DEV) * (0.1 + it)
            a.grad, b.grad = gr.clone(), gr.clone()
        v0 = [b._version for b in p2]
        o1.step()
        o2.step()
        assert all(b._version > v for b, v in zip(p2, v0))      # caches keyed on _version see the update
    for a, b in zip(p1, p2):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))
        assert torch.allclose(o1.state[a]["exp_avg"], o2.state[b]["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(o1.state[a]["exp_avg_sq"], o2.state[b]["exp_avg_sq"], rtol=1e-5, atol=1e-9)
    assert int(o2.state[p2[0]]["step"]) == 5


def test_adam_many_tensors_none_grads_and_mxnet_form():
    """300 tensors (three launches), some without a gradient; mx.optimizer.Adam's form against its formula"""
    from grid_gcn_amd import optim
    sizes = [(7 + i % 13, 3 + i % 5) for i in range(300)]
    ps = _params(sizes, 2)
    ref = [p.detach().clone().double() for p in ps]
    m = [torch.zeros_like(r) for r in ref]
    v = [torch.zeros_like(r) for r in ref]
    lr, b1, b2, eps, wd = 1e-2, 0.9, 0.999, 1e-8, 1e-3
    opt = optim.Adam(ps, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, mxnet=True)
    g = torch.Generator().manual_seed(3)
    for t in range(1, 4):
        for i, p in enumerate(ps):
            if i % 17 == 5:
                p.grad = None
                continue
            gr = torch.randn(p.shape, generator=g).to(DEV)
            p.grad = gr
            gd = gr.double() + wd * ref[i]
            m[i] = b1 * m[i] + (1 - b1) * gd
            v[i] = b2 * v[i] + (1 - b2) * gd * gd
            ref[i] = ref[i] - lr * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t) * m[i] / (v[i].sqrt() + eps)
        opt.step()
    for p, r in zip(ps, ref):
        assert float((p.double() - r).abs().max()) <= 3e-6 * max(1.0, float(r.abs().max()))


def test_adam_state_dict_round_trip_and_tensor_lr():
    from grid_gcn_amd import optim
    p1 = _params(SIZES[:5], 4)
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in p1]
    lr = torch.tensor(2e-3, device=DEV)
    o1 = optim.Adam(p1, lr=lr, weight_decay=1e-4)
    o2 = optim.Adam(p2, lr=2e-3, weight_decay=1e-4)
    g = torch.Generator().manual_seed(5)

    def grads():
        for a, b in zip(p1, p2):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()

    for _ in range(2):
        grads()
        o1.step()
        o2.step()
    sd = copy.deepcopy(o2.state_dict())
    o3 = optim.Adam(p2, lr=2e-3, weight_decay=1e-4)
    o3.load_state_dict(sd)
    lr.fill_(5e-4)                       # the kernel reads the device scalar at every launch
    o3.param_groups[0]["lr"] = 5e-4
    grads()
    o1.step()
    o3.step()
    for a, b in zip(p1, p2):
        assert torch.allclose(a, b, rtol=0, atol=2e-7)
    assert int(o3.state[p2[0]]["step"]) == 3


@pytest.mark.parametrize("ca,cb,mask,pad", [(4, 64, True, True), (4, 128, True, False), (4, 256, False, True),
                                            (3, None, False, True), (3, None, False, False), (4, 12, True, True),
                                            (4, 4, True, True), (3, 5, True, True), (4, 6, False, False),
                                            (5, 64, True, True)])
def test_cat_mask_matches_torch(ca, cb, mask, pad):
    from grid_gcn_amd.train import common as tcommon
    torch.manual_seed(ca * 100 + (cb or 0))
    B, O = 3, 37
    a = torch.randn(B, O, ca, device=DEV)
    b = torch.randn(B, O, cb, device=DEV, requires_grad=True) if cb else None
    mk = (torch.rand(B, O, device=DEV) > 0.3).float() if mask else None
    out, outp = tcommon.cat_mask(a, b, mk, pad=pad)
    bb = b if b is not None else torch.ones(B, O, 1, device=DEV)
    ref = torch.cat([a, bb * mk[..., None] if mk is not None else bb], dim=-1)
    assert torch.equal(out, ref)
    W = ref.shape[-1]
    if pad and W % 8:
        assert outp.shape[-1] == (W + 7) // 8 * 8 and torch.equal(outp[..., :W], ref)
        assert float(outp[..., W:].abs().max()) == 0.0
    else:
        assert outp is out
    if b is None:
        return
    # both outputs used, one through a column slice of a wider gradient (as the edge block's source rows)
    w1 = torch.randn_like(out)
    w2 = torch.randn_like(outp)
    loss = (out * w1).sum() + ((outp * w2).sum() if outp is not out else 0.0)
    loss.backward()
    b2 = b.detach().clone().requires_grad_(True)
    ref = torch.cat([a, b2 * mk[..., None] if mk is not None else b2], dim=-1)
    l2 = (ref * w1).sum() + ((ref * w2[..., :W]).sum() if outp is not out else 0.0)
    l2.backward()
    assert torch.allclose(b.grad, b2.grad, rtol=1e-6, atol=1e-6)


def test_ball_knn_reads_wider_rows_in_place():
    """xyz columns of [B, n, 4+C] rows (any row width): the indices of the packed call, rows >= upnum zero"""
    from grid_gcn_amd import ops
    torch.manual_seed(0)
    B, n, m = 2, 700, 300
    up = torch.rand(B, n, 7, device=DEV)
    down = torch.rand(B, m, 4, device=DEV)
    upnum = torch.tensor([[n], [n - 50]], dtype=torch.int32, device=DEV)
    downnum = torch.tensor([[m], [m - 9]], dtype=torch.int32, device=DEV)
    i1 = ops.BallKNN(up[..., :3].contiguous(), down[..., :3].contiguous(), downnum, upnum, k=3, radius=0.2)
    i2 = ops.BallKNN(up[..., :3], down[..., :3], downnum, upnum, k=3, radius=0.2)
    assert torch.equal(i1, i2)
    assert int(i2[1, n - 50:].abs().max()) == 0
    assert int((i1 >= 0).sum()) > 0


@pytest.mark.parametrize("B,N,O,P,C0,geo", [(2, 24, 256, 8, 64, True), (3, 1024, 3000, 5, 128, True),
                                            (1, 300, 77, 5, 32, False), (2, 5000, 600, 3, 16, True),
                                            (2, 1024, 40000, 5, 128, True)])
def test_edge_geo_forward_statistics_match_the_edge_pass(B, N, O, P, C0, geo):
    """BatchNorm sums of the source-side first conv from per-source counts / geo_vec sums
    (gridgcn_edge_geo_forward) against the pass over every (edge, channel) (gridgcn_edge_lin0_forward);
    att16 bit for bit, Gsum against a float64 scatter."""
    import ctypes
    from grid_gcn_amd import _lib
    lib = _lib.load()
    torch.manual_seed(B * 1000 + N)
    Cs = 4 + 8
    src = torch.rand(B, N, Cs, device=DEV)
    cent = torch.rand(B, O, 4, device=DEV)
    idx = torch.randint(-1, N, (B, O, P), device=DEV, dtype=torch.int32)     # (-1: "no neighbour", clipped)
    Ysrc = torch.randn(B * N, C0, device=DEV)
    Wg = torch.randn(3, C0, device=DEV)
    b = torch.randn(C0, device=DEV)
    E = B * O * P
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())                               # noqa: E731
    att_a = torch.empty(E, 16, device=DEV)
    sums_a = torch.zeros(2 * C0, dtype=torch.float64, device=DEV)
    rc = lib.gridgcn_edge_lin0_forward(p(Ysrc), p(src), p(idx), p(cent), 4, B, N, Cs, O, P, C0,
                                       p(Wg) if geo else None, p(b), None, p(att_a), p(sums_a), st)
    assert rc == 0
    att_b = torch.empty(E, 16, device=DEV)
    sums_b = torch.zeros(2 * C0, dtype=torch.float64, device=DEV)
    gg = torch.zeros(12, dtype=torch.float64, device=DEV)
    gsum = torch.empty(B * N, 4, device=DEV)
    nb = ctypes.c_size_t(0)
    assert lib.gridgcn_edge_geo_forward_workspace_bytes(B, N, O, P, ctypes.byref(nb)) == 0
    ws = torch.empty(nb.value, dtype=torch.uint8, device=DEV)
    rc = lib.gridgcn_edge_geo_forward(p(Ysrc), p(src), p(idx), p(cent), 4, B, N, Cs, O, P, C0,
                                      p(Wg) if geo else None, p(b), p(att_b), p(gsum), p(gg), p(sums_b),
                                      p(ws), nb.value, st)
    assert rc == 0
    assert torch.equal(att_a, att_b)
    # per-source (G, cnt) against a float64 scatter of the geo vectors
    flat = (idx.long() + torch.arange(B, device=DEV)[:, None, None] * N).clamp(0, B * N - 1).reshape(-1)
    ref = torch.zeros(B * N, 4, dtype=torch.float64, device=DEV)
    g4 = torch.cat([att_a[:, 1:4].double(), torch.ones(E, 1, dtype=torch.float64, device=DEV)], dim=1)
    ref.index_add_(0, flat, g4)
    assert float((gsum.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    n = float(E)
    mean_a, mean_b = sums_a[:C0] / n, sums_b[:C0] / n
    var_a, var_b = sums_a[C0:] / n - mean_a ** 2, sums_b[C0:] / n - mean_b ** 2
    assert float((mean_a - mean_b).abs().max()) <= 2e-6 * max(1.0, float(mean_a.abs().max()))
    assert float((var_a - var_b).abs().max()) <= 5e-6 * float(var_a.abs().max())


def test_eval_constants_cache_follows_training():
    """the cached evaluation constants (folded BatchNorm, packed weights) are rebuilt after eager training steps
    (optim.Adam and the kernels' running-statistics updates move the version counters by hand) and after
    replays of a captured step: evaluation == evaluation with the cache switched off"""
    from grid_gcn_amd import graph, model, optim, synth
    from grid_gcn_amd.train.options import OPT
    torch.manual_seed(1)
    net = model.GGCNSeg(model.SEG_8192, seed=3).to(DEV)
    data, npn = synth.make_batch(2, 8192, "planes", first_id=5)
    x = torch.from_numpy(data[..., :3].copy()).to(DEV)
    n = torch.from_numpy(npn).to(DEV)
    lab = torch.randint(0, 21, (2, 8192), device=DEV)
    opt = optim.Adam(net.parameters(), lr=1e-2)

    def evaluate():
        net.eval()
        with torch.no_grad():
            a = net(x, n).clone()
            OPT.EVAL_CACHE = False
            try:
                b = net(x, n).clone()
            finally:
                OPT.EVAL_CACHE = True
        net.train()
        return a, b

    def step():
        opt.zero_grad(set_to_none=True)
        model.seg_loss(net(x, n), lab).backward()
        opt.step()

    net.train()
    outs = []
    for _ in range(3):
        a, b = evaluate()
        assert torch.equal(a, b)
        outs.append(a)
        step()
    assert float((outs[0] - outs[1]).abs().max()) > 0 and float((outs[1] - outs[2]).abs().max()) > 0
    gs = graph.GraphedTrainStep(net, opt, model.seg_loss, (x, n), lab, warmup=1)
    for _ in range(2):
        gs()
        a, b = evaluate()
        assert torch.equal(a, b)
        assert float((a - outs[-1]).abs().max()) > 0
        outs.append(a)


def test_adam_checkpoints_travel_to_and_from_torch_adam():
    """(a) a torch.optim.Adam checkpoint has no 'mxnet' group key; (b) this optimizer's per-parameter
    `step` entries are views of ONE device counter -- a checkpoint must carry independent scalars, or
    torch.optim.Adam's _foreach_add_ advances the shared element once per parameter."""
    from grid_gcn_amd import optim
    p1 = _params(SIZES[:5], 6)
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in p1]
    own = optim.Adam(p1, lr=1e-3)
    ref = torch.optim.Adam(p2, lr=1e-3)
    g = torch.Generator().manual_seed(7)

    def grads():
        for a, b in zip(p1, p2):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()

    for _ in range(2):
        grads()
        own.step()
        ref.step()
    sd = own.state_dict()
    steps = [s["step"] for s in sd["state"].values()]
    assert len({t.data_ptr() for t in steps}) == len(steps) and all(float(t) == 2.0 for t in steps)
    # own -> torch: three more steps there == three more steps here
    t2 = torch.optim.Adam(p2, lr=1e-3)
    t2.load_state_dict(copy.deepcopy(sd))
    # torch -> own (no 'mxnet' key in that checkpoint)
    o2 = optim.Adam(p1, lr=1e-3)
    o2.load_state_dict(copy.deepcopy(ref.state_dict()))
    for _ in range(3):
        grads()
        o2.step()
        t2.step()
    assert all(float(s["step"]) == 5.0 for s in t2.state.values())
    assert int(o2.state[p1[0]]["step"]) == 5
    for a, b in zip(p1, p2):
        assert torch.allclose(a, b, rtol=0, atol=5e-7)


def test_adam_second_checkpoint_carries_the_advanced_step():
    """Optimizer.state_dict() returns the optimizer's own per-parameter dicts; writing the frozen scalar
    into them cut self.state[p]['step'] loose from the device counter, so every later checkpoint repeated the
    first one's count.  step, checkpoint, step, checkpoint: the second says 2 more, and the live entry is still
    the counter's view (a resumed optimizer continues at the right bias correction)."""
    from grid_gcn_amd import optim
    ps = _params(SIZES[:4], 11)
    own = optim.Adam(ps, lr=1e-3)
    g = torch.Generator().manual_seed(3)

    def step(n):
        for _ in range(n):
            for p in ps:
                p.grad = torch.randn(p.shape, generator=g).to(DEV)
            own.step()

    step(5)
    sd1 = own.state_dict()
    assert all(float(s["step"]) == 5.0 for s in sd1["state"].values())
    live = own.state[ps[0]]["step"]
    assert live.dtype == torch.int32 and int(live) == 5          # still the counter's view, not the frozen copy
    step(4)
    assert int(own.state[ps[0]]["step"]) == 9
    sd2 = own.state_dict()
    assert all(float(s["step"]) == 9.0 for s in sd2["state"].values())
    assert all(float(s["step"]) == 5.0 for s in sd1["state"].values())       # the first checkpoint is its own copy
    # resume from the second checkpoint: the counter continues at 9
    o2 = optim.Adam(ps, lr=1e-3)
    o2.load_state_dict(copy.deepcopy(sd2))
    for p in ps:
        p.grad = torch.zeros_like(p)
    o2.step()
    assert int(o2.state[ps[0]]["step"]) == 10


def test_eval_cache_sees_torch_fused_adam():
    """torch's fused Adam rewrites the weights without moving Tensor._version; the evaluation caches
    (folded BatchNorm vectors, packed weights, the wgb table, SubGUpdate.packed_layers) are keyed on the
    parameter generation that the global optimizer-step hook advances -- evaluation after eager training with
    that optimizer == evaluation with the cache off."""
    from grid_gcn_amd import model, synth
    from grid_gcn_amd.train import common as tcommon
    from grid_gcn_amd.train.options import OPT
    torch.manual_seed(2)
    net = model.GGCNSeg(model.SEG_8192, seed=3).to(DEV)
    data, npn = synth.make_batch(2, 8192, "planes", first_id=9)
    x = torch.from_numpy(data[..., :3].copy()).to(DEV)
    n = torch.from_numpy(npn).to(DEV)
    lab = torch.randint(0, 21, (2, 8192), device=DEV)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)

    def evaluate(cache):
        net.eval()
        OPT.EVAL_CACHE = cache
        try:
            with torch.no_grad():
                return net(x, n).clone()
        finally:
            OPT.EVAL_CACHE = True
            net.train()

    net.train()
    prev = None
    for _ in range(3):
        a = evaluate(True)
        assert torch.equal(a, evaluate(False))
        assert prev is None or float((a - prev).abs().max()) > 0
        prev = a
        v0 = net.fc2.weight._version
        opt.zero_grad(set_to_none=True)
        model.seg_loss(net(x, n), lab).backward()
        g0 = tcommon._PARAM_GEN[0]
        opt.step()
        assert tcommon._PARAM_GEN[0] > g0
    a = evaluate(True)
    assert torch.equal(a, evaluate(False)) and float((a - prev).abs().max()) > 0
