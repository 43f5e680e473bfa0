"""Round-6 kernels and host paths WRITTEN WHILE THE GPU POOL WAS CLOSED TO THE BUILD (`gpurun`: "GPU use for this
repository has been closed from outside the build"): none of these tests had run on hardware when it was
committed.  They sort last on purpose -- `pytest -x` reaches them after every verified test -- and every path they
exercise is OFF by default (train/options.py: NOZ_BWD_MOMENTS, INDEX_SIDE_STREAM; the C entries are new and nothing
else calls them), so a failure here says "the opt-in path is wrong", never "the shipped step is wrong".
DESIGN.md section 9 lists them with their status.

Until a GPU session has passed them they run only when asked for (GG_R6_UNVERIFIED=1): the round-end `pytest -m gpu`
of the driver must say what it said for the verified tree, not fail on code nobody has executed.  tools/session.sh sets
the variable."""
import ctypes
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("GG_R6_UNVERIFIED") != "1",
                                 reason="round-6 opt-in paths, never run on hardware: set GG_R6_UNVERIFIED=1")]

DEV = "cuda:0"


@pytest.mark.parametrize("ncent,P", [(655, 5), (64, 7), (2000, 5), (33, 1), (4096, 5), (70000, 5), (13, 3),
                                     (655360, 5)])
def test_att_bwd_noz_mom_equals_the_accumulating_form(ncent, P):
    """gridgcn_att_bwd_noz_mom (S1 / S2 from the forward's fp64 moments; gg_k_att_bwd_nz2<true> + the fused
    reduce / finish) against gridgcn_att_bwd_noz on the same inputs.  dX, the fp64 sums of the layer in front and the
    BatchNorm-backward vectors are the same arithmetic in the same order: equal bit for bit (sums: to their atomics'
    ordering).  dW differs only in its dense part -- (cz + bz (b2 - mu)) S1 + bz W2 S2 from fp64 moments instead of
    fp32-accumulated ones: within 2e-6 of the largest |dW| (and both within that of a float64 evaluation)."""
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _ptr, _stream
    lib = _lib.load()
    cin, C = 32, 128
    E = ncent * P
    g = torch.Generator(device=DEV).manual_seed(ncent * 17 + P)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    Z1 = rnd(E, cin)
    s1v, h1v, m1v, r1v = rnd(cin).abs() + 0.5, rnd(cin) * 0.1, rnd(cin) * 0.1, rnd(cin).abs() + 0.5
    W2, b2 = rnd(C, cin) * 0.2, rnd(C) * 0.1
    gamma, beta = rnd(C).abs() + 0.5, rnd(C) * 0.3
    s2v, m2v, r2v = rnd(C).abs() + 0.5, rnd(C) * 0.1, rnd(C).abs() + 0.5
    sums_a = rnd(2 * C).double()
    amax = torch.randint(0, P, (ncent, C), device=DEV, dtype=torch.int32, generator=g).to(torch.uint8)
    ga = rnd(ncent, C)
    st = _stream(Z1)
    assert lib.gridgcn_att_bwd_noz_mom_supported(E, cin, C, P) == 1
    # the forward's moments of a1 = relu(Z1 * s1 + h1)
    nb = ctypes.c_size_t(0)
    assert lib.gridgcn_att_fwd_noz_workspace_bytes(E, cin, C, ctypes.byref(nb)) == 0
    wsf = torch.empty(nb.value, dtype=torch.uint8, device=DEV)
    vec = torch.empty(4, C, device=DEV)
    rc = lib.gridgcn_att_bn2_moments(_ptr(Z1), _ptr(s1v), _ptr(h1v), _ptr(W2), _ptr(b2), _ptr(gamma), _ptr(beta), E,
                                     cin, C, 1e-3, 0.0, _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]), _ptr(vec[3]), None,
                                     None, None, None, _ptr(wsf), nb.value, st)
    assert rc == 0
    off = ctypes.c_size_t(0)
    assert lib.gridgcn_att_moments_offset(E, cin, C, ctypes.byref(off)) == 0
    assert off.value + 17 * 64 * 8 == nb.value
    mom = wsf[off.value:].view(torch.float64)
    # the moments themselves against float64
    a1 = torch.relu(Z1.double() * s1v.double() + h1v.double())
    S1, S2 = a1.sum(0), a1.t() @ a1
    mm = mom.cpu()
    S1k = mm[1024:1024 + 32] + mm[1024 + 32:1024 + 64]
    assert float((S1k - S1.cpu()).abs().max()) <= 1e-6 * max(1.0, float(S1.abs().max()))
    for k in (0, 5, 12, 31):
        for i in (0, 7, 31):
            got = float(mm[((k & 3) + 4 * (k >> 3)) * 64 + i + 32 * ((k >> 2) & 1)])
            assert abs(got - float(S2[k, i])) <= 2e-6 * max(1.0, float(S2.abs().max()))
    nbytes = ctypes.c_size_t(0)
    assert lib.gridgcn_att_bwd_noz_workspace_bytes(E, cin, C, ctypes.byref(nbytes)) == 0
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=DEV)

    def run(use_mom):
        dA1 = torch.full((E + 64, cin), 7.0, device=DEV)          # (guard rows: nothing may be written past E)
        dW2 = torch.empty(C, cin, device=DEV)
        v = torch.empty(4, C, device=DEV)
        acc = torch.zeros(3 * cin, dtype=torch.float64, device=DEV)
        if use_mom:
            rc = lib.gridgcn_att_bwd_noz_mom(_ptr(Z1), _ptr(s1v), _ptr(h1v), _ptr(m1v), _ptr(r1v), _ptr(W2), _ptr(b2),
                                             _ptr(s2v), _ptr(m2v), _ptr(r2v), _ptr(sums_a), _ptr(amax), _ptr(ga),
                                             int(P), E, cin, C, _ptr(mom), _ptr(dA1), _ptr(dW2), _ptr(v[0]),
                                             _ptr(v[1]), _ptr(v[2]), _ptr(v[3]), _ptr(acc[:2 * cin]), _ptr(ws),
                                             nbytes.value, st)
        else:
            rc = lib.gridgcn_att_bwd_noz(_ptr(Z1), _ptr(s1v), _ptr(h1v), _ptr(m1v), _ptr(r1v), _ptr(W2), _ptr(b2),
                                         _ptr(s2v), _ptr(m2v), _ptr(r2v), _ptr(sums_a), _ptr(amax), _ptr(ga), int(P),
                                         E, cin, C, _ptr(dA1), _ptr(dW2), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]),
                                         _ptr(v[3]), _ptr(acc[:2 * cin]), _ptr(acc[2 * cin:]), _ptr(ws),
                                         nbytes.value, st)
        assert rc == 0
        torch.cuda.synchronize()
        return dA1, dW2, v, acc

    x1, w1, v1, a1m = run(True)
    x0, w0, v0, a0 = run(False)
    assert bool((x1[E:] == 7.0).all())
    assert torch.equal(x1[:E], x0[:E])
    assert torch.equal(v1, v0)
    assert float((a1m[:2 * cin] - a0[:2 * cin]).abs().max()) <= 1e-12 * max(1.0, float(a0.abs().max()))
    scale = float(w0.abs().max())
    assert scale > 0 and bool(torch.isfinite(w1).all())
    assert float((w1 - w0).abs().max()) <= 2e-6 * scale * max(1.0, (E / 1e5) ** 0.5)
    # twice the same words (fixed summation order)
    x2, w2, _, _ = run(True)
    assert torch.equal(w2, w1) and torch.equal(x2[:E], x1[:E])


def test_att_bwd_noz_mom_rejects_what_it_does_not_take():
    from grid_gcn_amd import _lib
    lib = _lib.load()
    assert lib.gridgcn_att_bwd_noz_mom_supported(655360 * 5, 32, 128, 5) == 1
    assert lib.gridgcn_att_bwd_noz_mom_supported(655360 * 5, 16, 128, 5) == 0
    assert lib.gridgcn_att_bwd_noz_mom_supported(655360 * 5, 32, 64, 5) == 0
    assert lib.gridgcn_att_bwd_noz_mom_supported(101, 32, 128, 5) == 0          # E % P
    assert lib.gridgcn_att_bwd_noz_mom_supported(1 << 24, 32, 128, 4) == 0      # 32-bit byte offsets
    off = ctypes.c_size_t(0)
    assert lib.gridgcn_att_moments_offset(100, 16, 128, ctypes.byref(off)) != 0


@pytest.mark.parametrize("up_variant", ["ball", "gridify_up"])
def test_seg_step_with_round6_switches_equals_the_shipped_step(up_variant):
    """One training step of the segmentation net with every round-6 opt-in path ON against the shipped
    configuration: same loss to fp32 round-off, gradients within 1e-5 of their largest entry per tensor."""
    import copy
    from grid_gcn_amd import model, synth
    from grid_gcn_amd.train.options import OPT
    torch.manual_seed(3)
    cfg = dict(model.SEG_8192, dropout=0.0, up_neigh_fetch=up_variant == "ball")
    net = model.GGCNSeg(cfg, fixed_seed=True).to(DEV).train()
    state = copy.deepcopy(net.state_dict())
    data, npn = synth.make_batch(2, 8192, "planes", first_id=21)
    x = torch.from_numpy(data[..., :3].copy()).to(DEV)
    n = torch.from_numpy(npn).to(DEV)
    lab = torch.randint(0, 21, (2, 8192), device=DEV)
    res = []
    for on in (False, True):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        with OPT.override(**{k: on for k in R6_SWITCHES}):
            loss = model.seg_loss(net(x, n), lab)
            loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), [p.grad.detach().clone() for p in net.parameters()]))
    (l0, g0), (l1, g1) = res
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0)), (l0, l1)
    for a, b in zip(g0, g1):
        s = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * max(s, 1e-3), (float((a - b).abs().max()), s)


R6_SWITCHES = ("NOZ_BWD_MOMENTS", "INDEX_SIDE_STREAM")


def test_index_side_stream_inside_a_captured_step():
    """OPT.INDEX_SIDE_STREAM under graph.GraphedTrainStep: the fork / joins are captured as graph edges; three replays
    of the graphed step leave the same parameters as three replays of the single-stream graph (fixed sampling seed)."""
    import copy
    from grid_gcn_amd import graph, model, optim, synth
    from grid_gcn_amd.train.options import OPT
    cfg = dict(model.SEG_8192, dropout=0.0)
    data, npn = synth.make_batch(2, 8192, "planes", first_id=5)
    x = torch.from_numpy(data[..., :3].copy()).to(DEV)
    n = torch.from_numpy(npn).to(DEV)
    lab = torch.randint(0, 21, (2, 8192), device=DEV)
    torch.manual_seed(11)
    base = model.GGCNSeg(cfg, fixed_seed=True).to(DEV).train()
    state = copy.deepcopy(base.state_dict())
    out = []
    for on in (False, True):
        net = model.GGCNSeg(cfg, fixed_seed=True).to(DEV).train()
        net.load_state_dict(state)
        opt = optim.Adam(net.parameters(), lr=1e-3)
        with OPT.override(INDEX_SIDE_STREAM=on):
            step = graph.GraphedTrainStep(net, opt, model.seg_loss, (x, n), lab, warmup=1)
            for _ in range(3):
                loss = step()
        torch.cuda.synchronize()
        out.append((float(loss), [p.detach().clone() for p in net.parameters()]))
    (l0, p0), (l1, p1) = out
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)), (l0, l1)
    for a, b in zip(p0, p1):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
