"""GPU parity of the training-mode MLP kernels (csrc/gridgcn_train.hip: fp32 MFMA linear with
BatchNorm-statistics epilogue + BN/ReLU apply prologue, BN+ReLU backward) against the stock
PyTorch modules (Linear -> BatchNorm1d(batch stats) -> ReLU), forward, backward and running stats."""
import copy

import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from grid_gcn_amd.train import common as tcommon, edge as tedge, evalpath as teval, head as thead, mlp as tmlp
from grid_gcn_amd.train.options import OPT  # noqa: E402
from grid_gcn_amd.gridconv import mlp  # noqa: E402

DEV = "cuda:0"

CASES = [
    (5000, 3, [32, 32, 64]),
    (4097, 10, [16, 64]),
    (3000, 131, [128]),
    (1000, 260, [128]),
    (2500, 67, [64, 64, 128]),
    (77, 132, [128]),
    (6000, 131, [128, 128, 256]),
    (33, 4, [128]),
    # widths that are multiples of 8: the register-direct forward kernel (full / partial k chunks)
    (5000, 8, [32, 32, 64]),
    (4097, 16, [16, 64]),
    (3001, 136, [128]),
    (1000, 264, [128, 256]),
    (2500, 72, [64, 64, 128]),
    (6000, 128, [128, 128, 256]),
    (31, 24, [8, 128]),
    (70000, 64, [256, 32]),
]


@pytest.mark.parametrize("E,cin,dims", CASES, ids=["%d_%d_%s" % (c[0], c[1], "x".join(map(str, c[2])))
                                                    for c in CASES])
def test_mlp_train_matches_torch(E, cin, dims):
    torch.manual_seed(E + cin)
    ref = mlp(cin, dims).to(DEV).train()
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    new = copy.deepcopy(ref)
    x1 = (torch.randn(E, cin, device=DEV) * 1.5).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    assert tcommon.supported(list(new), x2)
    y1 = ref(x1)
    y2 = tmlp.mlp_bn_relu_train(x2, list(new))
    scale = float(y1.abs().max())
    assert float((y1 - y2).abs().max()) <= 2e-5 * max(1.0, scale)
    g = torch.randn_like(y1)
    y1.backward(g)
    y2.backward(g)

    def close(a, b, tol=2e-4):
        s = max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol * s, (float((a - b).abs().max()), s)
    # a ReLU whose pre-activation lies within round-off of zero may open in one implementation and
    # not in the other (different summation order): that changes whole rows of dX.  Allow such rows
    # at a rate of 1e-4 (~E*256 activations per layer, |z| < 1e-6 relative); all others must match.
    s = max(1e-3, float(x1.grad.abs().max()))
    bad = ((x2.grad - x1.grad).abs().amax(dim=1) > 2e-4 * s)
    nbad = int(bad.sum())
    assert nbad <= max(1, E // 10000), (nbad, E)      # (one such row whatever E)
    if nbad:
        # ... and every such row must HAVE a pre-activation within round-off of zero in the reference
        with torch.no_grad():
            h, zmin = x1.detach(), torch.full((E,), float("inf"), device=DEV)
            for layer in ref:
                zl = layer.lin(h)                     # (batch statistics, as in the step; buffers untouched)
                z = (zl - zl.mean(0)) / torch.sqrt(zl.var(0, unbiased=False) + layer.bn.eps) \
                    * layer.bn.weight + layer.bn.bias
                zmin = torch.minimum(zmin, (z.abs() / z.abs().amax(dim=0).clamp_min(1e-12)).amin(dim=1))
                h = torch.relu(z)
        assert float(zmin[bad].max()) < 2e-6, float(zmin[bad].max())
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), new.named_parameters()):
        if n1.endswith("lin.bias"):
            # analytically zero (a bias in front of BatchNorm); torch returns round-off noise
            assert float(p2.grad.abs().max()) == 0.0
            assert float(p1.grad.abs().max()) <= 1e-3 * max(1.0, float(g.abs().sum()) / E)
        else:
            # (each such row also moves the weight-gradient entries of its channel by one row's
            #  contribution -- dy * zhat for dgamma, dy * x for dW: up to ~1 % of a 4 k-row sum)
            close(p2.grad, p1.grad, 2e-2 if nbad else (5e-3 if E >= 20000 else 2e-4))
    for (n1, b1), (n2, b2) in zip(ref.named_buffers(), new.named_buffers()):
        if "num_batches" in n1:
            assert int(b1) == int(b2)
        else:
            close(b2, b1, 1e-5)


@pytest.mark.parametrize("E,cin,dims", [(6000, 128, [128, 128, 256]), (1500, 72, [64, 256, 32]), (2048, 256, [256, 128]),
                                        (16384, 136, [128, 64]), (95, 64, [256])],
                         ids=["6000", "1500", "2048", "16384", "95"])
def test_col_split_equals_whole_rows(E, cin, dims):
    """Layers of <= 16 K rows deal the output column tiles of the forward / dX kernels to separate
    workgroups (GRIDGCN_OPT_COL_SPLIT, round 4).  Every element of Z and dX is the same MFMA chain in the
    same order either way; only the BatchNorm sums are added up over differently shaped groups of rows, so
    the two paths agree to fp32 round-off of those sums, far inside the bar against the stock modules."""
    from grid_gcn_amd import _lib
    lib = _lib.load()
    torch.manual_seed(E)
    net = mlp(cin, dims).to(DEV).train()
    x = torch.randn(E, cin, device=DEV)
    g = torch.randn(E, dims[-1], device=DEV)
    res = []
    for on in (1, 0):
        _lib.check(lib.gridgcn_set_option(_lib.OPT_COL_SPLIT, on), "set_option")
        try:
            m = copy.deepcopy(net)
            xi = x.clone().requires_grad_(True)
            y = tmlp.mlp_bn_relu_train(xi, list(m))
            y.backward(g)
            res.append((y.detach(), xi.grad, [p.grad for p in m.parameters()]))
        finally:
            _lib.check(lib.gridgcn_set_option(_lib.OPT_COL_SPLIT, 1), "set_option")
    (y1, dx1, gp1), (y0, dx0, gp0) = res

    # (a hidden ReLU whose input lies within that round-off of zero may open on one path only and moves
    #  one row of dX / one row's share of the weight gradients: at most two such rows are tolerated)
    def close(a, b, tol, rows=0):
        bad = (a - b).abs() > tol * max(1e-3, float(b.abs().max()))
        nbad = int(bad.reshape(bad.shape[0], -1).any(dim=1).sum()) if bad.dim() > 1 else int(bad.sum())
        assert nbad <= rows, (nbad, float((a - b).abs().max()))
    close(y1, y0, 2e-6, rows=2)
    close(dx1, dx0, 2e-5, rows=2)
    flipped = not torch.equal(dx1 != 0, dx0 != 0) or float((dx1 - dx0).abs().max()) > 2e-5 * float(dx0.abs().max())
    for a, b in zip(gp1, gp0):
        close(a, b, 2e-2 if flipped else 2e-5, rows=a.shape[0] if flipped else 0)


@pytest.mark.parametrize("E,cin,dims", [(5000, 256, [512]), (4100, 264, [256, 512]), (8192, 128, [512, 256]),
                                        (8192, 128, [512]), (8192, 512, [256]), (20000, 128, [512]), (4096, 64, [512]),
                                        (300, 512, [512, 256]), (64, 1027, [512, 512]), (33, 259, [256, 512])],
                         ids=["5000_256", "4100_264", "8192_128", "8192_128a", "8192_512b", "20000_128", "4096_64",
                              "300_512", "64_1027", "33_259"])
def test_wide_layers_without_rocblas_match_torch(E, cin, dims):
    """Layers beyond 256 output / 384 input channels (last layer of the classifier and of the 200k-point
    workload, the classifier's FC head): register-direct kernels on 256-column slices where the layer is
    large, csrc/gridgcn_gemm.hip (any K) + the BatchNorm kernels otherwise -- against the stock modules."""
    torch.manual_seed(E + cin)
    ref = mlp(cin, dims).to(DEV).train()
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    new = copy.deepcopy(ref)
    x1 = torch.randn(E, cin, device=DEV).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    assert not tcommon.supported(list(new), x2) and tmlp.wide_supported(list(new), x2)
    y1 = ref(x1)
    y2 = tmlp.mlp_wide_train(x2, list(new))
    assert float((y1 - y2).abs().max()) <= 3e-5 * max(1.0, float(y1.abs().max()))
    g = torch.randn_like(y1)
    y1.backward(g)
    y2.backward(g)

    def close(a, b, tol):
        s = max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol * s, (float((a - b).abs().max()), s)
    # (a hidden ReLU whose input lies within round-off of zero may open in one implementation only: that moves
    #  whole rows of dX -- as in test_mlp_train_matches_torch, a handful of such rows is tolerated and then the
    #  weight gradients, which carry one row's share, get the wider bar)
    sx = max(1e-3, float(x1.grad.abs().max()))
    nbad = int(((x2.grad - x1.grad).abs().amax(dim=1) > 5e-4 * sx).sum())
    assert nbad <= (0 if len(dims) == 1 else max(2, E // 2000)), (nbad, E)
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), new.named_parameters()):
        if n1.endswith("lin.bias"):
            assert float(p2.grad.abs().max()) == 0.0
        else:
            close(p2.grad, p1.grad, 2e-2 if nbad else 5e-4)
    for (n1, b1), (n2, b2) in zip(ref.named_buffers(), new.named_buffers()):
        if "num_batches" in n1:
            assert int(b1) == int(b2)
        else:
            close(b2, b1, 1e-5)


def test_gemm_any_k_and_bias():
    """gridgcn_gemm_bias: modes 0 / 1 for any K (short last chunk), strided operands, optional bias."""
    torch.manual_seed(3)
    for M, N, K in ((70, 40, 259), (33, 195, 1027), (5, 3, 1), (128, 64, 7), (1000, 512, 515)):
        a = torch.randn(M, K + 5, device=DEV)[:, :K]
        b = torch.randn(N, K + 3, device=DEV)[:, :K]
        bias = torch.randn(N, device=DEV)
        want = a.double() @ b.double().t() + bias.double()
        got = tcommon._mm_nt(a, b, bias=bias)
        assert float((got.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())) * K ** 0.5
        b2 = torch.randn(K, N + 2, device=DEV)[:, :N]
        want = a.double() @ b2.double()
        got = tcommon._mm_nn(a, b2)
        assert float((got.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())) * K ** 0.5
        c = torch.randn(M, N, device=DEV)
        want = a.double().t() @ c.double()
        got = tcommon._tn_matmul(a, c)
        assert float((got.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())) * M ** 0.5


EB_CASES = [
    # B, O, P, cin, pt dims, C
    (2, 300, 5, 131, [128]),
    (2, 40, 32, 67, [64, 64, 128]),
    (1, 33, 128, 3, [32, 32, 64]),
    (3, 50, 7, 35, [32, 64]),
]


@pytest.mark.parametrize("B,O,P,cin,dims", EB_CASES,
                         ids=["%dx%dx%d_%d" % (c[0], c[1], c[2], c[3]) for c in EB_CASES])
def test_edge_block_train_matches_torch(B, O, P, cin, dims):
    """max_p att_mlp(att_vec) * pt_mlp(nf): fused (pairmax + sparse-gradient backward) vs modules."""
    torch.manual_seed(B * O + P)
    C = dims[-1]
    pt_ref = mlp(cin, dims).to(DEV).train()
    a1_ref, a2_ref = mlp(10, [C // 4]).to(DEV).train(), mlp(C // 4, [C]).to(DEV).train()
    for m in list(pt_ref.modules()) + list(a1_ref.modules()) + list(a2_ref.modules()):
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0.2, 0.3)
    pt_new, a1_new, a2_new = (copy.deepcopy(m) for m in (pt_ref, a1_ref, a2_ref))
    nf1 = torch.randn(B, O, P, cin, device=DEV).requires_grad_(True)
    nf2 = nf1.detach().clone().requires_grad_(True)
    av = torch.randn(B, O, P, 10, device=DEV)
    y1 = (a2_ref(a1_ref(av)) * pt_ref(nf1)).max(dim=2).values
    assert tedge.edge_block_supported(list(pt_new), [a1_new[0], a2_new[0]], nf2)
    y2 = tedge.edge_block_train(nf2, av, list(pt_new), [a1_new[0], a2_new[0]])
    s = max(1.0, float(y1.detach().abs().max()))
    assert float((y1 - y2).abs().max()) <= 3e-5 * s
    g = torch.randn_like(y1)
    y1.backward(g)
    y2.backward(g)

    def close(a, b, tol=5e-4):
        sc = max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol * sc, (float((a - b).abs().max()), sc)
    close(nf2.grad, nf1.grad)
    for ref, new in ((pt_ref, pt_new), (a1_ref, a1_new), (a2_ref, a2_new)):
        for (n1, p1), (n2, p2) in zip(ref.named_parameters(), new.named_parameters()):
            if not n1.endswith("lin.bias"):
                close(p2.grad, p1.grad)
        for (n1, b1), (n2, b2) in zip(ref.named_buffers(), new.named_buffers()):
            if "num_batches" not in n1:
                close(b2, b1, 1e-5)


@pytest.mark.parametrize("C,cin", [(32, 3), (16, 10), (128, 131), (256, 128), (64, 67), (128, 260),
                                   (4, 5), (256, 384)])
def test_pack_linear_layouts(C, cin):
    """gridgcn_pack_linear (one launch) == the three layouts built with torch ops, bit for bit."""
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _ptr, _stream, pack_conv_layer
    lib = _lib.load()
    torch.manual_seed(C * 1000 + cin)
    W = torch.randn(C, cin, device=DEV)
    b = torch.randn(C, device=DEV)
    K, ldw, nwp, nwb = tcommon.packed_sizes(C, cin)
    Wp, Bp = torch.full((nwp,), 7.0, device=DEV), torch.full((ldw,), 7.0, device=DEV)
    Wb, Wg = torch.full((nwb,), 7.0, device=DEV), torch.full((nwb,), 7.0, device=DEV)
    rc = lib.gridgcn_pack_linear(_ptr(W), _ptr(b), C, cin, 0, cin, 0, _ptr(Wp), _ptr(Bp), _ptr(Wb),
                                 _ptr(Wg), None, None, _stream(W))
    assert rc == 0
    rWp, rBp, rK, rldw, _ = pack_conv_layer(W.t(), b)
    assert (rK, rldw) == (K, ldw)
    assert torch.equal(Wp, rWp.reshape(-1)) and torch.equal(Bp, rBp)
    assert torch.equal(Wb, tcommon.pack_tiles(W).reshape(-1))
    assert torch.equal(Wg, tcommon.pack_groups(W))
    # rotated + zero padded columns == packing the explicitly permuted matrix
    if cin > 3:
        cinp = (cin + 7) & ~7
        Wperm = torch.zeros(C, cinp, device=DEV)
        Wperm[:, :cin - 3] = W[:, 3:]
        Wperm[:, cin - 3:cin] = W[:, :3]
        K2, ldw2, nwp2, nwb2 = tcommon.packed_sizes(C, cinp)
        outs = [torch.full((n,), 7.0, device=DEV) for n in (nwp2, nwb2, nwb2, cinp * ldw2)]
        refs = [torch.full((n,), 7.0, device=DEV) for n in (nwp2, nwb2, nwb2, cinp * ldw2)]
        for src, rot, cw, dst in ((W, 3, cin, outs), (Wperm, 0, cinp, refs)):
            rc = lib.gridgcn_pack_linear(_ptr(src), _ptr(b), C, cw, rot, cinp, 0, _ptr(dst[0]),
                                         _ptr(Bp), _ptr(dst[1]), _ptr(dst[2]), _ptr(dst[3]), None,
                                         _stream(W))
            assert rc == 0
        for o, r in zip(outs, refs):
            assert torch.equal(o, r)


@pytest.mark.parametrize("R,Cf,C0,rot", [(8192, 128, 128, 3), (2048, 64, 64, 3), (192, 256, 128, 3),
                                         (1000, 8, 36, 0), (33, 136, 4, 1), (40000, 32, 64, 3)])
def test_gemm_small_matches_torch(R, Cf, C0, rot):
    """gridgcn_gemm_small (csrc/gridgcn_gemm.hip), the three products of the source-point conv on column
    slices of wider tensors (row strides 4 + Cf and rot + Cf, not 16-byte aligned) against torch.matmul
    in float64; fp32 MFMA accumulation: 1e-5 of the largest entry times sqrt(K)."""
    torch.manual_seed(R + Cf)
    src = torch.randn(R, 4 + Cf, device=DEV)
    W0 = torch.randn(C0, rot + Cf, device=DEV)
    feat, Wf = src[:, 4:], W0[:, rot:]
    Y = tcommon._gemm_small(0, feat, Wf, torch.empty(R, C0, device=DEV), R, C0, Cf)
    ref = feat.double() @ Wf.double().t()
    assert float((Y - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * Cf ** 0.5
    if C0 % 8 == 0:
        dY = torch.randn(R, C0, device=DEV)
        g = torch.full((R, 4 + Cf), 7.0, device=DEV)
        tcommon._gemm_small(1, dY, Wf, g[:, 4:], R, Cf, C0, zero_left=4)
        ref = dY.double() @ Wf.double()
        assert float((g[:, 4:] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * C0 ** 0.5
        assert float(g[:, :4].abs().max()) == 0.0
    dY = torch.randn(R, C0, device=DEV)
    dW = torch.full((C0, rot + Cf), 7.0, device=DEV)
    tcommon._tn_matmul(dY, feat, out=dW[:, rot:])
    ref = dY.double().t() @ feat.double()
    assert float((dW[:, rot:] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * R ** 0.5
    assert rot == 0 or bool((dW[:, :rot] == 7.0).all())
    G = torch.randn(R, 4, device=DEV)
    t = tcommon._tn_matmul(Y, G)
    ref = Y.double().t() @ G.double()
    assert t.shape == (C0, 4) and float((t - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * R ** 0.5
    # bit-reproducible (fixed summation order), and the workspace's tickets are back at zero
    for _ in range(3):
        assert torch.equal(t, tcommon._tn_matmul(Y, G))


@pytest.mark.parametrize("ncent,P", [(655, 5), (64, 7), (2000, 5), (33, 1), (4096, 5), (70000, 5), (13, 3)])
def test_att_bwd_noz_round5_tile_loop_is_word_identical(ncent, P):
    """gg_k_att_bwd_nz2 (GRIDGCN_OPT_ATT_NZ_V2, the default: no per-tile divisions, range-checked buffer streams,
    no predicates on the partial last tile) against the round-4 kernel on the same inputs: dX and dW bit for bit
    (same arithmetic in the same order), the fp64 sums to their atomics' ordering."""
    import ctypes
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _ptr, _stream
    lib = _lib.load()
    OPT = 6                                   # GRIDGCN_OPT_ATT_NZ_V2
    cin, C = 32, 128
    E = ncent * P
    g = torch.Generator(device=DEV).manual_seed(ncent * 31 + P)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    Z1 = rnd(E, cin)
    s1v, h1v, m1v, r1v = rnd(cin).abs() + 0.5, rnd(cin) * 0.1, rnd(cin) * 0.1, rnd(cin).abs() + 0.5
    W2, b2 = rnd(C, cin) * 0.2, rnd(C) * 0.1
    s2v, m2v, r2v = rnd(C).abs() + 0.5, rnd(C) * 0.1, rnd(C).abs() + 0.5
    sums_a = rnd(2 * C).double()
    amax = torch.randint(0, P, (ncent, C), device=DEV, dtype=torch.int32, generator=g).to(torch.uint8)
    ga = rnd(ncent, C)
    nbytes = ctypes.c_size_t(0)
    assert lib.gridgcn_att_bwd_noz_workspace_bytes(E, cin, C, ctypes.byref(nbytes)) == 0
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=DEV)
    res = []
    try:
        for v2 in (1, 0):
            assert lib.gridgcn_set_option(OPT, v2) == 0 and lib.gridgcn_get_option(OPT) == v2
            dA1 = torch.full((E + 64, cin), 7.0, device=DEV)          # (guard rows: nothing may be written past E)
            dW2 = torch.empty(C, cin, device=DEV)
            v = torch.empty(4, C, device=DEV)
            acc = torch.zeros(3 * cin, dtype=torch.float64, device=DEV)
            rc = lib.gridgcn_att_bwd_noz(_ptr(Z1), _ptr(s1v), _ptr(h1v), _ptr(m1v), _ptr(r1v), _ptr(W2), _ptr(b2),
                                         _ptr(s2v), _ptr(m2v), _ptr(r2v), _ptr(sums_a), _ptr(amax), _ptr(ga), int(P),
                                         E, cin, C, _ptr(dA1), _ptr(dW2), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]),
                                         _ptr(v[3]), _ptr(acc[:2 * cin]), _ptr(acc[2 * cin:]), _ptr(ws),
                                         nbytes.value, _stream(Z1))
            assert rc == 0
            res.append((dA1, dW2, v, acc))
    finally:
        lib.gridgcn_set_option(OPT, 1)
    (x1, w1, v1, a1), (x0, w0, v0, a0) = res
    assert bool((x1[E:] == 7.0).all()) and bool((x0[E:] == 7.0).all())
    assert torch.equal(x1[:E], x0[:E])
    assert torch.equal(w1, w0) and torch.equal(v1, v0)
    assert float((a1 - a0).abs().max()) <= 1e-12 * max(1.0, float(a0.abs().max()))
    assert float(x0[:E].abs().max()) > 0 and bool(torch.isfinite(x1[:E]).all())


def _att_fwd_inputs(B, Nsrc, O, seed, geo=True):
    P, cin, C = 5, 32, 128
    ncent, E, R = B * O, B * O * 5, B * Nsrc
    g = torch.Generator(device=DEV).manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    d = dict(P=P, cin=cin, C=C, ncent=ncent, E=E, R=R)
    d["Ysrc"] = rnd(R, C)
    # (indices one past either end: take mode 'clip')
    d["nebidx"] = torch.randint(-1, Nsrc + 1, (B, O, P), device=DEV, dtype=torch.int32, generator=g)
    d["att16"] = rnd(E, 16)
    d["Wg"] = rnd(3, C) * 0.3 if geo else None
    d["b"] = rnd(C) * 0.1
    d["Z1"] = rnd(E, cin)
    d["s1"], d["h1"] = rnd(cin).abs() + 0.5, rnd(cin) * 0.3
    d["W2"], d["b2"] = rnd(C, cin) * 0.2, rnd(C) * 0.1
    d["scp"], d["shp"] = rnd(C).abs() + 0.5, rnd(C) * 0.3
    d["gamma"], d["beta"] = rnd(C).abs() + 0.5, rnd(C) * 0.3
    return d


@pytest.mark.parametrize("E", [32, 33, 511, 512, 8191, 70001, 1300000])
def test_att_bn2_moments_match_the_statistics_of_the_materialised_conv(E):
    """gridgcn_att_bn2_moments (csrc/gridgcn_attfwd.hip): the BatchNorm of z2 = W2 a1 + b2 from S1 = sum a1 and
    S2 = sum a1 a1^T, against float64 statistics of the materialised z2.  Bars: mean 2e-6 of the largest |mean| (or
    of the spread), rstd 5e-6 relative -- the S2 products are fp32 MFMA terms, folded into fp64 every 256 rows;
    measured 3e-7 / 8e-7 at E = 1.3 M.  Running estimates and the step counter as gridgcn_bn_finalize writes them;
    bit-reproducible (fixed summation order)."""
    import ctypes
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _ptr, _stream
    lib = _lib.load()
    cin, C, eps, mom = 32, 128, 1e-5, 0.1
    g = torch.Generator(device=DEV).manual_seed(E)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    Z1 = rnd(E, cin)
    s1, h1 = rnd(cin).abs() + 0.5, rnd(cin) * 0.3
    W2, b2 = rnd(C, cin) * 0.2, rnd(C) * 0.5
    gamma, beta = rnd(C).abs() + 0.5, rnd(C) * 0.3
    nbytes = ctypes.c_size_t(0)
    assert lib.gridgcn_att_fwd_noz_workspace_bytes(E, cin, C, ctypes.byref(nbytes)) == 0
    assert lib.gridgcn_att_fwd_noz_workspace_bytes(E, 16, C, ctypes.byref(nbytes)) != 0
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=DEV)
    outs = []
    for rep in range(2):
        vec = torch.empty(4, C, device=DEV)
        rm, rv = torch.full((C,), 0.25, device=DEV), torch.full((C,), 2.0, device=DEV)
        nbt = torch.full((1,), 41, dtype=torch.int64, device=DEV)
        sums = torch.empty(2 * C, dtype=torch.float64, device=DEV)
        rc = lib.gridgcn_att_bn2_moments(_ptr(Z1), _ptr(s1), _ptr(h1), _ptr(W2), _ptr(b2), _ptr(gamma), _ptr(beta), E,
                                         cin, C, eps, mom, _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]), _ptr(vec[3]),
                                         _ptr(rm), _ptr(rv), _ptr(nbt), _ptr(sums), _ptr(ws), nbytes.value, _stream(Z1))
        assert rc == 0
        outs.append((vec, rm, rv, nbt, sums))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    vec, rm, rv, nbt, sums = outs[0]
    a1 = torch.clamp_min(Z1 * s1 + h1, 0).double()
    z2 = a1 @ W2.double().t() + b2.double()
    mean, var = z2.mean(0), z2.var(0, unbiased=False)
    rstd = (var.float() + eps).rsqrt().double()
    spread = float(var.sqrt().max())
    assert float((vec[2].double() - mean).abs().max()) <= 2e-6 * max(float(mean.abs().max()), spread)
    assert float((vec[3].double() / rstd - 1).abs().max()) <= 5e-6
    sc = gamma.double() * rstd
    assert float((vec[0].double() - sc).abs().max()) <= 5e-6 * float(sc.abs().max())
    sh = beta.double() - mean * sc
    assert float((vec[1].double() - sh).abs().max()) <= 1e-5 * max(1.0, float(sh.abs().max()))
    assert float((sums[:C] - z2.sum(0)).abs().max()) <= 2e-6 * float(z2.abs().sum(0).max())
    assert float((sums[C:] / (z2 * z2).sum(0) - 1).abs().max()) <= 5e-6
    assert int(nbt) == 42
    unb = var * (E / max(E - 1, 1))
    assert float((rm.double() - (0.9 * 0.25 + 0.1 * mean)).abs().max()) <= 1e-5
    assert float((rv.double() - (0.9 * 2.0 + 0.1 * unb)).abs().max()) <= 1e-5 * max(1.0, float(unb.max()))


@pytest.mark.parametrize("B,Nsrc,O,geo", [(1, 40, 7, True), (2, 150, 33, True), (3, 150, 700, False), (1, 5000, 4099, True),
                                          (2, 20000, 65536, True)])
def test_att_pairmax_fwd_matches_the_materialised_path(B, Nsrc, O, geo):
    """gridgcn_att_pairmax_fwd (the second attention conv recomputed per 30-edge MFMA tile, csrc/gridgcn_attfwd.hip)
    against gridgcn_linear_fwd-style materialisation + gridgcn_pairmax_fwd_src on the same inputs and the SAME
    BatchNorm vectors.  The point branch is bit-identical; an attention value may differ in its last bits (bias first,
    MFMA order), so: agg to 2e-6 of its largest entry, the arg max equal except at near ties (an entry whose arg max
    differs must have a runner-up within 1e-5), zsel at equal arg max: point row bit-equal, attention row 2e-6.
    Against a float64 restatement as well.  Cases: one tile + one centre, tiles ending mid-way (ncent % 6 = 3, 0,
    1, 4), out-of-range neighbour indices (clipped), no geo_vec weights, an agg that is the left half of a wider
    buffer."""
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _ptr, _stream
    lib = _lib.load()
    d = _att_fwd_inputs(B, Nsrc, O, B * 1000 + O, geo)
    P, cin, C, ncent, E = d["P"], d["cin"], d["C"], d["ncent"], d["E"]
    st = _stream(d["Z1"])
    a1 = torch.clamp_min(d["Z1"] * d["s1"] + d["h1"], 0)
    z2 = a1.double() @ d["W2"].double().t() + d["b2"].double()
    mean, var = z2.mean(0), z2.var(0, unbiased=False)
    sca = (d["gamma"].double() * (var + 1e-5).rsqrt()).float()
    sha = (d["beta"].double() - mean * sca.double()).float()
    Z2 = z2.float().contiguous()
    lda = 256
    wide = [torch.full((ncent + 3, lda), 7.0, device=DEV) for _ in range(2)]
    amax = [torch.full((ncent + 3, C), 99, dtype=torch.uint8, device=DEV) for _ in range(2)]
    zsel = [torch.full((2, ncent, C), 7.0, device=DEV) for _ in range(2)]
    wg = _ptr(d["Wg"]) if geo else None
    rc = lib.gridgcn_att_pairmax_fwd(_ptr(d["Ysrc"]), _ptr(d["nebidx"]), _ptr(d["att16"]), wg, _ptr(d["b"]), B, Nsrc, O,
                                     _ptr(d["Z1"]), _ptr(d["s1"]), _ptr(d["h1"]), _ptr(d["W2"]), _ptr(d["b2"]),
                                     _ptr(d["scp"]), _ptr(d["shp"]), _ptr(sca), _ptr(sha), ncent, P, cin, C,
                                     _ptr(wide[0]), lda, _ptr(amax[0]), _ptr(zsel[0]), st)
    assert rc == 0
    rc = lib.gridgcn_pairmax_fwd_src_z(_ptr(d["Ysrc"]), _ptr(d["nebidx"]), _ptr(d["att16"]), wg, _ptr(d["b"]), B, Nsrc,
                                       O, _ptr(Z2), 0, _ptr(d["scp"]), _ptr(d["shp"]), _ptr(sca), _ptr(sha), ncent, P,
                                       C, _ptr(wide[1]), lda, _ptr(amax[1]), _ptr(zsel[1]), st)
    assert rc == 0
    for w, am in zip(wide, amax):
        assert bool((w[:, C:] == 7.0).all()) and bool((w[ncent:] == 7.0).all()) and bool((am[ncent:] == 99).all())
    agg1, agg0 = wide[0][:ncent, :C], wide[1][:ncent, :C]
    top = float(agg0.abs().max())
    assert float((agg1 - agg0).abs().max()) <= 2e-6 * top
    # float64 restatement of the whole thing
    flat = (d["nebidx"].long() + (torch.arange(B, device=DEV) * Nsrc)[:, None, None]).clamp(0, B * Nsrc - 1)
    zp = d["Ysrc"].double()[flat.reshape(-1)]
    if geo:
        zp = zp + d["att16"][:, 1:4].double() @ d["Wg"].double()
    zp = zp + d["b"].double()
    v = (torch.clamp_min(zp * d["scp"].double() + d["shp"].double(), 0)
         * torch.clamp_min(z2 * sca.double() + sha.double(), 0)).reshape(ncent, P, C)
    ref = v.max(1).values
    assert float((agg1.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    same = amax[0][:ncent] == amax[1][:ncent]
    if not bool(same.all()):
        srt = v.sort(1, descending=True).values
        gap = (srt[:, 0] - srt[:, 1])[~same]
        assert float(gap.max()) <= 1e-5 * float(ref.abs().max())
        assert float((~same).float().mean()) < 1e-3
    assert torch.equal(zsel[0][0][same], zsel[1][0][same])
    assert float((zsel[0][1][same] - zsel[1][1][same]).abs().max()) <= 2e-6 * float(z2.abs().max())
    # shapes the kernel declines: the callers keep the Z2 path
    args = [_ptr(d["Ysrc"]), _ptr(d["nebidx"]), _ptr(d["att16"]), wg, _ptr(d["b"]), B, Nsrc, O, _ptr(d["Z1"]),
            _ptr(d["s1"]), _ptr(d["h1"]), _ptr(d["W2"]), _ptr(d["b2"]), _ptr(d["scp"]), _ptr(d["shp"]), _ptr(sca),
            _ptr(sha), ncent]
    tail = [_ptr(wide[0]), lda, _ptr(amax[0]), _ptr(zsel[0]), st]
    assert lib.gridgcn_att_pairmax_fwd(*args, 7, cin, C, *tail) != 0
    assert lib.gridgcn_att_pairmax_fwd(*args, P, 16, C, *tail) != 0
    assert lib.gridgcn_att_pairmax_fwd(*args, P, cin, 64, *tail) != 0
    assert lib.gridgcn_att_pairmax_fwd(*args, P, cin, C, _ptr(wide[0]), lda, _ptr(amax[0]), None, st) != 0


@pytest.mark.parametrize("E,cin,prev_bn,nbn", [(32768, 128, True, 0), (65536, 256, True, 128), (40960, 128, False, 0),
                                               (131072, 256, True, 0)])
def test_linear_bwd_fused128_matches_separate_kernels(E, cin, prev_bn, nbn):
    """csrc/gridgcn_bwdfused.hip (GRIDGCN_OPT_BWD_FUSED128, round 5): dX, dW and the BatchNorm-backward sums of the
    layer in front from ONE pass over Z and dY, against the register-direct dX + dW kernels on the same inputs.
    Same terms, other summation orders: dX to 2e-6 of its largest entry (one association of the 128-channel sum
    differs), dW 1e-5 (fp32 sums over E rows), the sums (fp64 atomics of fp32 partials) 1e-6.
    Both against a float64 restatement as well."""
    import ctypes
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _ptr, _stream
    lib = _lib.load()
    OPT = 7
    C = 128
    g = torch.Generator(device=DEV).manual_seed(E + cin + nbn)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    Z, X, dY = rnd(E, C), rnd(E, cin), rnd(E, C)
    scale, shift = rnd(C).abs() + 0.5, rnd(C) * 0.1
    mean, rstd = rnd(C) * 0.1, rnd(C).abs() + 0.5
    sums = (rnd(2 * C) * 1e-3 * E).double()
    Wt = rnd(C, cin) * 0.1
    Wb, Wg = tcommon.pack_tiles(Wt), tcommon.pack_groups(Wt)
    Wdx = torch.empty(C * 32 * (4 if cin == 128 else 8), device=DEV)
    st = _stream(Z)
    assert lib.gridgcn_pack_linear(_ptr(Wt), None, C, cin, 0, cin, cin, None, None, None, None, None, _ptr(Wdx), st) == 0
    pv = [rnd(cin).abs() + 0.5, rnd(cin) * 0.1, rnd(cin) * 0.1, rnd(cin).abs() + 0.5]
    if nbn:                      # RawLink: identity on the columns beyond nbn
        pv[0][nbn:] = 1.0
        pv[1][nbn:] = 0.0
        X[:, nbn:].abs_()
    pb = [_ptr(t) for t in pv] if prev_bn else [None] * 4
    nbytes = ctypes.c_size_t(0)
    lib.gridgcn_linear_bwd_workspace_bytes(E, cin, C, ctypes.byref(nbytes))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=DEV)
    res = []
    try:
        for fused in (1, 0):
            assert lib.gridgcn_set_option(OPT, fused) == 0
            dX = torch.empty(E, cin, device=DEV)
            dW = torch.full((C, cin), 7.0, device=DEV)
            v = torch.empty(4, C, device=DEV)
            psums = torch.zeros(2 * (nbn or cin), dtype=torch.float64, device=DEV)
            rc = lib.gridgcn_linear_bwd_fin(
                _ptr(dY), _ptr(Z), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(sums), _ptr(v[0]), _ptr(v[1]),
                _ptr(v[2]), _ptr(v[3]), _ptr(X), pb[0], pb[1], pb[2], pb[3], _ptr(Wb), _ptr(Wg), _ptr(Wdx), cin, E, C,
                cin, cin, 0, C, 0, nbn, 0, _ptr(dX), _ptr(dW), _ptr(psums) if prev_bn else None, None, None, 0,
                _ptr(ws), nbytes.value, st)
            assert rc == 0
            res.append((dX, dW, v, psums))
    finally:
        lib.gridgcn_set_option(OPT, 1)
    (x1, w1, v1, p1), (x0, w0, v0, p0) = res
    # (dX: the two channel halves of a row tile are contracted by two waves and added once -- one fp32 association
    #  differs from the 128-step chain of the separate kernel)
    assert float((x1 - x0).abs().max()) <= 2e-6 * float(x0.abs().max())
    assert torch.equal(v1, v0)
    assert float((w1 - w0).abs().max()) <= 1e-5 * float(w0.abs().max())
    if prev_bn:
        assert float((p1 - p0).abs().max()) <= 1e-6 * float(p0.abs().max())
    # float64 restatement
    m1, m2 = sums[:C] / E, sums[C:] / E
    y = Z.double() * scale.double() + shift.double()
    dz = scale.double() * torch.where(y > 0, dY.double(), torch.zeros_like(y)) + \
        (Z.double() - mean.double()) * (-(scale.double() * rstd.double()) * m2) - scale.double() * m1
    act = torch.relu(X.double() * pv[0].double() + pv[1].double()) if prev_bn else X.double()
    refW = dz.t() @ act
    assert float((w1 - refW).abs().max()) <= 2e-5 * float(refW.abs().max())
    refX = dz @ Wt.double()
    assert float((x1 - refX).abs().max()) <= 2e-5 * float(refX.abs().max())


def test_tn_matmul_tall_product_bounded_workspace():
    """the transposed product's workspace used to grow with the row count (tiles x ceil(R/256) x 4 KB:
    2.6 GB for a [512 x 320] product over 10^6 rows).  The library now caps tiles x slices at 4096 partial tiles
    and a workgroup walks several 256-row sub-slices instead: same result, workspace <= ~16 MB."""
    import ctypes
    from grid_gcn_amd import _lib
    lib = _lib.load()
    n = ctypes.c_size_t(0)
    lib.gridgcn_gemm_small_workspace_bytes(512, 320, 1 << 20, ctypes.byref(n))
    assert n.value <= (17 << 20)
    torch.manual_seed(11)
    R, m, k = 300000 + 77, 96, 160          # 15 tiles -> 273 slices allowed, 1172 sub-slices: 5 per workgroup
    a, b = torch.randn(R, m, device=DEV), torch.randn(R, k, device=DEV)
    lib.gridgcn_gemm_small_workspace_bytes(m, k, R, ctypes.byref(n))
    assert n.value <= (17 << 20)
    t = tcommon._tn_matmul(a, b)
    ref = a.double().t() @ b.double()
    assert float((t - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * R ** 0.5
    assert torch.equal(t, tcommon._tn_matmul(a, b))


def test_pack_cache_batch_launch_equals_single_packs():
    """tcommon.PACKS: the one-launch rebuild of every layout of a module
    (gridgcn_pack_linear_batch) writes, bit for bit, what the per-layer gridgcn_pack_linear writes;
    after it each entry serves ONE lookup without a launch, every other lookup packs on its own."""
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _stream
    lib = _lib.load()
    torch.manual_seed(5)
    cache = tcommon._PackCache()
    shapes = [(32, 11, 3, 16, 0), (64, 32, 0, 32, 32), (128, 131, 3, 136, 131),
              (256, 128, 0, 128, 128), (13, 128, 0, 128, 128), (128, 256, 0, 256, 256)]
    holder = torch.nn.Module()
    calls = []
    for i, (C, cin_w, rot, cin, ndx) in enumerate(shapes):
        W = torch.nn.Parameter(torch.randn(C, cin_w, device=DEV))
        b = torch.nn.Parameter(torch.randn(C, device=DEV))
        holder.register_parameter("w%d" % i, W)
        holder.register_parameter("b%d" % i, b)
        K, ldw, nwp, nwb = tcommon.packed_sizes(C, cin)
        Cp = (C + 7) & ~7
        nt = (ndx + 31) // 32
        ntv = 1 if nt <= 1 else 2 if nt <= 2 else 4 if nt <= 4 else 8
        sizes = (nwp, ldw, nwb, cin * ldw, Cp * 32 * ntv if ndx else 0)
        calls.append((lib, W, b, C, cin_w, rot, cin, ndx, True, sizes, _stream(W)))
        cache.get(*calls[-1])
    assert len(cache.entries) == len(shapes)
    ent = list(cache.entries.values())

    def fill():
        for e in ent:
            e["pk"].fill_(7.0)

    def untouched():
        torch.cuda.synchronize()
        return [bool((e["pk"] == 7.0).all()) for e in ent]

    fill()
    for c in calls:
        cache.get(*c)                         # no prepack before it: the per-layer launch
    assert not any(untouched())
    singles = [e["pk"].clone() for e in ent]
    fill()
    cache.prepack(holder)                     # one launch for all six
    torch.cuda.synchronize()
    for e, ref in zip(ent, singles):
        assert torch.equal(e["pk"], ref)
    fill()
    for c in calls:
        cache.get(*c)                         # served from the batch: nothing is launched
    assert all(untouched())
    for c in calls[:2]:
        cache.get(*c)                         # second lookup in the same forward: packs again
    assert untouched() == [False, False, True, True, True, True]
    # a weight whose version moved after the batch launch is not served from it
    fill()
    cache.prepack(holder)
    with torch.no_grad():
        calls[3][1].add_(1.0)
    fill()
    cache.get(*calls[3])
    cache.get(*calls[4])
    assert untouched() == [True, True, True, False, True, True]
    # release(): what no layer looked up does not stay fresh
    cache.release(holder)
    cache.get(*calls[5])
    assert untouched()[5] is False
    # a dead weight drops its entry (and the module's table with it)
    del holder._parameters["w5"], calls[5], c, W
    import gc
    gc.collect()
    assert len(cache.entries) == len(shapes) - 1 and not cache.tables


@pytest.mark.parametrize("E,C", [(1000, 21), (70001, 21), (257, 8), (5000, 32), (300, 3)])
def test_softmax_ce_matches_torch(E, C):
    """gridgcn_softmax_ce_fwd/bwd == F.cross_entropy(ignore_index=0, reduction='mean')
    (SoftmaxOutput use_ignore / normalization='valid', ggcn_models_g.py:41)."""
    import torch.nn.functional as F
    torch.manual_seed(E + C)
    x1 = (torch.randn(E, C, device=DEV) * 3).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    lab = torch.randint(0, C, (E,), device=DEV)
    l1 = F.cross_entropy(x1, lab, ignore_index=0, reduction="mean")
    l2 = thead.softmax_ce(x2, lab, 0)
    assert abs(float(l1) - float(l2)) <= 2e-6 * max(1.0, abs(float(l1)))
    (l1 * 1.7).backward()
    (l2 * 1.7).backward()
    assert float((x1.grad - x2.grad).abs().max()) <= 1e-6 * float(x1.grad.abs().max()) + 1e-12
    assert float(x2.grad[lab == 0].abs().max()) == 0.0


@pytest.mark.parametrize("E,cin,C", [(4000, 128, 21), (70001, 128, 21), (333, 64, 13), (900, 256, 32)])
def test_linear_plain_and_loss_match_torch(E, cin, C):
    """the segmentation head: Linear(cin, C) on the MFMA kernels (class dim zero padded inside) +
    the fused loss, against torch.nn.Linear + F.cross_entropy: logits, loss, dX, dW, db."""
    import torch.nn.functional as F
    torch.manual_seed(E + cin + C)
    lin1 = torch.nn.Linear(cin, C).to(DEV)
    lin2 = copy.deepcopy(lin1)
    x1 = torch.randn(E, cin, device=DEV).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    lab = torch.randint(0, C, (E,), device=DEV)
    assert thead.linear_plain_supported(x2, lin2)
    y1 = lin1(x1)
    y2 = thead.linear_plain_train(x2, lin2)
    assert y2.shape == y1.shape
    assert float((y1 - y2).abs().max()) <= 2e-5 * max(1.0, float(y1.abs().max()))
    l1 = F.cross_entropy(y1, lab, ignore_index=0, reduction="mean")
    l2 = thead.softmax_ce(y2, lab, 0)
    assert abs(float(l1) - float(l2)) <= 1e-5
    l1.backward()
    l2.backward()

    def close(a, b, tol=2e-4):
        s = max(1e-6, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol * s, (float((a - b).abs().max()), s)
    close(x2.grad, x1.grad)
    close(lin2.weight.grad, lin1.weight.grad)
    close(lin2.bias.grad, lin1.bias.grad)
    # a generic (non padded) upstream gradient takes the copy path
    x3 = x1.detach().clone().requires_grad_(True)
    lin3 = copy.deepcopy(lin1)
    lin3.zero_grad()
    g = torch.randn(E, C, device=DEV)
    thead.linear_plain_train(x3, lin3).backward(g)
    x4 = x1.detach().clone().requires_grad_(True)
    lin1.zero_grad()
    lin1(x4).backward(g)
    close(x3.grad, x4.grad)
    close(lin3.weight.grad, lin1.weight.grad)
    close(lin3.bias.grad, lin1.bias.grad)


@pytest.mark.parametrize("E,cin,dims", [(5000, 4, [128]), (4097, 256, [128, 128]), (300, 67, [64, 32]),
                                        (70001, 128, [128, 256])])
def test_mlp_eval_matches_modules(E, cin, dims):
    """inference through the forward kernel with running statistics == the stock modules in eval()"""
    torch.manual_seed(E + cin)
    ref = mlp(cin, dims).to(DEV)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
            m.running_mean.normal_(0, 0.5)
            m.running_var.uniform_(0.5, 2.0)
    ref.eval()
    x = torch.randn(E, cin, device=DEV) * 1.5
    with torch.no_grad():
        y1 = ref(x)
        y2 = teval.mlp_bn_relu_eval(x, list(ref))
    assert y1.shape == y2.shape
    assert float((y1 - y2).abs().max()) <= 2e-5 * max(1.0, float(y1.abs().max()))


def test_unsupported_width_falls_to_modules():
    """stacks outside every kernel path (no BatchNorm; a width below 256 that does not divide
    256): stock modules, with a warning"""
    from grid_gcn_amd.gridconv import ConvBNReLU, run_mlp
    x = torch.randn(10, 8, device=DEV)
    for m in (torch.nn.Sequential(ConvBNReLU(8, 64, use_bn=False)), mlp(8, [48])):
        m = m.to(DEV).train()
        assert not tcommon.supported(list(m), x) and not tmlp.wide_supported(list(m), x)
        import grid_gcn_amd.gridconv as gcv
        gcv._warned_shapes.clear()
        with pytest.warns(RuntimeWarning):
            assert run_mlp(list(m), x).shape[0] == 10


@pytest.mark.parametrize("E,cin,dims", [(3000, 259, [256, 256, 512]), (1500, 1027, [512, 512]),
                                        (700, 8, [768]), (2000, 128, [512])])
def test_wide_stack_rocblas_plus_bn_kernels_matches_stock(E, cin, dims):
    """stacks beyond the MFMA kernels' widths (the 512-wide last layer of the classifier and of the
    200k-point workload): rocBLAS GEMMs + this library's BatchNorm kernels (tmlp.mlp_wide_train,
    with the supported sub-runs still on the MFMA chain) == the stock modules: forward, input and
    parameter gradients, running statistics."""
    import copy
    from grid_gcn_amd.gridconv import run_mlp
    torch.manual_seed(E + cin)
    ref = mlp(cin, dims).to(DEV).train()
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    new = copy.deepcopy(ref)
    assert not tcommon.supported(list(new), torch.empty(1, cin, device=DEV))
    x1 = (torch.randn(E, cin, device=DEV) * 1.5).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    y1 = ref(x1)
    y2 = run_mlp(list(new), x2)
    assert float((y1 - y2).abs().max()) <= 5e-5 * max(1.0, float(y1.abs().max()))
    g = torch.randn_like(y1)
    y1.backward(g)
    y2.backward(g)

    def close(a, b, tol=1e-3):
        s_ = max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol * s_, (float((a - b).abs().max()), s_)
    close(x2.grad, x1.grad)
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), new.named_parameters()):
        if n1.endswith("lin.bias"):
            continue
        close(p2.grad, p1.grad)
    for (n1, b1), (n2, b2) in zip(ref.named_buffers(), new.named_buffers()):
        if "num_batches" not in n1:
            close(b2, b1, 1e-5)


@pytest.mark.parametrize("E,cin,dims,C2,p", [(5000, 256, [128, 128], 21, 0.5), (70001, 128, [128], 21, 0.5),
                                             (333, 64, [64], 13, 0.3), (4097, 256, [128, 128], 21, 0.0)])
def test_head_train_matches_torch(E, cin, dims, C2, p):
    """fc1 chain -> Dropout(p) -> fc2 as one op (dropout folded into the neighbouring kernels)
    against the stock modules with the SAME mask (thead.dropout_mask regenerates it)."""
    torch.manual_seed(E + cin + C2)
    ref = mlp(cin, dims).to(DEV).train()
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    lin1 = torch.nn.Linear(dims[-1], C2).to(DEV)
    new, lin2 = copy.deepcopy(ref), copy.deepcopy(lin1)
    x1 = (torch.randn(E, cin, device=DEV) * 1.5).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    seed = 987654321012345 + E
    assert thead.head_supported(x2, list(new), lin2)
    mask = thead.dropout_mask(E, dims[-1], p, seed, DEV)
    keep = float((mask > 0).float().mean())
    assert abs(keep - (1.0 - p)) < 0.01, keep                  # the hash drops a fraction p
    import numpy as np
    assert set(mask.unique().tolist()) <= {0.0, float(np.float32(1.0 / (1.0 - float(np.float32(p)))))}
    if p > 0:                                                  # and its bits are not correlated
        m01 = (mask > 0).float()
        assert abs(float((m01[:, 1:] * m01[:, :-1]).mean()) - (1 - p) ** 2) < 0.01
        assert abs(float((m01[1:] * m01[:-1]).mean()) - (1 - p) ** 2) < 0.01
        assert not torch.equal(mask, thead.dropout_mask(E, dims[-1], p, seed + 1, DEV))
    y1 = lin1(ref(x1) * mask)
    y2 = thead.head_train(x2, list(new), p, lin2, seed)
    assert y1.shape == y2.shape
    assert float((y1 - y2).abs().max()) <= 2e-5 * max(1.0, float(y1.abs().max()))
    g = torch.randn_like(y1)
    y1.backward(g)
    y2.backward(g)

    def close(a, b, tol=2e-4):
        s = max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol * s, (float((a - b).abs().max()), s)
    s = max(1e-3, float(x1.grad.abs().max()))
    bad = ((x2.grad - x1.grad).abs().amax(dim=1) > 2e-4 * s)
    assert int(bad.sum()) <= max(1, E // 10000), (int(bad.sum()), E)      # (one such row whatever E)
    wtol = 2e-4 if E < 20000 else 5e-3
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), new.named_parameters()):
        if n1.endswith("lin.bias"):
            assert float(p2.grad.abs().max()) == 0.0
        else:
            close(p2.grad, p1.grad, wtol)
    close(lin2.weight.grad, lin1.weight.grad, wtol)
    close(lin2.bias.grad, lin1.bias.grad)
    for (n1, b1), (n2, b2) in zip(ref.named_buffers(), new.named_buffers()):
        if "num_batches" not in n1:
            close(b2, b1, 1e-5)


def test_head_fused_dropout_equals_stored_dropout(monkeypatch):
    """Dropout evaluated inside fc2's forward / dW kernels (OPT.FUSE_DROPOUT) against the path that
    stores the dropped activation: the same mask bit for bit, so outputs and gradients agree to rounding."""
    torch.manual_seed(11)
    E, C, C2, p, seed = 40003, 128, 21, 0.5, 1234567890123
    net = mlp(256, [C]).to(DEV).train()
    lin = torch.nn.Linear(C, C2).to(DEV)
    x = torch.randn(E, 256, device=DEV)
    g = torch.randn(E, C2, device=DEV)
    out = []
    for fuse in (True, False):
        monkeypatch.setattr(OPT, "FUSE_DROPOUT", fuse)
        n2, l2 = copy.deepcopy(net), copy.deepcopy(lin)
        xi = x.clone().requires_grad_(True)
        y = thead.head_train(xi, list(n2), p, l2, seed)
        y.backward(g)
        out.append((y.detach(), xi.grad, l2.weight.grad, l2.bias.grad, [q.grad for q in n2.parameters()]))
    (ya, xa, wa, ba, pa), (yb, xb, wb, bb, pb) = out
    assert float((ya - yb).abs().max()) <= 2e-6 * float(yb.abs().max())
    assert torch.equal(xa, xb)                     # (the dX kernel is the same launch in both)
    assert float((wa - wb).abs().max()) <= 1e-5 * float(wb.abs().max())
    assert torch.equal(ba, bb)
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) <= 1e-5 * max(1e-3, float(b.abs().max()))


def test_head_in_model_matches_unfused_head_without_dropout():
    """GGCNSeg with the fused head == the separate fc1 / dropout / fc2 ops when dropout is off."""
    from grid_gcn_amd import model, synth
    cfg = dict(model.SEG_8192, dropout=0.0)
    torch.manual_seed(3)
    net = model.GGCNSeg(cfg, fixed_seed=True).to(DEV).train()
    assert net.fused_head
    data, npn = synth.make_batch(2, 8192, "planes")
    x = torch.from_numpy(data[..., :3].copy()).to(DEV)
    n = torch.from_numpy(npn).to(DEV)
    lab = torch.randint(0, 21, (2, 8192), device=DEV)
    state = copy.deepcopy(net.state_dict())
    out = []
    for fused in (True, False):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        net.fused_head = fused
        logits = net(x, n)
        assert (net.last_tail_done == 2) == fused
        loss = model.seg_loss(logits, lab)
        loss.backward()
        out.append((logits.detach().clone(), float(loss),
                    {k: v.grad.detach().clone() for k, v in net.named_parameters()}))
    assert float((out[0][0] - out[1][0]).abs().max()) <= 2e-5 * max(1.0, float(out[1][0].abs().max()))
    assert abs(out[0][1] - out[1][1]) <= 1e-5
    for k, gref in out[1][2].items():
        sc = max(1e-6, float(gref.abs().max()))
        assert float((out[0][2][k] - gref).abs().max()) <= 2e-3 * sc + 1e-7, k


def test_softmax_ce_class_weights_match_weighted_gradient_op():
    """gridgcn_softmax_ce_bwd with class_weight == the reference's weighted_gradient op in front of
    SoftmaxOutput (custom_op/weighted_gradient.py), here model.WeightedGradient + F.cross_entropy."""
    import torch.nn.functional as F
    from grid_gcn_amd import model
    torch.manual_seed(5)
    E, C = 3001, 21
    lg1 = torch.randn(E, C, device=DEV).requires_grad_(True)
    lg2 = lg1.detach().clone().requires_grad_(True)
    lab = torch.randint(0, C, (E,), device=DEV)
    w = (torch.rand(C, device=DEV) * 2 + 0.25)
    l1 = F.cross_entropy(model.WeightedGradient.apply(lg1, w), lab, ignore_index=0)
    l2 = model.seg_loss(lg2, lab, weights=w)
    assert abs(float(l1) - float(l2)) <= 1e-5
    l1.backward()
    l2.backward()
    assert float((lg1.grad - lg2.grad).abs().max()) <= 1e-6 * float(lg1.grad.abs().max()) + 1e-12
    assert float(lg2.grad[lab == 0].abs().max()) == 0.0


@pytest.mark.parametrize("ncent,P,cin,C,sparse", [(700, 5, 32, 128, True), (1001, 12, 16, 64, True),
                                                  (4099, 1, 32, 64, False), (20000, 5, 32, 128, False)])
def test_att_bwd_fused_equals_separate_kernels(ncent, P, cin, C, sparse):
    """gg_k_att_bwd_fused (one pass over Z: dX, previous layer's BN-backward sums, dW) against the
    separate register-direct dX and dW kernels on the same inputs (GRIDGCN_OPT_ATT_BWD_FUSED = 0)."""
    import ctypes
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _ptr, _stream
    lib = _lib.load()
    E = ncent * P
    g = torch.Generator(device=DEV).manual_seed(ncent + C)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    Z, X = rnd(E, C), rnd(E, cin)
    scale, shift = rnd(C).abs() + 0.5, rnd(C) * 0.3
    mean, rstd = rnd(C) * 0.1, rnd(C).abs() + 0.5
    m1, m2 = rnd(C) * 1e-2, rnd(C) * 1e-2
    pv = [rnd(cin).abs() + 0.5, rnd(cin) * 0.3, rnd(cin) * 0.1, rnd(cin).abs() + 0.5]
    amax = torch.randint(0, P, (ncent, C), device=DEV, dtype=torch.int32, generator=g)
    amax8 = amax.to(torch.uint8)                # the kernels' one-byte arg max
    gval = rnd(ncent, C)
    dY = rnd(E, C)
    W = rnd(C, cin)
    Wb = tcommon.pack_tiles(W)
    Wdx = torch.empty(C * 32, device=DEV)
    _lib.check(lib.gridgcn_pack_linear(_ptr(W), None, C, cin, 0, cin, cin, None, None, None, None,
                                       None, _ptr(Wdx), _stream(W)), "pack")
    nbytes = ctypes.c_size_t(0)
    lib.gridgcn_linear_bwd_workspace_bytes(E, cin, C, ctypes.byref(nbytes))
    res = []
    for fused in (0, 1):
        _lib.check(lib.gridgcn_set_option(_lib.OPT_ATT_BWD_FUSED, fused), "gridgcn_set_option")
        dX = torch.full((E, cin), float("nan"), device=DEV)
        dW = torch.full((C, cin), float("nan"), device=DEV)
        psums = torch.zeros(2 * cin, dtype=torch.float64, device=DEV)
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=DEV)
        rc = lib.gridgcn_linear_bwd(
            None if sparse else _ptr(dY), _ptr(Z), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd),
            _ptr(m1), _ptr(m2), _ptr(X), _ptr(pv[0]), _ptr(pv[1]), _ptr(pv[2]), _ptr(pv[3]),
            _ptr(Wb), None, _ptr(Wdx), cin, E, C, cin, cin, 0, 0, _ptr(dX), _ptr(dW), _ptr(psums),
            _ptr(amax8) if sparse else None, _ptr(gval) if sparse else None, P if sparse else 0,
            _ptr(ws), nbytes.value, _stream(Z))
        _lib.check(rc, "gridgcn_linear_bwd")
        torch.cuda.synchronize()
        res.append((dX, dW, psums))
    assert lib.gridgcn_get_option(_lib.OPT_ATT_BWD_FUSED) == 1      # (left at its default)
    (x0, w0, s0), (x1, w1, s1) = res
    assert torch.isfinite(x1).all() and torch.isfinite(w1).all()
    # dX: the same MFMA chain over the channels in both kernels
    assert float((x0 - x1).abs().max()) <= 1e-6 * max(1.0, float(x0.abs().max()))
    # dW and the sums: different summation order over the rows
    assert float((w0 - w1).abs().max()) <= 2e-5 * max(1.0, float(w0.abs().max()))
    assert float((s0 - s1).abs().max()) <= 1e-5 * max(1.0, float(s0.abs().max()))


def test_bf16_mlp_precision_mode_close_to_fp32():
    """gridgcn_set_mlp_precision(1): bf16 MFMA operands, fp32 storage / accumulation / statistics.
    Tolerance of the variant (NOT the parity path, which stays fp32): output of a 3-layer
    conv+BN+ReLU stack within 2e-2 * max|y| of the fp32 kernels, input and weight gradients within
    1e-1 in relative L2 norm (measured: 6e-2 on the input gradient); and the switch really changes the arithmetic (the outputs differ)."""
    from grid_gcn_amd.train import common as tcommon, mlp as tmlp
    from grid_gcn_amd.gridconv import mlp
    torch.manual_seed(11)
    layers = mlp(136, [128, 128, 256]).to(DEV).train()
    x = torch.randn(40000, 136, device=DEV)
    res = {}
    try:
        for mode in ("fp32", "bf16"):
            tcommon.set_mlp_precision(mode)
            assert tcommon.get_mlp_precision() == mode
            for l in layers:
                l.zero_grad()
            xi = x.clone().requires_grad_(True)
            y = tmlp.mlp_bn_relu_train(xi, list(layers))
            (y * y).sum().backward()
            res[mode] = (y.detach().clone(), xi.grad.clone(),
                         [l.lin.weight.grad.clone() for l in layers])
    finally:
        tcommon.set_mlp_precision("fp32")
    y32, y16 = res["fp32"][0], res["bf16"][0]
    assert float((y32 - y16).abs().max()) > 0.0
    assert float((y32 - y16).abs().max()) <= 2e-2 * float(y32.abs().max())
    # gradients through three BatchNorm+ReLU layers: a ReLU that flips under the rounded operands
    # moves single elements by much more than the rounding itself -- bound the error in norm
    rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
    assert rel(res["bf16"][1], res["fp32"][1]) <= 1e-1
    for a, b in zip(res["fp32"][2], res["bf16"][2]):
        assert rel(b, a) <= 1e-1


def test_bf16_mode_wide_layer_close_to_fp32():
    """a 512-wide layer (256-column slices of the register-direct kernels) in the bf16 contraction mode:
    same stated tolerance as the narrow stacks above (cfg5's last layer; in fp32-only form it fell to the
    small-GEMM kernels in this mode: 27 instead of 20 ms per cfg5 step)"""
    from grid_gcn_amd.train import common as tcommon, mlp as tmlp
    from grid_gcn_amd.gridconv import mlp
    torch.manual_seed(3)
    net = mlp(128, [512, 256]).to(DEV).train()
    x = torch.randn(8192, 128, device=DEV)
    res = {}
    try:
        for mode in ("fp32", "bf16"):
            tcommon.set_mlp_precision(mode)
            n2 = copy.deepcopy(net)
            xx = x.clone().requires_grad_(True)
            assert tmlp.wide_supported(list(n2), xx)
            y = tmlp.mlp_wide_train(xx, list(n2))
            y.square().mean().backward()
            res[mode] = (y.detach(), xx.grad, [p.grad for p in n2.parameters()])
    finally:
        tcommon.set_mlp_precision("fp32")
    rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
    assert rel(res["bf16"][0], res["fp32"][0]) <= 2e-2
    assert rel(res["bf16"][1], res["fp32"][1]) <= 1e-1
    for a, b in zip(res["bf16"][2], res["fp32"][2]):
        if float(b.norm()) > 1e-6:
            assert rel(a, b) <= 1e-1


def test_bf16_mode_whole_model_loss_and_gradient_direction():
    """bf16 contraction mode on the whole segmentation network (hidden layers only: the first conv
    of every stack sees raw coordinates and stays fp32): the training loss moves by < 1e-3 relative;
    the gradient keeps its direction (cos > 0.9) -- it does not match closer than that because a
    0.4 % perturbation of the pair features flips the arg-max of the neighbour max-pool in a fraction
    of the (centre, channel) pairs, which re-routes their gradient to another edge."""
    import copy
    from grid_gcn_amd import model, synth
    from grid_gcn_amd.train import common as tcommon
    torch.manual_seed(3)
    cfg = dict(model.SEG_81920, dropout=0.0)
    net = model.GGCNSeg(cfg, fixed_seed=True).to(DEV).train()
    state = copy.deepcopy(net.state_dict())
    data, npn = synth.make_batch(2, 16384, "planes")
    x = torch.from_numpy(data[..., :3].copy()).to(DEV)
    n = torch.from_numpy(npn).to(DEV)
    lab = torch.randint(0, 21, (2, 16384), device=DEV)
    res = {}
    try:
        for mode in ("fp32", "bf16"):
            tcommon.set_mlp_precision(mode)
            net.load_state_dict(state)
            net.zero_grad()
            loss = model.seg_loss(net(x, n), lab)
            loss.backward()
            res[mode] = (float(loss), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double())
    finally:
        tcommon.set_mlp_precision("fp32")
    a, b = res["fp32"], res["bf16"]
    assert a[0] != b[0]
    assert abs(a[0] - b[0]) < 1e-3 * abs(a[0])
    assert not torch.isnan(b[1]).any()
    assert float((a[1] * b[1]).sum() / (a[1].norm() * b[1].norm())) > 0.9


def test_bf16_storage_of_attention_tensor_close_to_fp32_storage(monkeypatch):
    """bf16 mode with the [E, C] pre-activation of the second attention conv STORED as bf16
    (OPT.Z16_STORAGE: written by gridgcn_linear_fwd_direct_ld zfmt 1, read by
    gridgcn_pairmax_fwd_src_z and the fused attention backward) against the same mode with fp32
    storage.  Stated tolerance of the variant: aggregate within 1e-2 * max|y| (one bf16 rounding of a
    pre-activation whose BatchNorm+ReLU+product follow), every gradient within 5e-2 in relative L2
    norm (a few arg-max flips re-route single entries)."""
    import copy
    from grid_gcn_amd import ops
    from grid_gcn_amd.train import common as tcommon
    from grid_gcn_amd.train.options import OPT
    from grid_gcn_amd.gridconv import SubGUpdate
    torch.manual_seed(21)
    gen = torch.Generator().manual_seed(5)
    B, Nsrc, O, P, cin = 2, 300, 4000, 5, 128
    ref = SubGUpdate(cin, [128], localfdim=3).to(DEV).train()
    new = copy.deepcopy(ref)
    src1 = (torch.rand(B, Nsrc, 4 + cin, generator=gen) * 2 - 1).to(DEV).requires_grad_(True)
    src2 = src1.detach().clone().requires_grad_(True)
    nebidx = torch.randint(0, Nsrc, (B, O, P), generator=gen, dtype=torch.int32).to(DEV)
    cent = (torch.rand(B, O, 4, generator=gen) * 2 - 1).to(DEV)
    g = torch.randn(B, O, 128, generator=gen).to(DEV)
    outs = []
    monkeypatch.setattr(OPT, "NOZ_IN_BF16", False)        # (this shape would take the Z2-free pair: next test)
    try:
        tcommon.set_mlp_precision("bf16")
        for net, src, z16 in ((ref, src1, False), (new, src2, True)):
            monkeypatch.setattr(OPT, "Z16_STORAGE", z16)
            y = net.forward_src(cent, src, nebidx, None)
            y.backward(g)
            outs.append((y.detach(), src.grad.clone(), [p.grad.clone() for p in net.parameters()]))
    finally:
        tcommon.set_mlp_precision("fp32")
    (y0, s0, p0), (y1, s1, p1) = outs
    assert float((y0 - y1).abs().max()) > 0.0                       # the storage really changed
    assert float((y0 - y1).abs().max()) <= 1e-2 * float(y0.abs().max())
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))   # noqa: E731
    assert rel(s1[..., 4:], s0[..., 4:]) <= 5e-2
    for a, b in zip(p1, p0):
        assert rel(a, b) <= 5e-2, (rel(a, b), a.shape)


@pytest.mark.parametrize("B,Nsrc,O,geo", [(2, 150, 33, True), (3, 150, 700, False), (1, 5000, 4099, True)])
def test_att_max_eval_tile_kernel_matches_the_general_kernel(B, Nsrc, O, geo):
    """gridgcn_att_max_eval at the up layers' shape runs the tile kernel of the training forward without its arg max and
    saved pre-activations (GRIDGCN_OPT_ATT_EVAL_TILE, csrc/gridgcn_attfwd.hip <false>); against the general evaluation
    kernel (csrc/gridgcn_atteval.hip, option 0) on the same inputs: 2e-6 of the largest entry (the attention value in
    another summation order, the BatchNorm maps as fused multiply-adds), and against float64 1e-5; an agg that is the
    left half of a wider buffer, untouched elsewhere."""
    from grid_gcn_amd import _lib
    from grid_gcn_amd.ops import _ptr, _stream
    lib = _lib.load()
    d = _att_fwd_inputs(B, Nsrc, O, B * 77 + O, geo)
    P, cin, C, ncent = d["P"], d["cin"], d["C"], d["ncent"]
    sca, sha = d["gamma"], d["beta"]
    lda = 256
    outs = []
    try:
        for tile in (1, 0):
            assert lib.gridgcn_set_option(_lib.OPT_ATT_EVAL_TILE, tile) == 0
            assert lib.gridgcn_get_option(_lib.OPT_ATT_EVAL_TILE) == tile
            wide = torch.full((ncent + 3, lda), 7.0, device=DEV)
            rc = lib.gridgcn_att_max_eval(_ptr(d["Z1"]), _ptr(d["s1"]), _ptr(d["h1"]), _ptr(d["W2"]), _ptr(d["b2"]),
                                          _ptr(sca), _ptr(sha), _ptr(d["Ysrc"]), _ptr(d["nebidx"]), _ptr(d["att16"]),
                                          _ptr(d["Wg"]) if geo else None, _ptr(d["b"]), _ptr(d["scp"]), _ptr(d["shp"]),
                                          B, Nsrc, O, P, C, _ptr(wide), lda, _stream(wide))
            assert rc == 0
            assert bool((wide[:, C:] == 7.0).all()) and bool((wide[ncent:] == 7.0).all())
            outs.append(wide[:ncent, :C].clone())
    finally:
        lib.gridgcn_set_option(_lib.OPT_ATT_EVAL_TILE, 1)
    a1 = torch.clamp_min(d["Z1"] * d["s1"] + d["h1"], 0)
    z2 = a1.double() @ d["W2"].double().t() + d["b2"].double()
    flat = (d["nebidx"].long() + (torch.arange(B, device=DEV) * Nsrc)[:, None, None]).clamp(0, B * Nsrc - 1)
    zp = d["Ysrc"].double()[flat.reshape(-1)]
    if geo:
        zp = zp + d["att16"][:, 1:4].double() @ d["Wg"].double()
    zp = zp + d["b"].double()
    ref = (torch.clamp_min(zp * d["scp"].double() + d["shp"].double(), 0)
           * torch.clamp_min(z2 * sca.double() + sha.double(), 0)).reshape(ncent, P, C).max(1).values
    top = float(ref.abs().max())
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * top
    assert float((outs[0].double() - ref).abs().max()) <= 1e-5 * top
    assert float((outs[1].double() - ref).abs().max()) <= 1e-5 * top


def test_bf16_mode_takes_the_z2_free_attention_pair_where_it_applies(monkeypatch):
    """bf16 mode, up-layer shape (P = 5, attention 10 -> 32 -> 128): OPT.NOZ_IN_BF16 runs the second attention conv's
    forward and backward on the fp32 Z2-free kernels (csrc/gridgcn_attfwd.hip, gridgcn_attbwd_nz.hip) instead of the
    bf16-stored tensor: one conv of the block in exact fp32 instead of bf16 operands.  So, against the fp32 mode on the
    same inputs, the variant must be no farther away than plain bf16 mode is (output: 1.2 x in max norm; gradients:
    1.5 x in relative L2 norm -- single arg-max flips move entries either way), and no [E, 128] tensor is saved."""
    import copy
    from grid_gcn_amd.train import common as tcommon
    from grid_gcn_amd.train.options import OPT
    from grid_gcn_amd.gridconv import SubGUpdate
    torch.manual_seed(22)
    gen = torch.Generator().manual_seed(6)
    B, Nsrc, O, P, cin = 2, 300, 4000, 5, 128
    net0 = SubGUpdate(cin, [128], localfdim=3).to(DEV).train()
    src0 = (torch.rand(B, Nsrc, 4 + cin, generator=gen) * 2 - 1).to(DEV)
    nebidx = torch.randint(0, Nsrc, (B, O, P), generator=gen, dtype=torch.int32).to(DEV)
    cent = (torch.rand(B, O, 4, generator=gen) * 2 - 1).to(DEV)
    g = torch.randn(B, O, 128, generator=gen).to(DEV)
    outs = []
    monkeypatch.setattr(OPT, "Z16_STORAGE", False)
    try:
        for mode, nzb in (("fp32", True), ("bf16", False), ("bf16", True)):
            tcommon.set_mlp_precision(mode)
            monkeypatch.setattr(OPT, "NOZ_IN_BF16", nzb)
            net, src = copy.deepcopy(net0), src0.clone().requires_grad_(True)
            saved = []
            with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(tuple(t.shape)), t)[1], lambda t: t):
                y = net.forward_src(cent, src, nebidx, None)
            assert ((B * O * P, 128) in saved) == (mode == "bf16" and not nzb)
            y.backward(g)
            outs.append((y.detach(), src.grad.clone(), [p.grad.clone() for p in net.parameters()]))
    finally:
        tcommon.set_mlp_precision("fp32")
    (yf, sf, pf), (yb, sb, pb), (yn, sn, pn) = outs
    assert float((yb - yn).abs().max()) > 0.0
    assert float((yn - yf).abs().max()) <= 1.2 * float((yb - yf).abs().max())
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))   # noqa: E731
    assert rel(sn[..., 4:], sf[..., 4:]) <= 1.5 * rel(sb[..., 4:], sf[..., 4:]) + 1e-3
    for a, b, f in zip(pn, pb, pf):
        if float(f.norm()) > 0:
            assert rel(a, f) <= 1.5 * rel(b, f) + 1e-3, (rel(a, f), rel(b, f), a.shape)


@pytest.mark.parametrize("B,N,O,P,C,outlier", [(2, 50, 300, 5, 40, False), (2, 50, 300, 5, 40, True),
                                               (3, 257, 2000, 7, 128, True), (1, 1000, 5000, 5, 16, False)])
def test_edge_lin0_backward_sparse_fixed_point_matches_float64(B, N, O, P, C, outlier):
    """gridgcn_edge_lin0_backward_sparse (64-bit fixed-point LDS sums, csrc/gridgcn_edgelin.hip) against
    a float64 restatement of its formula: dYsrc, Gsum, wgs, gg.  `outlier`: one gradient entry 1e9 times
    the others (beyond the fixed-point headroom -> the fp32 side path) and a few indices that leave the
    cloud (the clip of mx.sym.take, utils/ops.py:78-93 -> the same side path)."""
    import ctypes
    from grid_gcn_amd import _lib
    lib = _lib.load()
    torch.manual_seed(B * 1000 + N + C)
    E = B * O * P
    nebidx = torch.randint(0, N, (B, O, P), device=DEV, dtype=torch.int32)
    if outlier:
        nebidx[0, 5, :] = -3          # clipped to row 0 of cloud 0
        nebidx[B - 1, 7, :] = N + 9   # clipped to the last row
        if B > 1:
            nebidx[1, 9, :] = -2      # lands in cloud 0 (rows N-2 ... of the cloud in front)
    att16 = torch.randn(E, 16, device=DEV)
    amax = torch.randint(0, P, (B * O, C), device=DEV, dtype=torch.uint8)
    gval = torch.randn(B * O, C, device=DEV)
    zsel = torch.randn(B * O, C, device=DEV)
    if outlier:
        gval[O - 1, C // 2] = 1e9
        zsel[O - 1, C // 2] = 10.0
    Ysrc = torch.randn(B * N, C, device=DEV)
    wgb = torch.randn(4, C, device=DEV)
    sc, sh, mean = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1, torch.randn(C, device=DEV)
    rstd, m1, m2 = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1, torch.randn(C, device=DEV) * 0.1
    if outlier:
        sc[C // 2], sh[C // 2] = 1.0, 0.0
    dYsrc = torch.empty(B * N, C, device=DEV)
    Gsum = torch.empty(B * N, 4, device=DEV)
    acc64 = torch.zeros(3 * C + 12, dtype=torch.float64, device=DEV)
    nb = ctypes.c_size_t(0)
    lib.gridgcn_edge_lin0_backward_sparse_workspace_bytes(B, N, C, ctypes.byref(nb))
    ws = torch.empty(nb.value, dtype=torch.uint8, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    rc = lib.gridgcn_edge_lin0_backward_sparse(
        p(nebidx), p(att16), p(amax), p(gval), p(zsel), p(Ysrc), p(wgb), p(wgb[3]), p(sc), p(sh), p(mean),
        p(rstd), p(m1), p(m2), B, N, O, P, C, p(dYsrc), p(Gsum), p(acc64), p(acc64[3 * C:]), p(ws), nb.value,
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    # ---- float64 restatement
    d = torch.float64
    rows = B * N
    s = torch.where(zsel * sc + sh > 0, sc * gval, torch.zeros_like(gval)).to(d)          # fp32 values, as the kernel
    cen = torch.arange(B * O, device=DEV)
    estar = cen[:, None] * P + amax.long()                                                    # [B*O, C]
    bof = (torch.arange(B, device=DEV) * N).repeat_interleave(O * P)
    flat = (nebidx.reshape(-1).long() + bof).clamp(0, rows - 1)                               # [E]
    dest = flat[estar]                                                                         # [B*O, C]
    part = torch.zeros(rows, C, dtype=d, device=DEV)
    part.scatter_add_(0, dest, s)
    geo = att16[:, 1:4].to(d)
    G = torch.zeros(rows, 3, dtype=d, device=DEV).index_add_(0, flat, geo)
    cnt = torch.zeros(rows, dtype=d, device=DEV).index_add_(0, flat, torch.ones(E, dtype=d, device=DEV))
    scd, bz = sc.to(d), -(sc.to(d) * rstd.to(d)) * m2.to(d)
    cz = -(scd * m1.to(d))
    lin = cnt[:, None] * (Ysrc.to(d) + wgb[3].to(d)) + G @ wgb[:3].to(d)
    ref = part + bz * lin + cnt[:, None] * (cz - mean.to(d) * bz)
    tol = 2e-6 * ref.abs().max(dim=0).values + 1e-5
    assert bool(((dYsrc.to(d) - ref).abs() <= tol).all()), float(((dYsrc.to(d) - ref).abs() / tol).max())
    assert float((Gsum[:, :3].to(d) - G).abs().max()) <= 1e-5 * max(1.0, float(G.abs().max()))
    assert torch.equal(Gsum[:, 3].to(d), cnt)
    gstar = geo[estar]                                                                         # [B*O, C, 3]
    wgs_ref = (gstar * s[..., None]).sum(0).t()                                                # [3, C]
    assert float((acc64[:3 * C].view(3, C) - wgs_ref).abs().max()) <= 1e-5 * float(wgs_ref.abs().max())
    gg_ref = torch.cat([(geo[:, :, None] * geo[:, None, :]).sum(0).reshape(9), geo.sum(0)])
    assert float((acc64[3 * C:] - gg_ref).abs().max()) <= 2e-5 * float(gg_ref.abs().max())


@pytest.mark.parametrize("ncent,P,C", [(8192, 128, 64), (2048, 32, 128), (192, 32, 256), (777, 17, 12),
                                       (40000, 16, 64), (5, 8, 4), (300000, 8, 32)])
def test_pairmax_fwd_first_argmax_exact(ncent, P, C):
    """gridgcn_pairmax_fwd: products, first arg max and the selected pre-activations EXACTLY as a
    sequential scan gives them, for shapes on the neighbour-split kernel (few (centre, quad) threads: the
    P neighbours dealt to 2-8 lanes, partial maxima merged) and on the one-thread-per-quad kernel (the
    last shape).  Many exact ties: ReLU zeros, and values quantised to a few levels."""
    import ctypes
    from grid_gcn_amd import _lib
    lib = _lib.load()
    torch.manual_seed(ncent + P + C)
    Zp = (torch.randn(ncent * P, C, device=DEV) * 2).round() / 2          # multiples of 0.5: ties
    Za = (torch.randn(ncent * P, C, device=DEV) * 2).round() / 2
    sp, hp = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
    sa, ha = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
    agg = torch.empty(ncent, C, device=DEV)
    amax = torch.empty(ncent, C, device=DEV, dtype=torch.uint8)
    zsel = torch.empty(2, ncent * C, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    rc = lib.gridgcn_pairmax_fwd(p(Zp), p(Za), p(sp), p(hp), p(sa), p(ha), ncent, P, C, p(agg), C, p(amax),
                                 p(zsel), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    y1 = torch.relu(Zp * sp + hp).view(ncent, P, C)           # the kernel's fp32 operations, one by one
    y2 = torch.relu(Za * sa + ha).view(ncent, P, C)
    v = y1 * y2
    best = v.max(dim=1).values
    first = (v == best[:, None, :]).to(torch.uint8).argmax(dim=1)            # first maximal neighbour
    assert torch.equal(agg, best)
    assert torch.equal(amax.long(), first)
    idx = first[:, None, :]
    assert torch.equal(zsel[0].view(ncent, C), Zp.view(ncent, P, C).gather(1, idx)[:, 0])
    assert torch.equal(zsel[1].view(ncent, C), Za.view(ncent, P, C).gather(1, idx)[:, 0])
