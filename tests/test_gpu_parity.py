"""GPU parity tests: HIP kernels (through the C ABI, grid_gcn_amd.ops) vs the CPU oracle and the
committed golden fixtures.  Bar: BIT-EXACT for every output (indices, masks, counts and the fp32
centres -- cent feeds the next layer's floor(), so 1 ulp is not good enough)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from oracle import oracle as orc  # noqa: E402
import cases  # noqa: E402
from golden import make_golden  # noqa: E402
from test_oracle import check_against_golden  # noqa: E402
from grid_gcn_amd import ops, synth  # noqa: E402

DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def NP(ts):
    return tuple(t.cpu().numpy() for t in ts)


def run_hip(name, args, kw):
    targs = [T(a) for a in args]
    if name.startswith("occaware"):
        return NP(ops.Gridify_occaware(*targs, **kw))
    if name.startswith("fastrand"):
        return NP(ops.Gridify_fast_rand(*targs, **kw))
    if name.startswith("gridify_knn"):
        return NP(ops.GridifyKNN(*targs, **kw))
    if name.startswith("gridify_up"):
        return NP(ops.GridifyUp(*targs, **kw))
    if name.startswith("gridify"):
        return NP(ops.Gridify(*targs, **kw))
    if name.startswith("ball_knn"):
        return NP((ops.BallKNN(*targs, **kw),))
    if name.startswith("knn"):
        return NP((ops.KNN(*targs, k=kw["k"]),))
    raise KeyError(name)


ALL = make_golden.all_cases()


def test_extension_loaded():
    """The HIP library is the thing under test -- fail loudly if it is not what ran."""
    from grid_gcn_amd import _lib
    lib = _lib.load()
    assert lib.gridgcn_abi_version() >= 1
    maps = open("/proc/self/maps").read()
    assert "libgridgcn_hip.so" in maps


@pytest.mark.parametrize("name,build,run", ALL, ids=[c[0] for c in ALL])
def test_hip_matches_oracle_and_golden(name, build, run):
    args, kw = build()
    want = run(args, kw)
    got = run_hip(name, args, kw)
    for j, (g, w) in enumerate(zip(got, want)):
        if not np.array_equal(g, w, equal_nan=True):
            bad = np.argwhere(g != w)
            raise AssertionError("%s output %d: %d mismatches, first at %s: got %s want %s" % (
                name, j, len(bad), bad[0], g[tuple(bad[0])], w[tuple(bad[0])]))
    check_against_golden(name, got)


@pytest.mark.parametrize("name", ["gridify_scan8k_L0", "gridify_seeded_1", "gridify_up_overflow"])
def test_hip_is_deterministic(name):
    """The reference is non-deterministic (atomics arrival order); this build is not."""
    build = {c[0]: c[1] for c in ALL}[name]
    args, kw = build()
    a = run_hip(name, args, kw)
    for _ in range(3):
        b = run_hip(name, args, kw)
        for x, y in zip(a, b):
            assert x.tobytes() == y.tobytes()


def test_gridify_full_size_properties():
    """BASELINE configs[3] size (B=8, N=81920, P=128): too slow for a python oracle loop but fine
    for the C oracle; also check the size-independent invariants."""
    cfg = synth.SEG_SCANNET_81920
    data, npn = synth.make_batch(8, cfg["num_points"], "planes")
    d, n = data, npn
    td, tn = T(data), T(npn)
    for l in range(3):
        kw = synth.gridify_kwargs(cfg, l)
        want = orc.gridify(d, n, **kw)
        got = NP(ops.Gridify(td, tn, **kw))
        for g, w in zip(got, want):
            assert g.tobytes() == w.tobytes(), "layer %d" % l
        idx, msk, cent, cmsk, cn = got
        m = msk.sum(-1).astype(np.int64)
        assert np.all(cmsk.sum(-1) == cn[:, 0])
        assert np.all(m[cmsk > 0] >= 1)
        d, n = want[2], want[4]
        td, tn = T(d), T(n)


def test_gridify_synth200k():
    """BASELINE configs[4]: 200k points, 64^3 grid, O=16384 (2 clouds to bound oracle time)."""
    cfg = synth.SYNTH_200K
    data, npn = synth.make_batch(2, cfg["num_points"], "planes", first_id=100)
    d, n = data, npn
    for l in range(4):
        kw = synth.gridify_kwargs(cfg, l)
        want = orc.gridify(d, n, **kw)
        got = NP(ops.Gridify(T(d), T(n), **kw))
        for j, (g, w) in enumerate(zip(got, want)):
            assert g.tobytes() == w.tobytes(), "layer %d out %d" % (l, j)
        d, n = want[2], want[4]


@pytest.mark.parametrize("n,m,k,radius,kind", [
    (4096, 512, 5, 0.1275, "ball"), (4096, 512, 3, 0.05, "planes"), (2000, 300, 6, 0.4, "ball"),
    (3000, 128, 5, 5.0, "ball"), (3000, 128, 4, 0.0, "ball"), (5000, 1024, 6, 0.02, "lattice"),
    (1500, 700, 5, 0.3, "special"),
    # B * n > 32768: one thread per query (the cases above: eight lanes per query, merged top-k lists)
    (12000, 256, 5, 0.1275, "ball"), (16384, 300, 3, 0.2, "lattice"), (11000, 200, 6, 0.3, "special")])
def test_ball_knn_grid_equals_scan(n, m, k, radius, kind):
    """BallKNN through the cell grid (gridgcn_ball_knn_grid) == the S0 oracle's all-pairs scan, bit
    for bit: ties (lattice: many equal distances), queries outside the known points' box, partial
    downnum / upnum, non-finite coordinates, radius 0 and radius >> extent."""
    rng = np.random.default_rng(n + m + k)
    B = 3
    if kind == "lattice":       # coordinates on a coarse lattice: lots of exactly equal distances
        un = rng.integers(-8, 9, (B, n, 3)).astype(np.float32) * 0.01
        kn = rng.integers(-8, 9, (B, m, 3)).astype(np.float32) * 0.01
    else:
        d1, _ = synth.make_batch(B, n, "planes" if kind == "planes" else "ball", first_id=7)
        d2, _ = synth.make_batch(B, m, "planes" if kind == "planes" else "ball", first_id=70)
        un, kn = d1[..., :3].copy() * 1.3, d2[..., :3].copy()      # some queries outside the box
    if kind == "special":
        kn[0, 5] = np.nan
        kn[1, 7, 1] = np.inf
        kn[2, 9] = -np.inf
        un[0, 3, 0] = np.nan
        un[1, 4] = np.inf
        un[2, 6] = 1e30
    dn = np.array([[m], [m - 17], [m // 2]], np.int32)
    upn = np.array([[n], [n - 100], [1]], np.int32)
    want = orc.ball_knn(un, kn, dn, upn, k=k, radius=radius)
    assert ops.BALL_GRID and n * m >= (1 << 16) and m >= 64          # the grid path is taken
    got = NP((ops.BallKNN(T(un), T(kn), T(dn), T(upn), k=k, radius=radius),))[0]
    assert np.array_equal(got, want)
    # and the library's own all-pairs kernel agrees as well
    import os
    ops.BALL_GRID = False
    try:
        scan = NP((ops.BallKNN(T(un), T(kn), T(dn), T(upn), k=k, radius=radius),))[0]
    finally:
        ops.BALL_GRID = True
    assert np.array_equal(scan, want)


def test_error_behaviour():
    data, npn = synth.make_batch(1, 256, "ball")
    kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 0)
    with pytest.raises(RuntimeError):
        ops.Gridify(T(data)[..., :3].contiguous(), T(npn), **kw)       # last dim must be 4
    with pytest.raises(RuntimeError):
        ops.Gridify(T(data).double(), T(npn), **kw)                    # dtype
    with pytest.raises(RuntimeError):
        ops.Gridify(T(data), T(npn).long(), **kw)                      # int32 counts
    bad = dict(kw, max_p_grid=200)
    with pytest.raises(RuntimeError):
        ops.Gridify(T(data), T(npn), **bad)
    with pytest.raises(RuntimeError):
        ops.BallKNN(T(data[..., :3]), T(data[..., :3]), T(npn), T(npn), k=7, radius=1.0)


def test_batch_take_matches_oracle_and_grad():
    rng = np.random.default_rng(0)
    # (20,6): global-atomic backward; (80,6) and (300,5): LDS-privatised backward (M >= 4N)
    for C, ishape in ((4, (20, 6)), (7, (20, 6)), (68, (20, 6)), (7, (80, 6)), (68, (80, 6)),
                      (132, (300, 5)), (260, (300, 5))):
        data = rng.standard_normal((3, 50, C)).astype(np.float32)
        index = rng.integers(-1, 51, (3,) + ishape).astype(np.int32)
        want = orc.batch_take(data, index)
        td = T(data).requires_grad_(True)
        out = ops.batch_take_g(td, T(index))
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
        g = rng.standard_normal(out.shape).astype(np.float32)
        out.backward(T(g))
        flat = np.clip(index + (np.arange(3) * 50)[:, None, None], 0, 149).reshape(-1)
        ref = np.zeros((150, C), np.float64)
        np.add.at(ref, flat, g.reshape(-1, C).astype(np.float64))
        np.testing.assert_allclose(td.grad.cpu().numpy().reshape(150, C), ref, rtol=1e-5, atol=1e-5)
        # neighbour indices ([-1, N-1], as the index ops produce): sorted segmented-sum backward
        index2 = rng.integers(-1, 50, (3,) + ishape).astype(np.int32)
        td2 = T(data).requires_grad_(True)
        out2 = ops.batch_take_g(td2, T(index2), neighbour_index=True)
        np.testing.assert_array_equal(out2.detach().cpu().numpy(), orc.batch_take(data, index2))
        out2.backward(T(g))
        flat2 = np.clip(index2 + (np.arange(3) * 50)[:, None, None], 0, 149).reshape(-1)
        ref2 = np.zeros((150, C), np.float64)
        np.add.at(ref2, flat2, g.reshape(-1, C).astype(np.float64))
        np.testing.assert_allclose(td2.grad.cpu().numpy().reshape(150, C), ref2, rtol=1e-5, atol=1e-5)


def test_torch_ops_namespace_runs_the_same_kernels():
    """torch.ops.gridgcn.* (grid_gcn_amd/torch_ops.py) == grid_gcn_amd.ops, bit for bit, and
    batch_take is differentiable through the dispatcher."""
    import grid_gcn_amd  # noqa: F401  (registers the ops)
    data, npn = synth.make_batch(2, 4096, "planes")
    d = torch.from_numpy(data).to(DEV)
    n = torch.from_numpy(npn).to(DEV)
    kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 0, seed=7)
    a = ops.Gridify(d, n, **kw)
    b = torch.ops.gridgcn.gridify(d, n, **kw)
    c = torch.ops.gridgcn.gridify_knn(d, n, **kw)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    for x, y in zip(ops.GridifyKNN(d, n, **kw), c):
        assert torch.equal(x, y)
    cent, cnum = a[2], a[4]
    i1 = ops.BallKNN(d[..., :3].contiguous(), cent[..., :3].contiguous(), cnum, n, k=5, radius=0.2)
    i2 = torch.ops.gridgcn.ball_knn(d[..., :3].contiguous(), cent[..., :3].contiguous(), cnum, n,
                                    k=5, radius=0.2)
    assert torch.equal(i1, i2)
    k1 = ops.KNN(d[..., :3].contiguous(), cent[..., :3].contiguous(), cnum, n, k=3)
    k2 = torch.ops.gridgcn.knn(d[..., :3].contiguous(), cent[..., :3].contiguous(), cnum, n, k=3)
    assert torch.equal(k1, k2)
    ukw = synth.gridify_up_kwargs(synth.SEG_SCANNET_8192, 2, seed=7)
    up = torch.zeros((2, ukw["max_o_grid"], 4), device=DEV)
    up[:, :4096] = d
    u1 = ops.GridifyUp(cent.contiguous(), up, cnum, n, **ukw)
    u2 = torch.ops.gridgcn.gridify_up(cent.contiguous(), up, cnum, n, **ukw)
    assert torch.equal(u1[0], u2[0]) and torch.equal(u1[1], u2[1])
    f1 = torch.randn(2, 4096, 12, device=DEV).requires_grad_(True)
    f2 = f1.detach().clone().requires_grad_(True)
    g1 = ops.batch_take_g(f1, a[0])
    g2 = torch.ops.gridgcn.batch_take(f2, a[0])
    assert torch.equal(g1, g2)
    w = torch.randn_like(g1)
    (g1 * w).sum().backward()
    (g2 * w).sum().backward()
    assert torch.allclose(f1.grad, f2.grad, rtol=1e-5, atol=1e-5)


SMALL = [c for c in ALL if c[0].startswith(("gridify", "occaware", "fastrand"))]


@pytest.mark.parametrize("name,build,run", SMALL, ids=[c[0] for c in SMALL])
def test_small_build_equals_split_build(name, build, run):
    """Clouds of <= 4096 points take the one-launch, all-LDS index build (gg_k_small_build, round 4); the
    three-launch two-level split stays the build of every larger cloud.  Same inputs through both
    (GRIDGCN_OPT_INDEX_SMALL 1 / 0): every output byte equal -- and the default path is what the oracle /
    golden test above has checked.  Cases with N > 4096 or max_o_grid > 4096 run the split build twice
    (nothing to compare, kept for the ids)."""
    from grid_gcn_amd import _lib
    lib = _lib.load()
    args, kw = build()
    assert lib.gridgcn_get_option(_lib.OPT_INDEX_SMALL) == 1
    a = run_hip(name, args, kw)
    try:
        _lib.check(lib.gridgcn_set_option(_lib.OPT_INDEX_SMALL, 0), "set_option")
        b = run_hip(name, args, kw)
    finally:
        _lib.check(lib.gridgcn_set_option(_lib.OPT_INDEX_SMALL, 1), "set_option")
    for j, (x, y) in enumerate(zip(a, b)):
        assert x.tobytes() == y.tobytes(), "%s output %d" % (name, j)


@pytest.mark.parametrize("N,G,P,O,k", [(1, 5, 4, 3, 3), (63, 1, 8, 1, 1), (64, 3, 2, 70, 3), (1000, 9, 4, 50, 3),
                                       (1025, 17, 3, 2000, 5), (2048, 40, 64, 1024, 7), (3001, 64, 1, 4096, 3),
                                       (4096, 2, 128, 8, 3), (4096, 200, 5, 4096, 3), (4096, 255, 2, 300, 1)])
def test_small_build_shapes(N, G, P, O, k):
    """The one-launch build at the edges of its domain (1 / 2 / 4 items per thread, 1 / 2 / 3 radix passes:
    G^3 up to 2^24 - 1, one voxel, ragged counts, every voxel over-full, more slots than voxels) against
    the oracle, bit for bit."""
    rng = np.random.default_rng(N * 131 + G)
    B = 3
    xyz = rng.uniform(-1.05, 1.05, (B, N, 3)).astype(np.float32)
    w = np.ones((B, N, 1), np.float32)
    w[1] = rng.integers(1, 5, (N, 1)).astype(np.float32)
    data = np.concatenate([xyz, w], 2)
    npn = np.array([[N], [max(1, N - 7)], [max(1, N // 2)]], np.int32)
    kw = dict(max_p_grid=P, max_o_grid=O, kernel_size=k, stride=1, loc=1, coord_shift=[1.0] * 3,
              voxel_size=[2.0 / G] * 3, grid_size=[G] * 3, seed=N + 17)
    want = orc.gridify(data, npn, **kw)
    got = NP(ops.Gridify(T(data), T(npn), **kw))
    for j, (g, x) in enumerate(zip(got, want)):
        assert g.tobytes() == x.tobytes(), "output %d" % j
    ukw = dict(kw, max_o_grid=777)
    ukw.pop("stride"); ukw.pop("loc")
    up = rng.uniform(-1.0, 1.0, (B, 777, 4)).astype(np.float32)
    upn = np.array([[777], [700], [1]], np.int32)
    want = orc.gridify_up(data, up, npn, upn, **ukw)
    got = NP(ops.GridifyUp(T(data), T(up), T(npn), T(upn), **ukw))
    for j, (g, x) in enumerate(zip(got, want)):
        assert g.tobytes() == x.tobytes(), "up output %d" % j
