"""CPU tier: the PRODUCT's index kernels (grid_gcn_amd/csrc/gridgcn_index*.hip, _query*.hip, _knn.hip, _ballgrid.hip,
_cas.hip, _fastrand.hip) behind their C-ABI entries (gridgcn_capi.hip), compiled for the host under the wave64 SIMT
emulator of tests/simt/ and run on numpy arrays, against the C oracle and the committed golden fixtures -- bit for bit, every output, every golden
case (the one 4.5 M-voxel case of the legacy build is left to the GPU tier: its kernels launch a work-item per voxel).

This is not the parity proof (that is the `-m gpu` tier: the same comparison on the real machine); it is what
can be said WITHOUT a GPU: the kernels' arithmetic and data flow, executed as written, reproduce the oracle.  Two
counters of the emulator are asserted to stay 0: shuffles that read a lane outside the set of lanes executing the
operation, and cross-lane operations reached in divergent control flow (where the emulator would have to guess the
hardware's reconvergence order)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import oracle as orc  # noqa: E402
from golden import make_golden  # noqa: E402
from test_oracle import check_against_golden  # noqa: E402
from grid_gcn_amd import _lib, synth  # noqa: E402
from simt import sim  # noqa: E402

ALL = [c for c in make_golden.all_cases() if c[0] != "gridify_legacy_4m_voxels"]


def run_sim(name, args, kw):
    if name.startswith("occaware"):
        return sim.Gridify_occaware(*args, **kw)
    if name.startswith("fastrand"):
        return sim.Gridify_fast_rand(*args, **kw)
    if name.startswith("gridify_knn"):
        return sim.GridifyKNN(*args, **kw)
    if name.startswith("gridify_up"):
        return sim.GridifyUp(*args, **kw)
    if name.startswith("gridify"):
        return sim.Gridify(*args, **kw)
    if name.startswith("ball_knn"):
        return (sim.BallKNN(*args, **kw),)
    if name.startswith("knn"):
        return (sim.KNN(*args, k=kw["k"]),)
    raise KeyError(name)


def _same(got, want, what):
    for j, (g, w) in enumerate(zip(got, want)):
        if not np.array_equal(g, w, equal_nan=True):
            bad = np.argwhere(g != w)
            raise AssertionError("%s output %d: %d mismatches, first at %s: got %s want %s" % (
                what, j, len(bad), bad[0], g[tuple(bad[0])], w[tuple(bad[0])]))


@pytest.fixture(autouse=True)
def _no_guesses():
    c0 = sim.counters()
    yield
    c1 = sim.counters()
    assert c1[2] == c0[2], "a shuffle read a lane outside its group"
    assert c1[3] == c0[3], "a cross-lane operation was reached in divergent control flow"


@pytest.mark.parametrize("name,build,run", ALL, ids=[c[0] for c in ALL])
def test_emulated_kernels_match_oracle_and_golden(name, build, run):
    args, kw = build()
    want = run(args, kw)
    got = run_sim(name, args, kw)
    _same(got, want, name)
    check_against_golden(name, got)


# Arrival order: the operators' outputs are DEFINED to be independent of it (the reservoir keeps the largest item
# number, ranks come from ballots and prefix sums) -- so the emulator runs the launch in other orders:
# every workgroup, wave and lane descending (7); workgroups in a pseudo-random permutation, waves descending (10).  The
# last-arriver hand-offs of the build then fall to other workgroups; the bytes must not move.
ORDERED = [c for c in ALL if not c[0].startswith(("gridify_seg80k", "gridify_synth200k", "gridify_cfg"))]


@pytest.mark.parametrize("order", [7, 10])
@pytest.mark.parametrize("name,build,run", ORDERED, ids=[c[0] for c in ORDERED])
def test_emulated_kernels_do_not_depend_on_arrival_order(name, build, run, order):
    if order == 10 and os.environ.get("GG_SIMT_FULL") != "1" and [c[0] for c in ORDERED].index(name) % 3:
        pytest.skip("permuted workgroups: every third case unless GG_SIMT_FULL=1")
    args, kw = build()
    try:
        sim.set_order(order)
        got = run_sim(name, args, kw)
    finally:
        sim.set_order(0)
    check_against_golden(name, got)


SMALL = [c for c in ALL if c[0].startswith(("gridify_mn40", "gridify_scan8k_L1", "gridify_oob", "gridify_weights",
                                            "gridify_ragged", "gridify_up_overflow", "fastrand_scan8k_overfull"))]


@pytest.mark.parametrize("name,build,run", SMALL, ids=[c[0] for c in SMALL])
def test_emulated_split_build_equals_small_build(name, build, run):
    """clouds of <= 4096 points through the three-launch two-level split (GRIDGCN_OPT_INDEX_SMALL = 0) as well: the
    same bytes as the one-launch build, which the test above has compared with the oracle"""
    args, kw = build()
    want = run(args, kw)
    try:
        sim.set_option(_lib.OPT_INDEX_SMALL, 0)
        got = run_sim(name, args, kw)
    finally:
        sim.set_option(_lib.OPT_INDEX_SMALL, 1)
    _same(got, want, name + " (split build)")


@pytest.mark.parametrize("N,G,P,O,k", [(1, 5, 4, 3, 3), (63, 1, 8, 1, 1), (64, 3, 2, 70, 3), (1000, 9, 4, 50, 3),
                                       (1025, 17, 3, 300, 5), (2048, 40, 64, 256, 7), (3001, 64, 1, 512, 3),
                                       (4096, 2, 128, 8, 3), (5000, 24, 5, 700, 3)])
def test_emulated_build_edge_shapes(N, G, P, O, k):
    """the shapes of test_gpu_parity.py::test_small_build_shapes (1 / 2 / 4 items per thread, one voxel, ragged
    counts, every voxel over-full, more slots than voxels) plus one cloud beyond the one-launch build"""
    rng = np.random.default_rng(N * 131 + G)
    B = 3
    xyz = rng.uniform(-1.05, 1.05, (B, N, 3)).astype(np.float32)
    w = np.ones((B, N, 1), np.float32)
    w[1] = rng.integers(1, 5, (N, 1)).astype(np.float32)
    data = np.concatenate([xyz, w], 2)
    npn = np.array([[N], [max(1, N - 7)], [max(1, N // 2)]], np.int32)
    kw = dict(max_p_grid=P, max_o_grid=O, kernel_size=k, stride=1, loc=1, coord_shift=[1.0] * 3,
              voxel_size=[2.0 / G] * 3, grid_size=[G] * 3, seed=N + 17)
    _same(sim.Gridify(data, npn, **kw), orc.gridify(data, npn, **kw), "gridify")
    ukw = dict(kw, max_o_grid=333)
    ukw.pop("stride"); ukw.pop("loc")
    up = rng.uniform(-1.0, 1.0, (B, 333, 4)).astype(np.float32)
    upn = np.array([[333], [300], [1]], np.int32)
    _same(sim.GridifyUp(data, up, npn, upn, **ukw), orc.gridify_up(data, up, npn, upn, **ukw), "gridify_up")


def test_emulated_legacy_build_matches_oracle():
    """the first-generation build (gridgcn_index_legacy.hip: the fallback for grids beyond 1024 slabs x 4096 voxels; its one
    golden case has 4.5 M voxels) at a size the emulator runs: a 64^3 grid whose slab plan is pushed out of range by
    GRIDGCN_OPT_INDEX_SLAB_SHIFT = -4 -- Gridify, GridifyKNN and GridifyUp through it, over-full voxels, ragged counts"""
    rng = np.random.default_rng(11)
    B, N, G = 2, 3000, 64
    xyz = rng.uniform(-0.33, 0.33, (B, N, 3)).astype(np.float32)
    xyz[1, :40] = 0.7                                        # one voxel far beyond P
    w = rng.integers(1, 4, (B, N, 1)).astype(np.float32)
    data = np.concatenate([xyz, w], 2)
    npn = np.array([[N], [N - 77]], np.int32)
    kw = dict(max_p_grid=2, max_o_grid=300, kernel_size=3, stride=1, loc=1, coord_shift=[1.0] * 3,
              voxel_size=[2.0 / G] * 3, grid_size=[G] * 3, seed=5)
    up = rng.uniform(-0.4, 0.4, (B, 200, 4)).astype(np.float32)
    upn = np.array([[200], [123]], np.int32)
    ukw = dict(kw, max_o_grid=200)
    ukw.pop("stride"); ukw.pop("loc")
    k0 = set(sim.emu.kernel_coverage())
    try:
        sim.set_option(_lib.OPT_INDEX_SMALL, 0)
        sim.set_option(_lib.OPT_INDEX_SLAB_SHIFT, -4)
        got = sim.Gridify(data, npn, **kw)
        got_knn = sim.GridifyKNN(data, npn, **kw)
        got_up = sim.GridifyUp(data, up, npn, upn, **ukw)
    finally:
        sim.set_option(_lib.OPT_INDEX_SMALL, 1)
        sim.set_option(_lib.OPT_INDEX_SLAB_SHIFT, 0)
    ran = set(sim.emu.kernel_coverage()) - k0 | k0
    assert {"gg_k_voxelize", "gg_k_slab_count", "gg_k_scatter", "gg_k_centres", "gg_k_legacy_pack_vtab"} <= ran, ran
    _same(got, orc.gridify(data, npn, **kw), "gridify (legacy build)")
    _same(got_knn, orc.gridify_knn(data, npn, **kw), "gridify_knn (legacy build)")
    _same(got_up, orc.gridify_up(data, up, npn, upn, **ukw), "gridify_up (legacy build)")


def test_emulated_ball_knn_grid_thread_per_query_form():
    """more than 32768 queries: the cell-grid BallKNN with a THREAD per query (gg_k_ball_grid_query; below that the
    eight-lanes-per-query form runs) -- both k ranges, against the oracle"""
    rng = np.random.default_rng(3)
    B, n, m = 2, 16500, 300
    un = rng.uniform(-1.0, 1.0, (B, n, 3)).astype(np.float32)
    kn = rng.uniform(-1.0, 1.0, (B, m, 3)).astype(np.float32)
    dn = np.array([[m], [m - 31]], np.int32)
    upn = np.array([[n], [n - 1000]], np.int32)
    for k, radius in ((3, 0.2), (5, 0.35)):
        want = orc.ball_knn(un, kn, dn, upn, k=k, radius=radius)
        got = sim.BallKNN(un, kn, dn, upn, k=k, radius=radius, grid=True)
        for b in range(B):
            assert np.array_equal(got[b, :upn[b, 0]], want[b, :upn[b, 0]])
    assert any(k.startswith("gg_k_ball_grid_query<") for k in sim.emu.kernel_coverage())


@pytest.mark.parametrize("n,m,k,radius,kind", [(700, 128, 5, 0.1275, "ball"), (600, 100, 3, 0.05, "planes"),
                                               (500, 128, 5, 5.0, "ball"), (500, 64, 4, 0.0, "ball"),
                                               (900, 300, 6, 0.02, "lattice"), (400, 200, 5, 0.3, "special")])
def test_emulated_ball_knn_grid_equals_scan_and_oracle(n, m, k, radius, kind):
    """BallKNN through the cell grid (gridgcn_ball_knn_grid) and through the all-pairs scan, both emulated, against
    the oracle: ties (lattice), queries outside the known points' box, partial counts, non-finite coordinates"""
    rng = np.random.default_rng(n + m + k)
    B = 2
    if kind == "lattice":
        un = rng.integers(-8, 9, (B, n, 3)).astype(np.float32) * 0.01
        kn = rng.integers(-8, 9, (B, m, 3)).astype(np.float32) * 0.01
    else:
        d1, _ = synth.make_batch(B, n, "planes" if kind == "planes" else "ball", first_id=7)
        d2, _ = synth.make_batch(B, m, "planes" if kind == "planes" else "ball", first_id=70)
        un, kn = d1[..., :3].copy() * 1.3, d2[..., :3].copy()
    if kind == "special":
        kn[0, 5] = np.nan
        kn[1, 7, 1] = np.inf
        un[0, 3, 0] = np.nan
        un[1, 4] = np.inf
    dn = np.array([[m], [m - 17]], np.int32)
    upn = np.array([[n], [n - 9]], np.int32)
    want = orc.ball_knn(un, kn, dn, upn, k=k, radius=radius)
    scan = sim.BallKNN(un, kn, dn, upn, k=k, radius=radius)
    grid = sim.BallKNN(un, kn, dn, upn, k=k, radius=radius, grid=True)
    for b in range(B):          # rows >= upnum are left untouched by the operator
        assert np.array_equal(scan[b, :upn[b, 0]], want[b, :upn[b, 0]])
        assert np.array_equal(grid[b, :upn[b, 0]], want[b, :upn[b, 0]])


@pytest.mark.parametrize("order", [0, pytest.param(10, marks=pytest.mark.skipif(
    os.environ.get("GG_SIMT_FULL") != "1", reason="permuted workgroups: with GG_SIMT_FULL=1"))])
def test_emulated_cfg2_batch32_gridify_chain(order):
    """BASELINE configs[1] at its real batch -- 32 clouds x 1024 points, ragged counts inside the batch -- through the
    three Gridify layers of the classifier, chained as the model chains them: the index half of the GPU tier's
    test_cls_cfg2_batch32_gridify_bit_exact_and_eval_logits (written in round 6 with the GPU pool closed: this is its
    first execution).  One workgroup per cloud in gg_k_small_build, cloud b on XCD b mod 8 -- B = 32 is four clouds per
    XCD, which no golden case has.  Layer 0 (k = 7: 343 voxels per centre, a minute under emulation) with
    GG_SIMT_FULL=1 only; the default tier runs layers 1 and 2 on the oracle's layer-0 centres."""
    full = os.environ.get("GG_SIMT_FULL") == "1"
    cfg = synth.CLS_MODELNET40
    data, npn = synth.make_batch(32, 1024, "ball")
    npn = npn.copy()
    npn[5, 0], npn[17, 0], npn[31, 0] = 1000, 513, 1
    d, n = data, npn
    try:
        sim.set_order(order)
        for l in range(3):
            kw = synth.gridify_kwargs(cfg, l, seed=3 + l)
            want = orc.gridify(d, n, **kw)
            if l > 0 or full:
                got = sim.Gridify(d, n, **kw)
                _same(got, want, "cfg2 b32 layer %d" % l)
            d, n = want[2], want[4]
    finally:
        sim.set_order(0)
