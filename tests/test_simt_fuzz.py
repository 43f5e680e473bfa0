"""CPU tier: the seeded random sweep of tests/test_gpu_fuzz.py (its own case generator: random grids, kernel sizes,
P / O, ragged counts, duplicate and out-of-grid points, integer weights) through the host-side emulation of the
index kernels -- and, instance by instance, through ANOTHER ARRIVAL ORDER of workgroups, waves and lanes
(tests/simt/simt_hip.h: simt_order), which the GPU tier cannot choose.  Every output bit for bit against the oracle.
GG_SIMT_FUZZ_N instances (4 by default: half a minute; GG_SIMT_FUZZ_N=200 -- 300 test instances -- ran clean in round 6: profiles/r6_simt_fuzz.txt)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import oracle as orc  # noqa: E402
from simt import sim  # noqa: E402
from test_gpu_fuzz import rand_case  # noqa: E402  (the generator only: nothing of that module's GPU tests runs here)

NFUZZ = int(os.environ.get("GG_SIMT_FUZZ_N", "4"))
ORDERS = [0, 7, 10, 5, 13, 2]       # bit mask: 1 / 2 / 4 workgroups / waves / lanes descending, 8 workgroups permuted


def same(want, got, what):
    got = got if isinstance(got, (tuple, list)) else (got,)
    for j, (w, g) in enumerate(zip(want, got)):
        g = np.asarray(g)
        assert w.shape == g.shape and w.tobytes() == g.tobytes(), \
            (what, j, int((w != g).sum()) if w.shape == g.shape else (w.shape, g.shape))


@pytest.fixture(autouse=True)
def _no_guesses():
    c0 = sim.counters()
    yield
    sim.set_order(0)
    c1 = sim.counters()
    assert c1[2] == c0[2], "a shuffle read a lane outside its group"
    assert c1[3] == c0[3], "a cross-lane operation was reached in divergent control flow"


@pytest.mark.parametrize("seed", range(NFUZZ))
def test_emulated_gridify_family_random(seed):
    rng = np.random.default_rng(1000 + seed)
    data, npn, kw = rand_case(rng)
    if kw["loc"] == 0:
        kw["loc"] = 1 if seed % 2 else 0
    sim.set_order(ORDERS[seed % len(ORDERS)])
    same(orc.gridify(data, npn, **kw), sim.Gridify(data, npn, **kw), ("gridify", seed, kw))
    same(orc.gridify_knn(data, npn, **kw), sim.GridifyKNN(data, npn, **kw), ("gridify_knn", seed, kw))
    same(orc.gridify_fast_rand(data, npn, **kw), sim.Gridify_fast_rand(data, npn, **kw), ("fast_rand", seed, kw))
    beta = float(rng.choice([0.0, 0.5, 1.0, 4.0]))
    same(orc.gridify_occaware(data, npn, beta=beta, **kw), sim.Gridify_occaware(data, npn, beta=beta, **kw),
         ("occaware", seed, kw, beta))


@pytest.mark.parametrize("seed", range(max(NFUZZ // 2, 1)))
def test_emulated_gridify_up_and_knn_random(seed):
    rng = np.random.default_rng(2000 + seed)
    data, npn, kw = rand_case(rng)
    B, N = data.shape[:2]
    M = int(rng.choice([16, 100, 777, 2048]))
    up = np.concatenate([rng.uniform(-1.1, 1.1, (B, M, 3)), np.ones((B, M, 1))], 2).astype(np.float32)
    upn = rng.integers(0, M + 1, (B, 1)).astype(np.int32)
    ku = dict(max_p_grid=int(rng.choice([1, 3, 5, 8, 16])), max_o_grid=M,
              kernel_size=int(rng.choice([1, 3, 5])), coord_shift=kw["coord_shift"],
              voxel_size=kw["voxel_size"], grid_size=kw["grid_size"], seed=kw["seed"])
    sim.set_order(ORDERS[(seed + 1) % len(ORDERS)])
    same(orc.gridify_up(data, up, npn, upn, **ku), sim.GridifyUp(data, up, npn, upn, **ku), ("gridify_up", seed, ku))
    k = int(rng.choice([1, 3, 5, 6]))
    r = float(rng.choice([0.05, 0.2, 0.5]))
    dn = np.maximum(npn, 1).astype(np.int32)
    un, kn = up[..., :3].copy(), data[..., :3].copy()
    want = orc.ball_knn(un, kn, dn, upn, k=k, radius=r)
    for grid in (False, True):
        got = sim.BallKNN(un, kn, dn, upn, k=k, radius=r, grid=grid)
        for b in range(B):                      # rows >= upnum are left untouched by the operator
            assert np.array_equal(got[b, :upn[b, 0]], want[b, :upn[b, 0]]), ("ball_knn", seed, k, r, grid)
    want = orc.knn(un, kn, dn, upn, k=k)
    got = sim.KNN(un, kn, dn, upn, k=k)
    for b in range(B):
        assert np.array_equal(got[b, :upn[b, 0]], want[b, :upn[b, 0]]), ("knn", seed, k)
