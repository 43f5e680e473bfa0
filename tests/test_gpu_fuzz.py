"""Seeded random sweep of the index operators against the oracle: grid shapes, kernel sizes, P / O,
cloud sizes, ragged counts, out-of-grid and duplicate points, integer weights -- every output tensor
bit for bit.  (The fixed golden cases pin known corner cases; this looks for the unknown ones.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from grid_gcn_amd import ops  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DEV = "cuda:0"
import os  # noqa: E402
NFUZZ = int(os.environ.get("GG_FUZZ_N", "24"))     # GG_FUZZ_N=400 for a long hunt


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rand_case(rng):
    B = int(rng.integers(1, 5))
    N = int(rng.choice([64, 257, 1000, 2048, 5000]))
    g = [int(rng.integers(1, 24)) for _ in range(3)]
    if rng.random() < 0.3:
        g = [int(rng.integers(1, 6))] * 3                      # few voxels: over-full buckets
    k = int(rng.choice([1, 3, 3, 5, 7]))
    P = int(rng.choice([1, 4, 8, 16, 32, 64, 128]))
    O = int(rng.choice([1, 7, 64, 200, 1024]))
    vs = [float(np.float32(2.0 / gi)) for gi in g]
    kind = rng.choice(["uniform", "clustered", "plane", "dups"])
    xyz = rng.uniform(-1, 1, (B, N, 3))
    if kind == "clustered":
        c = rng.uniform(-0.8, 0.8, (B, 4, 3))
        xyz = c[np.arange(B)[:, None], rng.integers(0, 4, (B, N))] + rng.normal(0, 0.05, (B, N, 3))
    elif kind == "plane":
        xyz[..., 2] = 0.1 * xyz[..., 0] + rng.normal(0, 0.003, (B, N))
    elif kind == "dups":
        xyz[:, N // 2:] = xyz[:, :N - N // 2]
    if rng.random() < 0.5:
        xyz *= 1.15                                            # some points leave the grid
    w = np.ones((B, N, 1))
    if rng.random() < 0.4:
        w = rng.integers(1, 5, (B, N, 1)).astype(np.float64)
    data = np.concatenate([xyz, w], 2).astype(np.float32)
    npn = np.full((B, 1), N, np.int32)
    if rng.random() < 0.5:
        npn = rng.integers(0, N + 1, (B, 1)).astype(np.int32)
    kw = dict(max_p_grid=P, max_o_grid=O, kernel_size=k, stride=1, loc=int(rng.integers(0, 2)),
              coord_shift=[1.0, 1.0, 1.0], voxel_size=vs, grid_size=g,
              seed=int(rng.integers(0, 2 ** 40)))
    return data, npn, kw


def same(want, got, what):
    for j, (w, g) in enumerate(zip(want, got)):
        g = g.cpu().numpy()
        assert w.shape == g.shape and w.tobytes() == g.tobytes(), \
            (what, j, int((w != g).sum()) if w.shape == g.shape else (w.shape, g.shape))


@pytest.mark.parametrize("seed", range(NFUZZ))
def test_gridify_family_random(seed):
    rng = np.random.default_rng(1000 + seed)
    data, npn, kw = rand_case(rng)
    d, n = T(data), T(npn)
    if kw["loc"] == 0:
        kw["loc"] = 1 if seed % 2 else 0
    same(orc.gridify(data, npn, **kw), ops.Gridify(d, n, **kw), ("gridify", seed, kw))
    same(orc.gridify_knn(data, npn, **kw), ops.GridifyKNN(d, n, **kw), ("gridify_knn", seed, kw))
    B, N = data.shape[:2]
    if B * N * kw["kernel_size"] ** 3 < 2 ** 31:
        kf = dict(kw)
        same(orc.gridify_fast_rand(data, npn, **kf), ops.Gridify_fast_rand(d, n, **kf),
             ("fast_rand", seed, kf))
    beta = float(rng.choice([0.0, 0.5, 1.0, 4.0]))
    same(orc.gridify_occaware(data, npn, beta=beta, **kw), ops.Gridify_occaware(d, n, beta=beta, **kw),
         ("occaware", seed, kw, beta))


@pytest.mark.parametrize("seed", range(max(NFUZZ // 2, 1)))
def test_gridify_up_and_knn_random(seed):
    rng = np.random.default_rng(2000 + seed)
    data, npn, kw = rand_case(rng)
    B, N = data.shape[:2]
    M = int(rng.choice([16, 100, 777, 2048]))
    up = np.concatenate([rng.uniform(-1.1, 1.1, (B, M, 3)), np.ones((B, M, 1))], 2).astype(np.float32)
    upn = rng.integers(0, M + 1, (B, 1)).astype(np.int32)
    ku = dict(max_p_grid=int(rng.choice([1, 3, 5, 8, 16])), max_o_grid=M,
              kernel_size=int(rng.choice([1, 3, 5])), coord_shift=kw["coord_shift"],
              voxel_size=kw["voxel_size"], grid_size=kw["grid_size"], seed=kw["seed"])
    same(orc.gridify_up(data, up, npn, upn, **ku), ops.GridifyUp(T(data), T(up), T(npn), T(upn), **ku),
         ("gridify_up", seed, ku))
    k = int(rng.choice([1, 3, 5, 6]))
    r = float(rng.choice([0.05, 0.2, 0.5]))
    dn = np.maximum(npn, 1).astype(np.int32)
    want = orc.ball_knn(up[..., :3], data[..., :3], dn, upn, k=k, radius=r)
    got = ops.BallKNN(T(up[..., :3].copy()), T(data[..., :3].copy()), T(dn), T(upn), k=k, radius=r)
    same((want,), (got,), ("ball_knn", seed, k, r))
    want = orc.knn(up[..., :3], data[..., :3], dn, upn, k=k)
    got = ops.KNN(T(up[..., :3].copy()), T(data[..., :3].copy()), T(dn), T(upn), k=k)
    same((want,), (got,), ("knn", seed, k))


BIG = [
    # B, N, grid, k, P, O   -- chunk sizes 1024..4096, slab splits, the legacy build beyond 2^22 voxels
    (8, 81920, [40, 40, 40], 3, 128, 1024),
    (3, 200000, [64, 64, 64], 3, 64, 16384),
    (2, 120000, [100, 100, 100], 3, 32, 4096),
    (2, 50000, [200, 200, 200], 3, 16, 2048),          # 8 M voxels: legacy index build
    (5, 30000, [31, 17, 5], 5, 64, 300),
    (1, 131072, [8, 8, 8], 7, 128, 512),               # every voxel over-full
    (16, 8192, [15, 15, 15], 3, 32, 256),
]


@pytest.mark.parametrize("case", BIG, ids=["%dx%d_g%d" % (c[0], c[1], c[2][0]) for c in BIG])
def test_gridify_large_random(case):
    B, N, g, k, P, O = case
    rng = np.random.default_rng(B * 7 + N)
    xyz = rng.uniform(-1.02, 1.02, (B, N, 3))
    xyz[:, : N // 3, 2] = 0.3 * xyz[:, : N // 3, 0] + rng.normal(0, 0.002, (B, N // 3))   # a wall
    data = np.concatenate([xyz, rng.integers(1, 3, (B, N, 1))], 2).astype(np.float32)
    npn = rng.integers(N // 2, N + 1, (B, 1)).astype(np.int32)
    kw = dict(max_p_grid=P, max_o_grid=O, kernel_size=k, stride=1, loc=1, coord_shift=[1.0, 1.0, 1.0],
              voxel_size=[float(np.float32(2.0 / gi)) for gi in g], grid_size=g, seed=99)
    d, n = T(data), T(npn)
    same(orc.gridify(data, npn, **kw), ops.Gridify(d, n, **kw), ("gridify", case))
    if k == 3:
        M = min(N, 20000)
        up = data[:, :M].copy()
        ku = dict(max_p_grid=5, max_o_grid=M, kernel_size=3, coord_shift=kw["coord_shift"],
                  voxel_size=kw["voxel_size"], grid_size=g, seed=5)
        upn = np.minimum(npn, M).astype(np.int32)
        same(orc.gridify_up(data, up, npn, upn, **ku), ops.GridifyUp(d, T(up), n, T(upn), **ku),
             ("gridify_up", case))
