"""Seeded parity cases shared by the golden-fixture generator, the CPU tests and the GPU tests.

Each case is (name, builder) where builder() returns (op, args, kwargs) with numpy inputs.
Inputs are regenerated from seeds (numpy default_rng), so fixtures only hold outputs.
"""
import numpy as np

from grid_gcn_amd import synth


def _chain(cfg, B, kind, upto, oracle_gridify, seed=0, first_id=0):
    """inputs of layer `upto` = oracle outputs of the previous layers (layer chaining,
    segmentation/models/ggcn_models_g.py:154-166)."""
    data, npnts = synth.make_batch(B, cfg["num_points"], kind, first_id=first_id)
    outs = None
    for l in range(upto):
        outs = oracle_gridify(data, npnts, **synth.gridify_kwargs(cfg, l, seed))
        data, npnts = outs[2], outs[4]
    return data, npnts


def gridify_cases(oracle_gridify):
    cases = []
    for l in range(3):
        cases.append(("gridify_mn40_L%d_b1" % l,
                      lambda l=l: (_chain(synth.CLS_MODELNET40, 1, "ball", l, oracle_gridify),
                                   synth.gridify_kwargs(synth.CLS_MODELNET40, l))))
    for l in range(3):
        cases.append(("gridify_scan8k_L%d" % l,
                      lambda l=l: (_chain(synth.SEG_SCANNET_8192, 2, "planes", l, oracle_gridify),
                                   synth.gridify_kwargs(synth.SEG_SCANNET_8192, l))))

    def overflow():
        rng = np.random.default_rng(77)
        xyz = rng.uniform(-0.99, -0.01, (2, 4096, 3)).astype(np.float32)   # 2 voxels of a 2x1x1 grid
        xyz[..., 0] = rng.uniform(-0.99, 0.99, (2, 4096)).astype(np.float32)
        data = np.concatenate([xyz, np.ones((2, 4096, 1), np.float32)], 2)
        kw = dict(max_p_grid=8, max_o_grid=1, kernel_size=3, stride=1, loc=1, coord_shift=[1, 1, 1],
                  voxel_size=[1.0, 1.0, 1.0], grid_size=[2, 1, 1], seed=0)
        return (data, np.array([[4096], [3000]], np.int32)), kw
    cases.append(("gridify_overflow", overflow))

    def oob():
        rng = np.random.default_rng(78)
        xyz = rng.uniform(-1.2, 1.2, (2, 2048, 3)).astype(np.float32)
        xyz[0, :16, 0] = 1.0          # exactly on the upper face: (1+1)/0.05 == 40 -> dropped
        xyz[0, 16:32, 1] = -1.0       # exactly on the lower face: kept
        xyz[1, :8] = np.float32(np.nan)
        xyz[1, 8:12] = np.float32(np.inf)
        data = np.concatenate([xyz, np.ones((2, 2048, 1), np.float32)], 2)
        kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 0)
        kw.update(max_o_grid=512, max_p_grid=16)
        return (data, np.array([[2048], [2000]], np.int32)), kw
    cases.append(("gridify_oob", oob))

    for seed in (1, 123456, 2 ** 40 + 17):
        def seeded(seed=seed):
            data, npnts = synth.make_batch(2, 8192, "planes", first_id=10)
            kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 0, seed)
            kw.update(max_p_grid=8, max_o_grid=300)   # force every reservoir
            return (data, npnts), kw
        cases.append(("gridify_seeded_%d" % seed, seeded))

    def weights():
        # non-integer / large weights: S0's float accumulation order matters (gridify.cu:258,268)
        rng = np.random.default_rng(79)
        data, npnts = synth.make_batch(2, 4096, "ball", first_id=20)
        data[0, :, 3] = rng.uniform(0.0, 3.0, 4096).astype(np.float32)
        data[1, :, 3] = rng.integers(1, 1 << 20, 4096).astype(np.float32)
        kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 1)
        kw.update(max_p_grid=16, max_o_grid=64)
        return (data, npnts), kw
    cases.append(("gridify_weights", weights))

    def k7dense():
        data, npnts = synth.make_batch(1, 4096, "ball", first_id=30)
        kw = synth.gridify_kwargs(synth.CLS_MODELNET40, 0)
        kw.update(voxel_size=[0.1] * 3, grid_size=[20] * 3, max_p_grid=32, max_o_grid=256)
        return (data, npnts), kw
    cases.append(("gridify_k7_dense", k7dense))

    def ragged():
        data, npnts = synth.make_batch(3, 1000, "ball", first_id=40)   # N not a multiple of 64
        npnts = np.array([[1000], [0], [37]], np.int32)               # empty and short clouds
        kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 0)
        kw.update(max_o_grid=100, loc=0)
        return (data, npnts), kw
    cases.append(("gridify_ragged", ragged))

    def bigvox():
        # 168 x 168 x 160 = 4.5 M voxels: beyond the two-level split's 1024 slabs x 4096 voxels (2^22), i.e. the
        # index build of csrc/gridgcn_index_legacy.hip (round-1 kernels kept as the large-grid fallback) -- a golden
        # case of its own (hygiene): over-full buckets and both reservoirs included
        data, npnts = synth.make_batch(2, 8192, "planes", first_id=90)
        data[:, :3000, :3] *= np.float32(0.03)        # a dense clump: buckets beyond P, neighbourhoods beyond P
        npnts = np.array([[8192], [6000]], np.int32)
        kw = dict(max_p_grid=8, max_o_grid=700, kernel_size=3, stride=1, loc=1, coord_shift=[1, 1, 1],
                  voxel_size=[2.0 / 168, 2.0 / 168, 2.0 / 160], grid_size=[168, 168, 160], seed=11)
        return (data, npnts), kw
    cases.append(("gridify_legacy_4m_voxels", bigvox))
    return cases


def gridify_knn_cases(oracle_gridify):
    def scan():
        data, npnts = synth.make_batch(2, 8192, "planes")
        kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 0)
        kw.update(max_p_grid=32)
        return (data, npnts), kw

    def sparse():   # fewer than P candidates in all shells -> padding rule
        data, npnts = synth.make_batch(2, 512, "ball", first_id=50)
        kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 0)
        kw.update(max_p_grid=16, max_o_grid=128, kernel_size=5)
        return (data, npnts), kw

    def dense():    # many candidates per shell, bucket reservoir active
        data, npnts = synth.make_batch(1, 8192, "ball", first_id=51)
        kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 1)
        kw.update(max_p_grid=24, max_o_grid=200)
        return (data, npnts), kw
    return [("gridify_knn_scan8k_L0", scan), ("gridify_knn_sparse_k5", sparse),
            ("gridify_knn_dense", dense)]


def gridify_up_cases(oracle_gridify):
    cases = []
    cfg = synth.SEG_SCANNET_8192
    for u in range(3):
        def up(u=u):
            # up layer u: down set = centres of down layer 2-u, up set = centres of layer 1-u
            # (or the input points), segmentation/models/ggcn_models_g.py:191-210
            data, npnts = synth.make_batch(2, cfg["num_points"], "planes")
            levels = [(data, npnts)]
            for l in range(3):
                o = oracle_gridify(levels[-1][0], levels[-1][1], **synth.gridify_kwargs(cfg, l))
                levels.append((o[2], o[4]))
            down, dn = levels[3 - u]
            upd, un = levels[2 - u]
            if u == 2:
                upd = upd.copy()
                upd[0, 5, :3] = [0.97, 0.97, 0.97]    # an up point whose k^3 region is empty
            return (down, upd, dn, un), synth.gridify_up_kwargs(cfg, u)
        cases.append(("gridify_up_scan8k_%d" % u, up))

    def upovf():
        data, npnts = synth.make_batch(2, 2048, "ball", first_id=60)
        upd, un = synth.make_batch(2, 512, "ball", first_id=70)
        un = np.array([[512], [300]], np.int32)
        kw = dict(max_p_grid=4, max_o_grid=512, kernel_size=3, coord_shift=[1, 1, 1],
                  voxel_size=[0.25] * 3, grid_size=[8] * 3, seed=5)
        return (data, upd, npnts, un), kw
    cases.append(("gridify_up_overflow", upovf))
    return cases


def knn_cases():
    cases = []
    for (n, m, r) in ((256, 24, 1.02), (1024, 256, 0.34), (3000, 1500, 0.1275)):
        def c(n=n, m=m, r=r):
            rng = np.random.default_rng(n + m)
            un = rng.uniform(-1, 1, (2, n, 3)).astype(np.float32)
            kn = rng.uniform(-1, 1, (2, m, 3)).astype(np.float32)
            kn[0, 3] = kn[0, 1]                       # duplicate points: tie-break by index
            dn = np.array([[m], [max(1, m - 7)]], np.int32)
            upn = np.array([[n], [n - 5]], np.int32)
            return (un, kn, dn, upn), dict(k=5, radius=r)
        cases.append(("ball_knn_%d_%d" % (n, m), c))

    def few():
        rng = np.random.default_rng(5)
        un = rng.uniform(-1, 1, (1, 64, 3)).astype(np.float32)
        kn = rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32)
        return (un, kn, np.array([[3]], np.int32), np.array([[64]], np.int32)), dict(k=6, radius=5.0)
    cases.append(("ball_knn_few", few))
    return cases


def gridify_variant_cases():
    """(name, build) of the two operator variants: `occaware_*` = Gridify_occaware (coverage-aware
    sampling, OUR specification -- parity unpinned), `fastrand_*` = the fast_rand build of Gridify."""
    cases = []

    def occ(beta, O, kind):
        def build():
            data, npnts = synth.make_batch(2, 8192, kind, first_id=30)
            kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, 0, 7)
            kw.update(max_o_grid=O, beta=beta)
            return (data, npnts), kw
        return build
    cases.append(("occaware_scan8k_b1", occ(1.0, 256, "planes")))
    cases.append(("occaware_scan8k_b0", occ(0.0, 100, "ball")))

    def fr(layer, over):
        def build():
            data, npnts = synth.make_batch(2, 8192, "planes", first_id=40)
            kw = synth.gridify_kwargs(synth.SEG_SCANNET_8192, layer)
            kw.update(over)
            return (data, npnts), kw
        return build
    cases.append(("fastrand_scan8k_L0", fr(0, {})))
    cases.append(("fastrand_scan8k_overfull", fr(1, dict(max_p_grid=8, max_o_grid=300))))
    cases.append(("fastrand_scan8k_loc0", fr(0, dict(loc=0, max_p_grid=16))))
    return cases
