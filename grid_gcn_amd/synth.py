"""Seeded synthetic point clouds and the layer tables of the reference configs.

No dataset is available offline, so tests and bench.py use these clouds.  Value distribution
follows the reference loaders: points normalised into the unit ball (utils/utils.py:47-61),
augmentation level 6 = scale U[0.8,1.25] + shift U[-0.1,0.1]^3
(data_loader/new_ggcn_gpu_modelnet_loader.py:224-230), so a few % of the points leave the
[0,2)^3 grid and exercise the drop rule of gridify.cu:136-138.
"""
import numpy as np


def make_cloud(n, cloud_id=0, kind="ball", aug=True):
    """One cloud [n,3] float32.  kind: 'ball' (uniform in unit ball) or 'planes'
    (70 % of the points projected onto 6 random planes: ScanNet-like occupancy)."""
    rng = np.random.default_rng(1234 + int(cloud_id))
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-12
    r = rng.random(n) ** (1.0 / 3.0)
    pts = d * r[:, None]
    if kind == "planes":
        nrm = rng.standard_normal((6, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        off = rng.uniform(-0.5, 0.5, 6)
        which = rng.integers(0, 6, n)
        on_plane = rng.random(n) < 0.7
        dist = np.einsum("ij,ij->i", pts, nrm[which]) - off[which]
        proj = pts - dist[:, None] * nrm[which]
        proj += rng.normal(0, 0.004, proj.shape)
        pts = np.where(on_plane[:, None], proj, pts)
        nr = np.linalg.norm(pts, axis=1)
        pts = np.where((nr > 1.0)[:, None], pts / (nr[:, None] + 1e-12), pts)
    elif kind != "ball":
        raise ValueError(kind)
    if aug:
        pts = pts * rng.uniform(0.8, 1.25) + rng.uniform(-0.1, 0.1, 3)[None, :]
    return pts.astype(np.float32)


def make_batch(B, n, kind="ball", aug=True, first_id=0):
    """data[B,n,4] float32 with w=1 and actual_numpoints[B,1] int32 = n
    (what the reference loaders feed: data + actual_centnum, SURVEY §2 'Data loaders')."""
    xyz = np.stack([make_cloud(n, first_id + b, kind, aug) for b in range(B)])
    data = np.concatenate([xyz, np.ones((B, n, 1), np.float32)], axis=2)
    return np.ascontiguousarray(data), np.full((B, 1), n, np.int32)


# ---- layer tables (the shape contract: SURVEY App. B) ---------------------------------------
# classification/configs/configs.yaml:47-63
CLS_MODELNET40 = dict(
    num_points=1024, coord_shift=[1.0, 1.0, 1.0], loc=1,
    down=[
        dict(voxel_size=[0.05] * 3, grid_size=[40] * 3, kernel_size=7, max_p_grid=64, max_o_grid=1024),
        dict(voxel_size=[0.25] * 3, grid_size=[8] * 3, kernel_size=3, max_p_grid=64, max_o_grid=128),
        dict(voxel_size=[2.0] * 3, grid_size=[1] * 3, kernel_size=1, max_p_grid=128, max_o_grid=1),
    ])

# segmentation/configs/configs.yaml:71-111  (8192-pt)
SEG_SCANNET_8192 = dict(
    num_points=8192, coord_shift=[1.0, 1.0, 1.0], loc=1,
    down=[
        dict(voxel_size=[0.05] * 3, grid_size=[40] * 3, kernel_size=3, max_p_grid=64, max_o_grid=1024),
        dict(voxel_size=[0.133333] * 3, grid_size=[15] * 3, kernel_size=3, max_p_grid=32, max_o_grid=256),
        dict(voxel_size=[0.4] * 3, grid_size=[5] * 3, kernel_size=3, max_p_grid=32, max_o_grid=24),
    ],
    up=[
        dict(voxel_size=[0.4] * 3, grid_size=[5] * 3, kernel_size=3, max_p_grid=5, max_o_grid=256),
        dict(voxel_size=[0.133333] * 3, grid_size=[15] * 3, kernel_size=3, max_p_grid=5, max_o_grid=1024),
        dict(voxel_size=[0.05] * 3, grid_size=[40] * 3, kernel_size=3, max_p_grid=5, max_o_grid=8192),
    ])

# segmentation/configs/configs.yaml:144-189 (commented 81920-pt block)
SEG_SCANNET_81920 = dict(
    num_points=81920, coord_shift=[1.0, 1.0, 1.0], loc=1,
    down=[
        dict(voxel_size=[0.05] * 3, grid_size=[40] * 3, kernel_size=3, max_p_grid=128, max_o_grid=1024),
        dict(voxel_size=[0.133333] * 3, grid_size=[15] * 3, kernel_size=3, max_p_grid=32, max_o_grid=256),
        dict(voxel_size=[0.4] * 3, grid_size=[5] * 3, kernel_size=3, max_p_grid=32, max_o_grid=24),
    ],
    up=[
        dict(voxel_size=[0.4] * 3, grid_size=[5] * 3, kernel_size=3, max_p_grid=5, max_o_grid=256),
        dict(voxel_size=[0.133333] * 3, grid_size=[15] * 3, kernel_size=3, max_p_grid=5, max_o_grid=1024),
        dict(voxel_size=[0.05] * 3, grid_size=[40] * 3, kernel_size=3, max_p_grid=5, max_o_grid=81920),
    ])

# builder-defined HBM stress config (BASELINE.json configs[4]; not in the reference)
SYNTH_200K = dict(
    num_points=200000, coord_shift=[1.0, 1.0, 1.0], loc=1,
    down=[
        dict(voxel_size=[2.0 / 64] * 3, grid_size=[64] * 3, kernel_size=3, max_p_grid=64, max_o_grid=16384),
        dict(voxel_size=[2.0 / 32] * 3, grid_size=[32] * 3, kernel_size=3, max_p_grid=64, max_o_grid=4096),
        dict(voxel_size=[2.0 / 16] * 3, grid_size=[16] * 3, kernel_size=3, max_p_grid=64, max_o_grid=1024),
        dict(voxel_size=[2.0 / 8] * 3, grid_size=[8] * 3, kernel_size=3, max_p_grid=64, max_o_grid=256),
    ])


def gridify_kwargs(cfg, layer, seed=0):
    """kwargs of one Gridify call site (segmentation/models/ggcn_models_g.py:154-159)."""
    L = cfg["down"][layer]
    return dict(max_p_grid=L["max_p_grid"], max_o_grid=L["max_o_grid"], kernel_size=L["kernel_size"],
                stride=1, loc=cfg["loc"], coord_shift=cfg["coord_shift"], voxel_size=L["voxel_size"],
                grid_size=L["grid_size"], seed=seed)


def gridify_up_kwargs(cfg, layer, seed=0):
    """kwargs of one GridifyUp call site (segmentation/models/ggcn_models_g.py:206-210)."""
    L = cfg["up"][layer]
    return dict(max_p_grid=L["max_p_grid"], max_o_grid=L["max_o_grid"], kernel_size=L["kernel_size"],
                coord_shift=cfg["coord_shift"], voxel_size=L["voxel_size"], grid_size=L["grid_size"],
                seed=seed)


def gridify_algorithmic_bytes(n, max_o_grid, max_p_grid):
    """bytes one cloud must move through a Gridify call."""
    return 16 * n + 8 * max_o_grid * max_p_grid + 20 * max_o_grid + 8
