"""Seeded GridConv layer cases (gridconv_seg_L0 / _L1, gridconv_up2,
gridconv_cls_L0) shared by the fixture generator (tests/golden/make_gridconv_golden.py), the CPU
tests and the GPU tests.  Index inputs come from the S0 oracle, weights from numpy RNGs
(oracle/gridconv_ref.make_mlp) -- nothing here touches grid_gcn_amd's modules.

Each builder returns a dict:
    kind      'seg' | 'cls'
    cent      [B,O,4]  centres (oracle Gridify), or the up points for the up layer
    src       [B,Nsrc,4+C] source rows (x,y,z,w,features)
    nebidx    [B,O,P] i32
    centmsk   [B,O] | None
    center_ori_feats [B,O,Cc] | None
    w         weights: dict of layer lists (oracle/gridconv_ref layout)
    spec      constructor arguments of the product module
    rows      slice of centre rows kept in the fixture
"""
import numpy as np

from grid_gcn_amd import synth
from oracle import gridconv_ref as ref
from oracle import oracle as orc


def _seg_l0():
    cfg = synth.SEG_SCANNET_8192
    data, npn = synth.make_batch(1, 8192, "planes", first_id=40)
    idx, _, cent, cmsk, cn = orc.gridify(data, npn, **synth.gridify_kwargs(cfg, 0))
    rng = np.random.default_rng(1001)
    w = dict(pt=ref.make_mlp(rng, 3, [32, 32, 64]), att1=ref.make_mlp(rng, 10, [16]),
             att2=ref.make_mlp(rng, 16, [64]))
    return dict(kind="seg", cent=cent, src=data, nebidx=idx, centmsk=cmsk, center_ori_feats=None,
                w=w, spec=dict(in_feats=0, pt_mlp=[32, 32, 64], localfdim=0, relu=True),
                rows=slice(None), next=(cent, cn))


def _seg_l1():
    """layer 1 of the 81920-pt config's style: features + geo_vec (localfdim 3), relu False."""
    cfg = synth.SEG_SCANNET_8192
    l0 = _seg_l0()
    cent0, cn0 = l0["next"]
    rng = np.random.default_rng(1002)
    feats0 = rng.normal(0, 1, (1, cent0.shape[1], 64)).astype(np.float32)
    src = np.concatenate([cent0, feats0], axis=2)
    idx, _, cent, cmsk, _ = orc.gridify(cent0, cn0, **synth.gridify_kwargs(cfg, 1))
    w = dict(pt=ref.make_mlp(rng, 67, [64, 64, 128]), att1=ref.make_mlp(rng, 10, [32]),
             att2=ref.make_mlp(rng, 32, [128]))
    return dict(kind="seg", cent=cent, src=src, nebidx=idx, centmsk=cmsk, center_ori_feats=None,
                w=w, spec=dict(in_feats=64, pt_mlp=[64, 64, 128], localfdim=3, relu=False),
                rows=slice(None))


def _seg_up2():
    """last up layer (ggcn_models_g.py:191-231): BallKNN indices incl. -1, centre MLP on the up
    points' own (x,y,z,w), concat, update MLP; no centre mask."""
    cfg = synth.SEG_SCANNET_8192
    l0 = _seg_l0()
    cent0, cn0 = l0["next"]
    data, npn = synth.make_batch(1, 8192, "planes", first_id=40)
    rng = np.random.default_rng(1003)
    feats = rng.normal(0, 1, (1, cent0.shape[1], 128)).astype(np.float32)
    src = np.concatenate([cent0, feats], axis=2)
    U = cfg["up"][2]
    radius = U["voxel_size"][0] * U["kernel_size"] * 1.7 / 2
    idx = orc.ball_knn(data[..., :3], cent0[..., :3], cn0, npn, k=U["max_p_grid"], radius=radius)
    w = dict(pt=ref.make_mlp(rng, 128, [128]), att1=ref.make_mlp(rng, 10, [32]),
             att2=ref.make_mlp(rng, 32, [128]), center=ref.make_mlp(rng, 4, [128]),
             update=ref.make_mlp(rng, 256, [128]))
    return dict(kind="seg", cent=data, src=src, nebidx=idx, centmsk=None, center_ori_feats=data,
                w=w, spec=dict(in_feats=128, pt_mlp=[128], localfdim=0, relu=True, center_in=4,
                               center_dim=[128], out_dim=[128]),
                rows=slice(0, None, 16))


def _cls_l0():
    cfg = synth.CLS_MODELNET40
    data, npn = synth.make_batch(1, 1024, "ball", first_id=50)
    idx, _, cent, cmsk, _ = orc.gridify(data, npn, **synth.gridify_kwargs(cfg, 0))
    rng = np.random.default_rng(1004)
    w = dict(pt=ref.make_mlp(rng, 3, [64, 64, 128]), att1=ref.make_mlp(rng, 4, [64]),
             att2=ref.make_mlp(rng, 64 + 128 + 3, [128, 128]))
    return dict(kind="cls", cent=cent, src=data, nebidx=idx, centmsk=cmsk, center_ori_feats=None,
                w=w, spec=dict(in_feats=0, pt_mlp=[64, 64, 128], att_ele=[64, 128, 128],
                               localfdim=3, relu=True),
                rows=slice(0, None, 2))


CASES = {"gridconv_seg_L0": _seg_l0, "gridconv_seg_L1": _seg_l1, "gridconv_up2": _seg_up2,
         "gridconv_cls_L0": _cls_l0}


def reference_output(case, train, dtype=np.float64):
    """The layer output [B,O,C'] (channels last) of the independent restatement."""
    nb = orc.batch_take(case["src"], case["nebidx"]).astype(dtype)         # [B,O,P,4+C] (ops.py:78-93)
    neighbors = np.transpose(nb, (0, 3, 1, 2))                             # NCHW (ggcn_models_g.py:172-175)
    centers_xyz = np.transpose(case["cent"][..., 0:3].astype(dtype), (0, 2, 1))
    cm = None if case["centmsk"] is None else case["centmsk"].astype(dtype)
    sp = case["spec"]
    if case["kind"] == "seg":
        cof = None
        if case["center_ori_feats"] is not None:
            cof = np.transpose(case["center_ori_feats"].astype(dtype), (0, 2, 1))
        out = ref.sub_g_update_seg(centers_xyz, neighbors, sp["in_feats"] > 0, cm, case["w"],
                                   localfdim=sp["localfdim"], relu=sp["relu"], train=train,
                                   center_ori_feats=cof)
    else:
        out = ref.sub_g_update_cls(centers_xyz, neighbors, sp["in_feats"] > 0, cm, case["w"],
                                   localfdim=sp["localfdim"], relu=sp["relu"], train=train)
    return np.transpose(out, (0, 2, 1))


def load_layers(seq, layers):
    """copy one oracle layer list into a torch stack of gridconv.ConvBNReLU modules."""
    import torch
    assert len(seq) == len(layers)
    with torch.no_grad():
        for m, p in zip(seq, layers):
            m.lin.weight.copy_(torch.from_numpy(p["W"]))
            m.lin.bias.copy_(torch.from_numpy(p["b"]))
            m.bn.weight.copy_(torch.from_numpy(p["gamma"]))
            m.bn.bias.copy_(torch.from_numpy(p["beta"]))
            m.bn.running_mean.copy_(torch.from_numpy(p["rmean"]))
            m.bn.running_var.copy_(torch.from_numpy(p["rvar"]))


def build_module_f64(case):
    """the same module in float64 (exact comparison with the restatement)."""
    return build_module(case, double=True)


def build_module(case, double=False):
    """the product module of the case with the oracle's weights."""
    from grid_gcn_amd import gridconv, model_cls
    sp, w = case["spec"], case["w"]
    if case["kind"] == "seg":
        m = gridconv.SubGUpdate(sp["in_feats"], sp["pt_mlp"], sp["localfdim"], sp["relu"],
                                center_in=sp.get("center_in"), center_dim=sp.get("center_dim", ()),
                                out_dim=sp.get("out_dim", ()))
    else:
        m = model_cls.SubGUpdateCls(sp["in_feats"], sp["pt_mlp"], sp["att_ele"], sp["localfdim"],
                                    sp["relu"])
    if double:
        m = m.double()
    load_layers(m.pt_mlp, w["pt"]); load_layers(m.att1, w["att1"]); load_layers(m.att2, w["att2"])
    if getattr(m, "center_mlp", None) is not None:
        load_layers(m.center_mlp, w["center"])
    if getattr(m, "update_mlp", None) is not None:
        load_layers(m.update_mlp, w["update"])
    return m
