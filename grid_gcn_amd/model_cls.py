"""ModelNet40 classification network `get_symbol_cls_ggcn`
(classification/models/ggcn_models_g.py:37-111) and its GridConv variant
(classification/models/gcn_module_g.py:64-114 verts_pair_func, :116-209 sub_g_update,
:212-223 contextvec_func) restated for PyTorch-ROCm on top of the HIP index operators.

Shipped config (classification/configs/configs.yaml:47-63): attfdim=4, localfdim=3,
att_full='next', cntxt_mlp_lst=[[],[],[]] (context = max over P of the raw edge features, tiled),
att_ele_dim, gcn_outDim=[[],[],[]], relu=True, group_all=False, 3 Gridify layers (k = 7/3/1; the
last is a 1-voxel grid = global pooling).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import synth
from .gridconv import ConvBNReLU, check_shipped_branches, mlp, run_mlp
from .model import HipIndexOps, WeightedGradient, call_seed, release_packs

CLS_MN40 = dict(
    grid=synth.CLS_MODELNET40, inputDim=[0, 128, 256],
    pt_ele_dim=[[64, 64, 128], [128, 128, 256], [256, 256, 512]],
    att_ele_dim=[[64, 128, 128], [128, 256, 256], [256, 512, 512]],
    localfdim=3, attfdim=4, relu=True, num_classes=40, bn_decay=0.9, dropout=0.5)


class SubGUpdateCls(nn.Module):
    """classification sub_g_update: pt-MLP, attention MLP fed with
    concat(att1(att_vec), pt-MLP output, context) (att_full='next' + contextvec), product, max."""

    def __init__(self, in_feats, pt_mlp, att_ele, localfdim=3, relu=True, bn_decay=0.9, attfdim=4,
                 aggtype="gcn", pool_type="max_pooling", att_full="next", elevation=(),
                 up_center_inte="concat", cntxt_mlp=()):
        super().__init__()
        check_shipped_branches("classification", attfdim=attfdim, localfdim=localfdim,
                               aggtype=aggtype, pool_type=pool_type, att_full=att_full,
                               elevation=elevation, up_center_inte=up_center_inte,
                               cntxt_mlp=cntxt_mlp)
        self.has_feats = in_feats > 0
        self.localfdim = localfdim
        self.relu = relu
        cin = 3 if not self.has_feats else in_feats + (3 if localfdim != 0 else 0)
        self.cin = cin
        C = pt_mlp[-1]
        att_ele = list(att_ele)
        att_ele[-1] = C                                        # gcn_module_g.py:86
        self.pt_mlp = mlp(cin, pt_mlp, bn_decay)
        self.att1 = mlp(4, [att_ele[0]], bn_decay)             # :88-89  (attfdim = 4)
        self.att2 = mlp(att_ele[0] + C + cin, att_ele[1:], bn_decay)   # :93-101
        self.out_channels = C

    mfma_train = True

    def forward_src(self, cent, src, nebidx, center_masks=None):
        """On the GPU from the un-gathered source points: the whole edge block on the hand-written
        kernels (tcls._EdgeBlockClsTrain; evaluation under no_grad: edge_block_cls_eval with
        the running statistics) -- no gathered [E, 4+C] tensor, no concat, no tiled context.  None when the shapes are outside the kernels' domain (the caller
        then gathers and uses forward())."""
        from .train import cls as tcls, evalpath as teval
        train = self.training and torch.is_grad_enabled()
        infer = not self.training and not torch.is_grad_enabled()
        if not (self.mfma_train and src.is_cuda and (train or infer)):
            return None
        if self.has_feats and self.localfdim == 0:
            return None
        pt, a1, a2 = list(self.pt_mlp), list(self.att1), list(self.att2)
        if not tcls.edge_block_cls_supported(pt, a1, a2, src, nebidx.shape[2]):
            return None
        if train:
            agg = tcls.edge_block_cls_train(src, nebidx, cent, pt, a1, a2)
        else:
            agg = teval.edge_block_cls_eval(src, nebidx, cent, pt, a1, a2)
        # (relu of a product of two relu outputs is the identity: gcn_module_g.py's relu=True)
        if center_masks is not None:
            agg = agg * center_masks[..., None]
        return agg

    def forward(self, centers_xyz, neighbors, center_masks=None):
        nbr_xyz = neighbors[..., 0:3]
        geo_vec = nbr_xyz - centers_xyz[:, :, None, :]
        geo_dist = torch.sqrt(torch.sum(geo_vec * geo_vec, dim=-1, keepdim=True))
        att_vec = torch.cat([geo_dist, geo_vec], dim=-1)                       # attfdim == 4
        if not self.has_feats:
            nf0 = geo_vec
        elif self.localfdim != 0:
            nf0 = torch.cat([geo_vec, neighbors[..., 4:]], dim=-1)
        else:
            nf0 = neighbors[..., 4:]
        ctx = nf0.max(dim=2, keepdim=True).values.expand_as(nf0)               # contextvec_func
        # training on the GPU: every conv+BN+ReLU stack the MFMA kernels take (<= 256 output and
        # <= 384 input channels) runs through them, the wider ones through the stock modules
        nf = run_mlp(list(self.pt_mlp), nf0, self.mfma_train)
        a1 = run_mlp(list(self.att1), att_vec, self.mfma_train)
        att = run_mlp(list(self.att2), torch.cat([a1, nf, ctx], dim=-1), self.mfma_train)
        agg = (att * nf).max(dim=2).values
        if self.relu:
            agg = F.relu(agg)
        if center_masks is not None:
            agg = agg * center_masks[..., None]
        return agg


class FCBNReLU(nn.Module):
    """fully_connected of utils/ops.py:205-216: FC -> BN -> ReLU -> Dropout."""

    def __init__(self, cin, cout, bn_decay, dropout):
        super().__init__()
        self.l = ConvBNReLU(cin, cout, bn_decay)
        self.dropout = dropout

    def forward(self, x):
        # (run_mlp: the hand-written kernels in training and evaluation on the GPU, the stock module elsewhere)
        return F.dropout(run_mlp([self.l], x), self.dropout, self.training)


class GGCNCls(nn.Module):
    def __init__(self, cfg=CLS_MN40, index_ops=HipIndexOps, seed=0, fixed_seed=False):
        """seed / fixed_seed: as GGCNSeg (training redraws the voxel sampling at every call)."""
        super().__init__()
        self.cfg, self.ix, self.seed = cfg, index_ops, seed
        self.fixed_seed = fixed_seed
        self.forward_no = 0
        self.register_forward_hook(release_packs)
        self.seed_dev = None      # see GGCNSeg
        self._take_kw = (dict(neighbour_index=True)
                         if isinstance(index_ops, type) and issubclass(index_ops, HipIndexOps) else {})
        self.layers = nn.ModuleList(
            SubGUpdateCls(cfg["inputDim"][i], cfg["pt_ele_dim"][i], cfg["att_ele_dim"][i],
                          cfg["localfdim"], cfg["relu"], cfg["bn_decay"])
            for i in range(len(cfg["grid"]["down"])))
        c = self.layers[-1].out_channels * cfg["grid"]["down"][-1]["max_o_grid"]
        self.fc1 = FCBNReLU(c, 512, cfg["bn_decay"], cfg["dropout"])          # get_cls_head :29-34
        self.fc2 = FCBNReLU(512, 256, cfg["bn_decay"], cfg["dropout"])
        self.fc3 = nn.Linear(256, cfg["num_classes"])
        nn.init.xavier_uniform_(self.fc3.weight)
        nn.init.zeros_(self.fc3.bias)

    def forward(self, data_xyz, actual_centnum):
        """data_xyz [B,N,3], actual_centnum [B,1] i32 -> logits [B,40]."""
        g, ix = self.cfg["grid"], self.ix
        data = torch.cat([data_xyz, torch.ones_like(data_xyz[..., :1])], dim=2)   # :55
        data_loc, num = data, actual_centnum
        fwd_no = self.forward_no
        if self.training:
            self.forward_no += 1
            if data_xyz.is_cuda and torch.is_grad_enabled():
                from .train import common as tcommon, head as thead
                tcommon.PACKS.prepack(self)   # all weight layouts of the step, one launch
        for i, layer in enumerate(self.layers):
            seed = self.seed if (self.fixed_seed or not self.training) else \
                call_seed(self.seed, fwd_no, i)
            seed_dev = self.seed_dev if (self.training and not self.fixed_seed) else None
            sd = dict(seed_dev=seed_dev) if (seed_dev is not None and self._take_kw) else {}
            nebidx, nebidxmsk, cent, centmsk, num = ix.Gridify(
                data_loc.detach().contiguous(), num, **synth.gridify_kwargs(g, i, seed), **sd)
            data_loc = cent
            cf = layer.forward_src(cent, data, nebidx, centmsk) if self._take_kw else None
            if cf is None:
                neighbors = ix.batch_take_g(data.contiguous(), nebidx, **self._take_kw)  # :94
                cf = layer(cent[..., 0:3], neighbors, centmsk)                        # :104
            data = torch.cat([cent, cf], dim=2)                                   # :106
        net = cf.reshape(cf.shape[0], -1)                                         # flatten=True
        h = self.fc2(self.fc1(net))
        if h.is_cuda and self._take_kw:
            from .train import common as tcommon, head as thead
            return thead.linear_mm(h, self.fc3)
        return self.fc3(h)


def cls_loss(logits, label, weights=None):
    """SoftmaxOutput(normalization='batch') (ggcn_models_g.py:34), behind the optional
    'weighted_gradient' op (:32-33) when per-class weights are given."""
    if weights is not None:
        logits = WeightedGradient.apply(logits, torch.as_tensor(weights, dtype=logits.dtype,
                                                                device=logits.device))
    return F.cross_entropy(logits, label.long(), reduction="mean")


def cls_forward_flops(net, B):
    """Algorithmic flops of one forward of GGCNCls on B clouds: 2 * edges *
    sum(Cin*Cout) over the pt / att1 / att2 convs of every layer (the att2 input counts its full
    concat width, as the reference computes it) + the FC head."""
    g = net.cfg["grid"]
    fl = 0.0
    for i, layer in enumerate(net.layers):
        L = g["down"][i]
        e = B * L["max_o_grid"] * L["max_p_grid"]
        macs = sum(m.lin.in_features * m.lin.out_features
                   for seq in (layer.pt_mlp, layer.att1, layer.att2) for m in seq)
        fl += 2.0 * e * macs
    head = (net.fc1.l.lin.in_features * net.fc1.l.lin.out_features +
            net.fc2.l.lin.in_features * net.fc2.l.lin.out_features +
            net.fc3.in_features * net.fc3.out_features)
    return fl + 2.0 * B * head
