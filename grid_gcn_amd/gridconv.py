"""GridConv layer (`sub_g_update`) and its primitives, restated for PyTorch-ROCm.

Follows segmentation/models/gcn_module_g_att.py:172-287 (sub_g_update), :120-170
(verts_pair_func), :45-79 (aggregation_func), :24-43 (update_func) and the 1x1-conv primitives of
utils/ops.py:141-158,236-260 (Convolution(1x1, bias) -> BatchNorm(eps=1e-3, momentum=bn_decay,
fix_gamma=False) -> ReLU).  Tensors are edge-major / channels-last ([B,O,P,C]); a 1x1 conv on
NCHW is the same contraction as a Linear on the last axis, BatchNorm statistics run over every
other axis exactly as MXNet's axis=1 BatchNorm on [B,C,O,P].

Two execution paths with identical semantics:
  * "torch": every op is a stock PyTorch op (rocBLAS/MIOpen underneath) -- the fp32 reference
    the fused HIP kernels are checked against;
  * "fused": gather + geo features + per-edge MLPs + attention product + max over P inside the
    hand-written gfx950 kernels of csrc/gridgcn_conv.hip (inference-mode BatchNorm).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-3  # MXNet BatchNorm default eps


class ConvBNReLU(nn.Module):
    """conv1d/conv2d(kernel 1) + BatchNorm + ReLU of utils/ops.py:141-158 on channels-last input."""

    def __init__(self, cin, cout, bn_decay=0.9, use_bn=True, use_relu=True):
        super().__init__()
        self.lin = nn.Linear(cin, cout, bias=True)
        nn.init.xavier_uniform_(self.lin.weight)      # mx.init.Xavier (base_solver.py:62)
        nn.init.zeros_(self.lin.bias)
        # MXNet momentum m: running = m*running + (1-m)*batch ; torch uses (1-m)
        self.bn = nn.BatchNorm1d(cout, eps=BN_EPS, momentum=1.0 - bn_decay) if use_bn else None
        self.use_relu = use_relu

    def forward(self, x):
        y = self.lin(x)
        if self.bn is not None:
            shp = y.shape
            y = self.bn(y.reshape(-1, shp[-1])).reshape(shp)
        return F.relu(y) if self.use_relu else y

    def folded(self):
        """(W^T [cin,cout], bias [cout]) with inference-mode BatchNorm folded in."""
        w = self.lin.weight.detach().t().contiguous()
        b = self.lin.bias.detach().clone()
        if self.bn is not None:
            s = self.bn.weight.detach() / torch.sqrt(self.bn.running_var + self.bn.eps)
            w = w * s[None, :]
            b = (b - self.bn.running_mean) * s + self.bn.bias.detach()
        return w.contiguous(), b.contiguous()


def mlp(cin, dims, bn_decay=0.9):
    """mlp2d_c / mlp1d_c (utils/ops.py:236-260)."""
    layers = []
    for d in dims:
        layers.append(ConvBNReLU(cin, d, bn_decay))
        cin = d
    return nn.Sequential(*layers)


_warned_shapes = set()


def _warn_stock_fallback(layers, x):
    """The MFMA training kernels take stacks of <= 256 output / <= 384 input channels (fp32,
    contiguous rows); wider conv+BN+ReLU stacks run on 256-column slices of those kernels or on
    csrc/gridgcn_gemm.hip + this library's BatchNorm kernels
    (tmlp.mlp_wide_train); anything else (no BatchNorm, no ReLU, other dtypes) on the stock
    PyTorch modules -- several times slower.  Say so once per shape instead of silently."""
    key = (tuple((l.lin.in_features, l.lin.out_features) for l in layers), str(x.dtype))
    if key not in _warned_shapes:
        _warned_shapes.add(key)
        import warnings
        warnings.warn("grid_gcn_amd: conv+BN+ReLU stack %s (%s) is outside the hand-written training "
                      "kernels' domain; running it on the stock PyTorch modules" % (key[0], key[1]),
                      RuntimeWarning, stacklevel=3)


def run_mlp(layers, x, mfma=True):
    """A stack of ConvBNReLU layers.  Training on the GPU: one autograd op over the hand-written
    kernels (tmlp.mlp_bn_relu_train); otherwise the stock PyTorch modules."""
    if mfma and x.is_cuda and layers and layers[0].training and torch.is_grad_enabled():
        from .train import common as tcommon, evalpath as teval, mlp as tmlp
        if tcommon.supported(layers, x):
            return tmlp.mlp_bn_relu_train(x, layers)
        if tmlp.wide_supported(layers, x):
            # beyond the MFMA kernels' widths: 256-column slices / gridgcn_gemm + this library's BatchNorm kernels
            return tmlp.mlp_wide_train(x, layers)
        _warn_stock_fallback(layers, x)
    if mfma and x.is_cuda and layers and not layers[0].training and not torch.is_grad_enabled():
        from .train import common as tcommon, evalpath as teval, mlp as tmlp
        if tcommon.supported(layers, x) and all(l.lin.in_features <= 1024 for l in layers):
            return teval.mlp_bn_relu_eval(x, layers)      # same MFMA kernel, running stats
    for l in layers:
        x = l(x)
    return x


def geo_features(neighbors, centers_xyz):
    """geo_vec, geo_dist and the attfdim=10 attention input (gcn_module_g_att.py:190-194, 217-218).
    neighbors [B,O,P,4+C], centers_xyz [B,O,3]."""
    nbr_xyz = neighbors[..., 0:3]
    cexp = centers_xyz[:, :, None, :].expand_as(nbr_xyz)
    geo_vec = nbr_xyz - cexp
    geo_dist = torch.sqrt(torch.sum(geo_vec * geo_vec, dim=-1, keepdim=True))
    att_vec = torch.cat([geo_dist, geo_vec, cexp, nbr_xyz], dim=-1)
    return geo_vec, geo_dist, att_vec


class Tail:
    """What directly follows a GridConv block in the CALLER (the segmentation head behind the last
    up layer): conv+BN+ReLU `layers` that may join the block's update chain, and optionally the
    Dropout + class-score Linear behind them, head = (p, nn.Linear, seed_dev | None[, host seed |
    None]).  A per-call
    request object: finish() reports in `done` what it absorbed (0 nothing, 1 the layers, 2 the
    head as well) -- the module itself keeps no per-call state."""

    def __init__(self, layers=(), head=None):
        self.layers, self.head, self.done = tuple(layers), head, 0


# Branches of the reference's sub_g_update that are restated here.  Everything else in its
# signature / `configs` (segmentation/models/gcn_module_g_att.py:172-287,
# classification/models/gcn_module_g.py:116-209) is used by none of the shipped yaml files and is
# REFUSED rather than silently ignored.
_SHIPPED = {
    "segmentation": dict(attfdim=(10,), localfdim=(0, 3), aggtype=("gcn",),
                         pool_type=("max", "max_pooling"), att_full=("",),
                         up_center_inte=("concat",)),
    "classification": dict(attfdim=(4,), localfdim=(0, 3), aggtype=("gcn",),
                           pool_type=("max", "max_pooling"), att_full=("next",),
                           up_center_inte=("concat",)),
}
_BRANCH_REF = dict(
    attfdim="gcn_module_g_att.py:196-216 (attfdim 5/11/12 add point-density features)",
    localfdim="gcn_module_g_att.py:220-233 (localfdim 4/5/10/11/12 geo_feats)",
    aggtype="gcn_module_g_att.py:262-264 (agg_gcn: pooling before the MLP)",
    pool_type="gcn_module_g_att.py:45-79 (aggregation_func: only max pooling is restated)",
    att_full="gcn_module_g_att.py:143-148 ('last' / 'next' concat in front of the second attention conv)",
    up_center_inte="gcn_module_g_att.py:279-282 ('add' instead of concat)",
    elevation="gcn_module_g_att.py:243-247 (elevation MLP on the geometric features of a first layer)",
    cntxt_mlp="gcn_module_g_att.py:253-255 (context MLP; only the classifier's empty one is restated)")


def check_shipped_branches(family, elevation=(), cntxt_mlp=None, **opts):
    """NotImplementedError for an option value outside what this package restates."""
    ok = _SHIPPED[family]
    for k, v in opts.items():
        if v not in ok[k]:
            raise NotImplementedError(
                "%s sub_g_update with %s=%r is not restated here (supported: %s); reference branch: %s"
                % (family, k, v, list(ok[k]), _BRANCH_REF[k]))
    if elevation is not None and len(elevation) > 0:
        raise NotImplementedError("sub_g_update with an elevation MLP %r is not restated here; "
                                  "reference branch: %s" % (list(elevation), _BRANCH_REF["elevation"]))
    if cntxt_mlp is not None and (family != "classification" or len(cntxt_mlp) > 0):
        raise NotImplementedError("sub_g_update with cntxt_mlp=%r is not restated here; reference "
                                  "branch: %s" % (cntxt_mlp, _BRANCH_REF["cntxt_mlp"]))


class SubGUpdate(nn.Module):
    """sub_g_update for aggtype='gcn', pool 'max_pooling', attfdim=10 (the shipped seg configs).

    in_feats   : C of the gathered features (0 -> first layer, neighbours carry xyz,w only)
    localfdim  : 0 or 3 (3: geo_vec is concatenated to the features, gcn_module_g_att.py:249-250)
    pt_mlp     : per-edge feature MLP dims
    center_in  : channels of center_ori_feats (None: down layer)
    center_dim / out_dim : centre MLP and update MLP dims (up layers)
    """

    def __init__(self, in_feats, pt_mlp, localfdim=0, relu=True, center_in=None, center_dim=(),
                 out_dim=(), bn_decay=0.9, attfdim=10, aggtype="gcn", pool_type="max_pooling",
                 att_full="", elevation=(), up_center_inte="concat", cntxt_mlp=None):
        super().__init__()
        check_shipped_branches("segmentation", attfdim=attfdim, localfdim=localfdim,
                               aggtype=aggtype, pool_type=pool_type, att_full=att_full,
                               elevation=elevation, up_center_inte=up_center_inte,
                               cntxt_mlp=cntxt_mlp)
        self.has_feats = in_feats > 0
        self.localfdim = localfdim
        self.relu = relu
        cin = 3 if not self.has_feats else in_feats + (3 if localfdim != 0 else 0)
        self.cin = cin
        C = pt_mlp[-1]
        self.pt_mlp = mlp(cin, pt_mlp, bn_decay)
        self.att1 = mlp(10, [C // 4], bn_decay)          # update_att_mlp2d_frst (:141)
        self.att2 = mlp(C // 4, [C], bn_decay)           # update_att_mlp2d_scnd (:152)
        self.center_mlp = mlp(center_in, list(center_dim), bn_decay) if center_in else None
        agg_c = C + (center_dim[-1] if (center_in and len(center_dim)) else (center_in or 0))
        self.update_mlp = mlp(agg_c, list(out_dim), bn_decay) if len(out_dim) else None
        self.out_channels = out_dim[-1] if len(out_dim) else agg_c

    mfma_train = True   # training on the GPU: MLPs through csrc/gridgcn_train.hip

    def edge_inputs(self, neighbors, centers_xyz):
        geo_vec, _, att_vec = geo_features(neighbors, centers_xyz)
        if not self.has_feats:
            nf = geo_vec                                                   # :242-243
        elif self.localfdim != 0:
            nf = torch.cat([geo_vec, neighbors[..., 4:]], dim=-1)          # :249-250
        else:
            nf = neighbors[..., 4:]
        return nf, att_vec

    def forward(self, centers_xyz, neighbors, center_masks=None, center_ori_feats=None, tail=None):
        """centers_xyz [B,O,3], neighbors [B,O,P,4+C] (already gathered), center_masks [B,O]|None,
        center_ori_feats [B,O,Cc]|None  ->  [B,O,out_channels]."""
        nf, att_vec = self.edge_inputs(neighbors, centers_xyz)
        pair = self.att2(self.att1(att_vec)) * self.pt_mlp(nf)             # :135-167
        agg = pair.max(dim=2).values                                       # :57-59 (unmasked, F10)
        return self.finish(agg, center_masks, center_ori_feats, tail=tail)

    def forward_src(self, cent, src, nebidx, center_masks=None, center_ori_feats=None, tail=None,
                    defer_mask=False):
        """Training path on the GPU: the edge inputs (gather + geo features + concat) come from
        one HIP kernel (ops.edge_inputs, scatter-add backward); the MLPs with batch-statistics
        BatchNorm are stock PyTorch ops.  defer_mask: the caller multiplies by center_masks itself
        (tcommon.cat_mask: the mask and the concat with the centres in one launch)."""
        from . import ops
        if defer_mask:
            center_masks = None
        att_layers, pt_layers = [self.att1[0], self.att2[0]], list(self.pt_mlp)
        src = src.contiguous()
        if self.mfma_train and self.training and torch.is_grad_enabled():
            from .train import common as tcommon, edge as tedge
            if tedge.edge_block_src_supported(pt_layers, att_layers, src, self.has_feats,
                                                  nebidx.shape[2]):
                # first conv on the source points, gathered afterwards: no [E, 3+Cf] tensor at all
                buf, out = None, None
                if center_ori_feats is not None and self.center_mlp is not None and \
                        tcommon.supported(list(self.center_mlp), src):
                    # update_func concatenates (centre MLP output, aggregate): both producers
                    # write their half of one buffer instead of a concat pass
                    B, O = nebidx.shape[0], nebidx.shape[1]
                    ccf = self.center_mlp[-1].lin.out_features
                    C = pt_layers[-1].lin.out_features
                    buf = torch.empty((B * O, ccf + C), dtype=torch.float32, device=src.device)
                    out = tcommon.alias_columns(buf, ccf, C)
                agg = tedge.edge_block_src_train(src, nebidx, cent.contiguous(), pt_layers,
                                                     att_layers, self.localfdim, out=out)
                return self.finish(agg, center_masks, center_ori_feats, buf=buf, tail=tail)
            if tedge.edge_block_supported(pt_layers, att_layers, src, nebidx.shape[2]) and \
                    ops.edge_inputs_rows_supported(src, self.has_feats):
                # rows laid out for the MFMA kernels (features | geo_vec | zero padding)
                nf, att16, rot = ops.edge_inputs_rows(src, nebidx, cent.contiguous(),
                                                      has_feats=self.has_feats,
                                                      localfdim=self.localfdim)
                agg = tedge.edge_block_train(nf, att16, pt_layers, att_layers, rot)
                return self.finish(agg, center_masks, center_ori_feats, tail=tail)
        nf, att_vec = ops.edge_inputs(src, nebidx, cent.contiguous(),
                                      has_feats=self.has_feats, localfdim=self.localfdim)
        if self.mfma_train and self.training and torch.is_grad_enabled():
            from .train import common as tcommon, edge as tedge
            if tedge.edge_block_supported(pt_layers, att_layers, nf, nebidx.shape[2]):
                agg = tedge.edge_block_train(nf, att_vec, pt_layers, att_layers)
                return self.finish(agg, center_masks, center_ori_feats, tail=tail)
        pair = run_mlp(att_layers, att_vec, self.mfma_train) * \
            run_mlp(pt_layers, nf, self.mfma_train)
        agg = pair.max(dim=2).values
        return self.finish(agg, center_masks, center_ori_feats, tail=tail)

    def packed_layers(self):
        """BatchNorm-folded, padded weights for the fused kernel (cached; eval mode only)."""
        from . import ops
        from .train import common as tcommon
        # (the parameter generation: fused optimizers rewrite weights without moving Tensor._version)
        key = (tcommon._PARAM_GEN[0],) + tuple(p._version for p in self.parameters()) + tuple(
            b._version for b in self.buffers())
        if getattr(self, "_packed_key", None) != key:
            # LDS row of the first pt layer = the gathered source row (x,y,z,w,features) with
            # columns 0..3 overwritten by (geo_vec, 0): map the layer's input channels onto it
            nfeat = self.cin - (3 if (not self.has_feats or self.localfdim != 0) else 0)
            if not self.has_feats:
                rows = [0, 1, 2]
            elif self.localfdim != 0:
                rows = [0, 1, 2] + [4 + j for j in range(nfeat)]
            else:
                rows = [4 + j for j in range(nfeat)]
            k0 = (4 + (nfeat if self.has_feats else 0) + 3) & ~3
            pt = []
            for i, l in enumerate(self.pt_mlp):
                w, b = l.folded()
                pt.append(ops.pack_conv_layer(w, b, rows, k0) if i == 0
                          else ops.pack_conv_layer(w, b))
            self._packed = (pt, [ops.pack_conv_layer(*self.att1[0].folded(), K=12),
                                 ops.pack_conv_layer(*self.att2[0].folded())])
            self._packed_key = key
        return self._packed

    def forward_fused(self, cent, src, nebidx, center_masks=None, center_ori_feats=None, tail=None):
        """Inference path through the hand-written gfx950 kernel (csrc/gridgcn_conv.hip):
        src [B,Nsrc,4+C] (NOT gathered), nebidx [B,O,P], cent [B,O,>=3]."""
        from . import ops
        assert not self.training, "the fused kernel folds BatchNorm: eval() mode only"
        from .train import common as tcommon, evalpath as teval
        att_layers, pt_layers = [self.att1[0], self.att2[0]], list(self.pt_mlp)
        if teval.edge_block_src_eval_supported(pt_layers, att_layers, src, self.has_feats):
            # up layers (one point conv): that conv on the source points, gathered by the max kernel
            if center_ori_feats is not None and self.center_mlp is not None and \
                    tcommon.supported(list(self.center_mlp), src) and \
                    all(l.lin.in_features <= 1024 for l in self.center_mlp):
                # update_func's concat: centre MLP and aggregate write the two halves of one buffer
                B, O = nebidx.shape[0], nebidx.shape[1]
                ccf = self.center_mlp[-1].lin.out_features
                C = pt_layers[-1].lin.out_features
                buf = torch.empty((B * O, ccf + C), dtype=torch.float32, device=src.device)
                teval.edge_block_src_eval(src, nebidx, cent.contiguous(), pt_layers[0],
                                              att_layers, self.localfdim,
                                              out=tcommon.alias_columns(buf, ccf, C))
                teval.mlp_bn_relu_eval(center_ori_feats, list(self.center_mlp),
                                           out=tcommon.alias_columns(buf, 0, ccf))
                return self.finish(buf.view(B, O, ccf + C), center_masks, None, tail=tail)
            agg = teval.edge_block_src_eval(src, nebidx, cent.contiguous(), pt_layers[0],
                                                att_layers, self.localfdim)
            return self.finish(agg, center_masks, center_ori_feats, tail=tail)
        pt, att = self.packed_layers()
        agg = ops.gridconv_forward(src.contiguous(), nebidx, cent.contiguous(), pt, att,
                                   has_feats=self.has_feats, localfdim=self.localfdim)
        return self.finish(agg, center_masks, center_ori_feats, tail=tail)

    def finish(self, agg, center_masks, center_ori_feats, buf=None, tail=None):
        link = None
        nonneg = False   # both halves of the concat are >= 0: update_func's ReLU is the identity
        if center_ori_feats is not None and buf is not None:
            # `agg` already sits in the right half of buf; the centre MLP writes the left half
            from .train import common as tcommon, evalpath as teval, head as thead, mlp as tmlp
            B, O = center_ori_feats.shape[0], center_ori_feats.shape[1]
            ccf = self.center_mlp[-1].lin.out_features
            nonneg = True
            if self._raw_link_ok(center_ori_feats, buf, tail, center_masks):
                # ... as the RAW output of its last conv: the update MLP applies that layer's
                # BatchNorm+ReLU while it loads the concat, and its input-gradient kernel accumulates
                # the layer's BatchNorm-backward sums (tcommon.RawLink): one activation pass and one
                # reduce pass over [B*O, ccf] less
                link = tcommon.RawLink(ccf, buf.shape[1], buf.device)
            cf = tmlp.mlp_bn_relu_train(center_ori_feats, list(self.center_mlp),
                                             out=tcommon.alias_columns(buf, 0, ccf), link=link)
            agg = tcommon._Cat2.apply(cf, agg, buf).reshape(B, O, buf.shape[1])
        elif center_ori_feats is not None:
            cf = (run_mlp(list(self.center_mlp), center_ori_feats, self.mfma_train)
                  if self.center_mlp is not None else center_ori_feats)
            agg = torch.cat([cf, agg], dim=-1)                             # up_center_inte=concat
        if self.relu and not nonneg:
            agg = F.relu(agg)                                              # update_func :31-32
        if self.update_mlp is not None:
            layers = list(self.update_mlp)
            if tail is not None and tail.layers and center_masks is None:
                # conv+BN+ReLU layers that directly follow this block (the head's fc1 after the last
                # up layer) join the same chain: one activation pass and one reduce pass less
                layers += list(tail.layers)
                tail.done = 1
                if tail.head is not None and self.mfma_train and agg.is_cuda and \
                        self.training and torch.is_grad_enabled():
                    from .train import common as tcommon, evalpath as teval, head as thead, mlp as tmlp
                    p, lin = tail.head[:2]
                    if thead.head_supported(agg, layers, lin):
                        # ... and the Dropout + class-score Linear behind them (thead._HeadTrain)
                        tail.done = 2
                        seed_dev = tail.head[2] if len(tail.head) > 2 else None
                        seed = tail.head[3] if len(tail.head) > 3 else None
                        return thead.head_train(agg, layers, p, lin, seed=seed,
                                                    seed_dev=seed_dev, prev=link)
                if tail.head is not None and self.mfma_train and agg.is_cuda and \
                        not self.training and not torch.is_grad_enabled():
                    from .train import common as tcommon, evalpath as teval, head as thead, mlp as tmlp
                    p, lin = tail.head[:2]
                    if thead.head_supported(agg, layers, lin) and \
                            all(l.lin.in_features <= 1024 for l in layers):
                        tail.done = 2                   # evaluation: dropout is the identity
                        return teval.head_eval(agg, layers, lin)
            if link is not None:
                from .train import common as tcommon, evalpath as teval, head as thead, mlp as tmlp
                agg = tmlp.mlp_bn_relu_train(agg, layers, prev=link)
            else:
                agg = run_mlp(layers, agg, self.mfma_train)
        if center_masks is not None:
            agg = agg * center_masks[..., None]                            # :284-285
        return agg

    raw_link = True   # training: centre MLP output handed to the update MLP raw (tcommon.RawLink)

    def _raw_link_ok(self, center_ori_feats, buf, tail, center_masks):
        """the update MLP (with the tail layers it absorbs) and the centre MLP's last layer are all
        shapes of the register-direct kernels, which alone read a row-strided Z"""
        from .train import common as tcommon
        if not (self.raw_link and self.mfma_train and self.training and torch.is_grad_enabled()
                and self.update_mlp is not None and buf.is_cuda):
            return False
        layers = list(self.update_mlp)
        if tail is not None and tail.layers and center_masks is None:
            layers += list(tail.layers)
        cm = list(self.center_mlp)
        C = cm[-1].lin.out_features
        cin_last = cm[-2].lin.out_features if len(cm) > 1 else center_ori_feats.shape[-1]
        cin_last = (cin_last + 3) & ~3
        need_dx = len(cm) > 1 or center_ori_feats.requires_grad
        return (tcommon.supported(layers, buf) and buf.shape[1] % 8 == 0 and buf.shape[1] <= 256
                and tcommon._dw_direct_ok(C, cin_last) and C % 8 == 0
                and (not need_dx or cin_last <= 256))
