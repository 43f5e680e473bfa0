"""Callers of the index operators: the ScanNet segmentation network `get_symbol_seg_ggcn`
(segmentation/models/ggcn_models_g.py:110-237), restated so that the HIP operators are exercised
with the reference's dataflow (layer chaining data_loc <- cent, actual_centnum threaded through,
data_layer = concat(cent, feats), up path BallKNN | GridifyUp -> gather -> sub_g_update).

Shape contract = segmentation/configs/configs.yaml:71-111 (8192-pt) and :144-189 (81920-pt).
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, synth
from .gridconv import ConvBNReLU, SubGUpdate, run_mlp

# last linear layer + softmax cross-entropy on the hand-written kernels (GPU, float32)
HEAD_KERNELS = True
# fc1 -> dropout -> fc2 as one op with the dropout folded into the neighbouring kernels
FUSED_HEAD = True

SEG_8192 = dict(
    grid=synth.SEG_SCANNET_8192, inputDim=[0, 64, 128], pt_ele_dim=[[32, 32, 64], [64, 64, 128],
                                                                   [128, 128, 256]],
    localfdim=0, relu=True, up_inputDim=[256, 128, 128], up_center_dim=[[128]] * 3,
    up_pt_ele_dim=[[128]] * 3, up_gcn_outDim=[[128]] * 3, up_neigh_fetch=True, num_classes=21,
    bn_decay=0.9, dropout=0.5)
SEG_81920 = dict(SEG_8192, grid=synth.SEG_SCANNET_81920, localfdim=3, relu=False)


class HipIndexOps:
    """Default index provider: the gfx950 kernels behind the C ABI (grid_gcn_amd.ops)."""
    Gridify = staticmethod(ops.Gridify)
    GridifyUp = staticmethod(ops.GridifyUp)
    BallKNN = staticmethod(ops.BallKNN)
    batch_take_g = staticmethod(ops.batch_take_g)


class HipIndexOpsKNN(HipIndexOps):
    """Same, with the centres' neighbours chosen as the exact top-P by distance to the voxel centre
    (mx.sym.GridifyKNN, gridifyknn.cu; built by the reference's Makefile but used by none of its
    shipped configs)."""
    Gridify = staticmethod(ops.GridifyKNN)


def _is_hip(ix):
    return isinstance(ix, type) and issubclass(ix, HipIndexOps)


def call_seed(seed, forward_no, call_no):
    """Seed of one index-operator call.  The reference reseeds cuRAND from gettimeofday().tv_usec
    at EVERY operator call (gridify.cu:377-379), so the random voxel sampling and the neighbour
    reservoirs are redrawn every iteration.  Here the draw is a splitmix64 step of (model seed,
    number of the training forward, number of the call inside it): fresh every call, reproducible
    from the model seed, identical for two models fed the same sequence (the parity tests)."""
    x = (int(seed) + 0x9E3779B97F4A7C15 * (forward_no * 64 + call_no + 1)) & (2 ** 64 - 1)
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
    return (x ^ (x >> 31)) & (2 ** 63 - 1)


_SIDE_STREAMS = {}      # device -> the stream of the index operators (OPT.INDEX_SIDE_STREAM), one per process


class _IndexAhead:
    """OPT.INDEX_SIDE_STREAM (round 6; OFF until a GPU session has run tests/test_zz_r6_unverified.py): every index
    operator of a step -- the Gridify of each down layer, the BallKNN / GridifyUp of each up layer -- depends on point
    COORDINATES only (ggcn_models_g.py:154-159, :204-210: `data_loc <- cent`), never on features.  They are ~0.2 ms of
    latency-bound launches on a handful of workgroups each; here everything behind the first Gridify runs on a side
    stream, in the shadow of the first layer's GridConv kernels, and the main stream waits on an event per result
    where the reference's graph consumes it.  Inside a captured step the fork and the joins are graph edges.
    Memory: the results are allocated on the side stream and read on the main one; they stay referenced by the
    autograd graph until the backward is through, and the next step's fork (side waits for main) orders any reuse
    of their blocks behind every reader."""

    def __init__(self, net, data, num, fwd_no, sd):
        cfg, g, ix = net.cfg, net.cfg["grid"], net.ix
        dev = data.device
        main = torch.cuda.current_stream(dev)
        side = _SIDE_STREAMS.get(dev)
        if side is None:
            side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
        self.main, self.down_res, self.up_res = main, [], []
        kw = dict(synth.gridify_kwargs(g, 0, net._seed(fwd_no, 0)), **sd)
        first = ix.Gridify(data.detach().contiguous(), num, **kw)          # (needed at once: on the main stream)
        self.down_res.append((first, None))
        side.wait_stream(main)
        locs, nums = [data, first[2]], [num, first[4]]
        with torch.cuda.stream(side):
            for i in range(1, len(net.down)):
                kw = dict(synth.gridify_kwargs(g, i, net._seed(fwd_no, i)), **sd)
                r = ix.Gridify(locs[-1].detach().contiguous(), nums[-1], **kw)
                ev = torch.cuda.Event()
                ev.record(side)
                self.down_res.append((r, ev))
                locs.append(r[2]); nums.append(r[4])
            for i in range(len(net.up)):
                down, upl = locs[-i - 1], locs[-i - 2]
                downnum, upnum = nums[-i - 1], nums[-i - 2]
                U = g["up"][i]
                if cfg["up_neigh_fetch"]:
                    radius = U["voxel_size"][0] * U["kernel_size"] * 1.7 / 2
                    nb = ix.BallKNN(upl[..., 0:3].detach(), down[..., 0:3].detach(), downnum, upnum,
                                    k=U["max_p_grid"], radius=radius)
                else:
                    nb, _ = ix.GridifyUp(down.detach().contiguous(), upl.detach().contiguous(), downnum, upnum,
                                         **synth.gridify_up_kwargs(g, i, net._seed(fwd_no, 16 + i)), **sd)
                ev = torch.cuda.Event()
                ev.record(side)
                self.up_res.append((nb, ev))

    def down(self, i):
        r, ev = self.down_res[i]
        if ev is not None:
            self.main.wait_event(ev)
        return r

    def up(self, i):
        nb, ev = self.up_res[i]
        self.main.wait_event(ev)
        return nb


class GGCNSeg(nn.Module):
    def __init__(self, cfg=SEG_8192, index_ops=HipIndexOps, seed=0, fixed_seed=False):
        """seed: base of the per-call sampling seeds (training mode redraws the random voxel
        sampling at every call, see call_seed).  fixed_seed=True, or eval mode: every call uses
        `seed` itself -- the reference's fast_apprxmt build (seed 0)."""
        super().__init__()
        self.cfg = cfg
        self.ix = index_ops
        self.seed = seed
        self.fixed_seed = fixed_seed
        self.forward_no = 0
        self.register_forward_hook(release_packs)
        # int64 GPU scalar added to every sampling / dropout seed inside the kernels (None: off).
        # graph.GraphedTrainStep sets and bumps it so that replays of ONE captured step still redraw
        self.seed_dev = None
        g = cfg["grid"]
        nd = len(g["down"])
        self.down = nn.ModuleList()
        feat_c = [0]                                   # channels (without the 4 loc dims) per level
        for i in range(nd):
            layer = SubGUpdate(cfg["inputDim"][i], cfg["pt_ele_dim"][i], cfg["localfdim"],
                               cfg["relu"], bn_decay=cfg["bn_decay"])
            self.down.append(layer)
            feat_c.append(layer.out_channels)
        self.up = nn.ModuleList()
        last_c = feat_c[-1]
        for i in range(len(g["up"])):
            this_c = 4 + feat_c[nd - 1 - i]            # f_this_layer carries xyz,w too (:212)
            layer = SubGUpdate(last_c, cfg["up_pt_ele_dim"][i], cfg["localfdim"], cfg["relu"],
                               center_in=this_c, center_dim=cfg["up_center_dim"][i],
                               out_dim=cfg["up_gcn_outDim"][i], bn_decay=cfg["bn_decay"])
            self.up.append(layer)
            last_c = layer.out_channels
        # get_seg_head (:30-43)
        self.fc1 = ConvBNReLU(last_c, 128, cfg["bn_decay"])
        # fc1 directly follows the last up layer's update MLP: one conv+BN+ReLU chain (see finish())
        self.fc2 = nn.Linear(128, cfg["num_classes"])
        nn.init.xavier_uniform_(self.fc2.weight)
        nn.init.zeros_(self.fc2.bias)
        # ... and so do fc1/dropout + fc2 (:36-38): train/head.py: _HeadTrain
        self.fused_head = HEAD_KERNELS and FUSED_HEAD and _is_hip(index_ops)
        # indices from the index operators lie in [-1, N-1]: sorted segmented-sum gather backward
        self._take_kw = dict(neighbour_index=True) if _is_hip(index_ops) else {}

    fused = True   # eval mode: run GridConv through csrc/gridgcn_conv.hip (BatchNorm folded)
    jobs = None    # set to a list to record (name, layer, cent, src, nebidx) of every fused call
    edge_kernel = True  # training: edge inputs from one HIP kernel instead of take+slice+cat ops
    glue_kernels = True  # training: concat + mask + padding of a layer boundary in one launch

    def use_fused(self):
        # the fused evaluation kernels return tensors without a grad_fn: eval() WITH grad enabled
        # (frozen-BatchNorm fine-tuning, saliency) must take the differentiable path
        return (self.fused and (not self.training) and (not torch.is_grad_enabled())
                and _is_hip(self.ix))

    def _seed(self, forward_no, call_no):
        if self.fixed_seed or not self.training:
            return self.seed
        return call_seed(self.seed, forward_no, call_no)

    def _seed_dev(self):
        """The device-side seed increment (graph.GraphedTrainStep) only applies where a fresh draw
        per call is the contract: training mode without fixed_seed.  Evaluation and fixed_seed nets
        sample with `seed` itself whatever was replayed before."""
        return self.seed_dev if (self.training and not self.fixed_seed) else None

    def forward(self, data_xyz, actual_centnum):
        """data_xyz [B,N,3] f32, actual_centnum [B,1] i32 -> logits [B,N,num_classes]."""
        cfg, g, ix = self.cfg, self.cfg["grid"], self.ix
        B, N, _ = data_xyz.shape
        nd = len(self.down)
        # training on the HIP path: concat / centre mask / zero padding of the layer boundaries in one
        # launch each (train/common.py: cat_mask) instead of 2-4 framework ops
        glue = (self.glue_kernels and _is_hip(ix) and self.edge_kernel and data_xyz.is_cuda
                and data_xyz.dtype == torch.float32 and self.training and torch.is_grad_enabled())
        if glue:
            from .train import common as tcommon, head as thead
            from .train.options import OPT
            glue = OPT.GLUE_KERNELS
        if glue:
            data, data_pad = tcommon.cat_mask(data_xyz.detach(), None, None, pad=True)  # :137
        else:
            data = torch.cat([data_xyz, torch.ones_like(data_xyz[..., :1])], dim=2)     # :137
            data_pad = data
        locs = [data]                 # centers_reverse_lst: [B,n,4] per level
        feats = [data]                # center_locnfeat_alllayers: [B,n,4+C]
        feats_pad = [data_pad]        # (rows zero-padded to whole 32-byte pieces: the centre MLPs' input)
        masks, nums = [], [actual_centnum]
        data_loc, data_layer = data, data
        fwd_no = self.forward_no
        if self.training:
            self.forward_no += 1
            if data_xyz.is_cuda and torch.is_grad_enabled():
                from .train import common as tcommon, head as thead
                from .train.options import OPT
                tcommon.PACKS.prepack(self)   # all weight layouts of the step, one launch
        seed_dev = self._seed_dev()
        sd = dict(seed_dev=seed_dev) if (seed_dev is not None and _is_hip(ix)) else {}
        ahead = None
        if glue and OPT.INDEX_SIDE_STREAM and ix is HipIndexOps:
            ahead = _IndexAhead(self, data, actual_centnum, fwd_no, sd)
        for i, layer in enumerate(self.down):
            if ahead is not None:
                nebidx, nebidxmsk, cent, centmsk, centnum = ahead.down(i)
            else:
                kw = dict(synth.gridify_kwargs(g, i, self._seed(fwd_no, i)), **sd)
                nebidx, nebidxmsk, cent, centmsk, centnum = ix.Gridify(
                    data_loc.detach().contiguous(), nums[-1], **kw)                 # :154-159
            data_loc = cent
            if self.use_fused():
                if self.jobs is not None:
                    self.jobs.append(("down%d" % i, layer, cent, data_layer, nebidx))
                cf = layer.forward_fused(cent, data_layer, nebidx, centmsk)
            elif _is_hip(self.ix) and self.edge_kernel:
                cf = layer.forward_src(cent, data_layer, nebidx, centmsk, defer_mask=glue)
            else:
                neighbors = ix.batch_take_g(data_layer.contiguous(), nebidx, **self._take_kw)  # :172-173
                cf = layer(cent[..., 0:3], neighbors, centmsk)                      # :185
            if glue:
                data_layer, dl_pad = tcommon.cat_mask(cent, cf, centmsk, pad=i != nd - 1)   # :186
            else:
                data_layer = dl_pad = torch.cat([cent, cf], dim=2)                  # :186
            locs.append(cent); feats.append(data_layer); masks.append(centmsk); nums.append(centnum)
            feats_pad.append(dl_pad)
        f_last = feats[-1]
        nup = len(self.up)
        # what follows the last up layer (fc1, dropout, fc2): offered to that layer's chain per call
        from .gridconv import Tail
        # dropout seed of the fused head: with a device-side increment the host part is a constant of
        # the MODEL seed (ranks / model instances built with different seeds drop different masks)
        tail = Tail((self.fc1,), (cfg["dropout"], self.fc2, seed_dev,
                                  call_seed(self.seed, 0, 63) if seed_dev is not None else None)
                    if self.fused_head else None)
        for i, layer in enumerate(self.up):
            down, upl = locs[-i - 1], locs[-i - 2]
            downnum, upnum = nums[-i - 1], nums[-i - 2]
            U = g["up"][i]
            if ahead is not None:
                nebidx = ahead.up(i)
            elif cfg["up_neigh_fetch"]:
                radius = U["voxel_size"][0] * U["kernel_size"] * 1.7 / 2            # :204
                # (the HIP operator reads the xyz columns of the [B,n,4] rows in place)
                cont = (lambda t: t) if _is_hip(ix) else (lambda t: t.contiguous())
                nebidx = ix.BallKNN(cont(upl[..., 0:3].detach()), cont(down[..., 0:3].detach()),
                                    downnum, upnum, k=U["max_p_grid"], radius=radius)   # :85
            else:
                nebidx, _ = ix.GridifyUp(down.detach().contiguous(), upl.detach().contiguous(),
                                         downnum, upnum,
                                         **synth.gridify_up_kwargs(g, i, self._seed(fwd_no, 16 + i)),
                                         **sd)  # :206-210
            f_this = feats[-i - 2]
            if glue and layer.center_mlp is not None and layer.mfma_train and \
                    tcommon.supported(list(layer.center_mlp), f_this):
                f_this = feats_pad[-i - 2]        # (the MFMA kernels take the zero-padded rows as they are)
            cmask = masks[-i - 2] if i != nup - 1 else None                         # :224
            if self.use_fused():
                if self.jobs is not None:
                    self.jobs.append(("up%d" % i, layer, upl, f_last, nebidx))
                cf = layer.forward_fused(upl, f_last, nebidx, cmask, center_ori_feats=f_this,
                                         tail=tail if i == nup - 1 else None)
            elif _is_hip(self.ix) and self.edge_kernel:
                cf = layer.forward_src(upl, f_last, nebidx, cmask, center_ori_feats=f_this,
                                       tail=tail if i == nup - 1 else None, defer_mask=glue)
            else:
                neighbors = ix.batch_take_g(f_last.contiguous(), nebidx, **self._take_kw)  # :217-218
                cf = layer(upl[..., 0:3], neighbors, cmask, center_ori_feats=f_this,
                           tail=tail if i == nup - 1 else None)                       # :229
            if i != nup - 1:                      # (the last layer's features go to the head only)
                if glue:
                    f_last = tcommon.cat_mask(upl, cf, cmask)[0]                  # :231
                else:
                    f_last = torch.cat([upl, cf], dim=2)                            # :231
        self.last_tail_done = tail.done           # (introspection only: which head path ran)
        if tail.done == 2:                        # fc1, dropout and fc2 ran inside the last up layer
            return cf
        net = cf if tail.done else run_mlp([self.fc1], cf)
        net = F.dropout(net, self.cfg["dropout"], self.training)
        if HEAD_KERNELS and self.training and torch.is_grad_enabled() and _is_hip(self.ix):
            from .train import common as tcommon, head as thead
            from .train.options import OPT
            if thead.linear_plain_supported(net, self.fc2):
                return thead.linear_plain_train(net, self.fc2)
        return self.fc2(net)


def _mlp_macs(seq):
    return sum(l.lin.in_features * l.lin.out_features for l in (seq or []))


def seg_forward_flops(net, B, N):
    """Algorithmic flops of ONE forward of GGCNSeg on B clouds of N points:
    2 * rows * sum(Cin*Cout) over every 1x1 conv -- per edge (B*O*P rows) for the point / attention
    MLPs, per centre for the centre / update MLPs, per point for the head.  Element-wise work,
    BatchNorm and the max-pool are not counted.  A training step = 3x (backward = 2x forward).
    Returns (edge_flops, per_point_flops)."""
    g = net.cfg["grid"]
    edge = rows = 0.0
    for i, layer in enumerate(net.down):
        L = g["down"][i]
        e = B * L["max_o_grid"] * L["max_p_grid"]
        edge += 2.0 * e * (_mlp_macs(layer.pt_mlp) + _mlp_macs(layer.att1) + _mlp_macs(layer.att2))
        rows += 2.0 * B * L["max_o_grid"] * (_mlp_macs(layer.center_mlp) + _mlp_macs(layer.update_mlp))
    for i, layer in enumerate(net.up):
        U = g["up"][i]
        m = N if i == len(net.up) - 1 else U["max_o_grid"]
        e = B * m * U["max_p_grid"]
        edge += 2.0 * e * (_mlp_macs(layer.pt_mlp) + _mlp_macs(layer.att1) + _mlp_macs(layer.att2))
        rows += 2.0 * B * m * (_mlp_macs(layer.center_mlp) + _mlp_macs(layer.update_mlp))
    rows += 2.0 * B * N * (_mlp_macs([net.fc1]) + net.fc2.in_features * net.fc2.out_features)
    return edge, rows


def release_packs(module, _inputs, _output):
    """forward hook of the three nets: see train/common.py: _PackCache.release"""
    from .train import common as tcommon
    tcommon.PACKS.release(module)


class WeightedGradient(torch.autograd.Function):
    """custom_op/weighted_gradient.py:10-26 as stock PyTorch ops: identity forward; backward
    multiplies every row of the gradient by max_c [grad_c < 0] * weight_c (class axis last)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(weight)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        f = ((g < 0).to(g.dtype) * w).amax(dim=-1, keepdim=True)
        return f * g, None


def seg_loss(logits, label, weights=None):
    """SoftmaxOutput(use_ignore=True, ignore_label=0, normalization='valid')
    (segmentation/models/ggcn_models_g.py:41): mean cross-entropy over labels != 0.
    weights: optional per-class gradient weights (the 'weighted_gradient' op of :39-40)."""
    lg, lb = logits.reshape(-1, logits.shape[-1]), label.reshape(-1).long()
    if lg.is_cuda and lg.dtype == torch.float32 and lg.shape[1] <= 32 and HEAD_KERNELS:
        from .train import head as thead
        return thead.softmax_ce(lg, lb, 0, weights)       # csrc/gridgcn_head.hip
    if weights is not None:
        lg = WeightedGradient.apply(lg, torch.as_tensor(weights, dtype=lg.dtype, device=lg.device))
    return F.cross_entropy(lg, lb, ignore_index=0, reduction="mean")


def bn_decay_at(step, bn_decay=0.9, factor=0.5, clip=0.99):
    """BaseSolver.reset_bn_decay (segmentation/train_test/base_solver.py:71-74): the BatchNorm
    momentum after `step` decays, min(1 - bn_decay * factor**step, clip)."""
    return min(1.0 - bn_decay * (factor ** step), clip)


def set_bn_decay(net, bn_decay):
    """Give every BatchNorm of `net` the MXNet momentum `bn_decay` (running = bn_decay * running +
    (1 - bn_decay) * batch); the kernels read the module's momentum at every call."""
    for m in net.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.momentum = 1.0 - bn_decay
