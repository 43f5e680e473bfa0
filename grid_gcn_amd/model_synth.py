"""BASELINE.json configs[4]: the builder-defined HBM stress workload.

Not in the reference: 200 000-point synthetic clouds, K = P = 64 neighbours, FOUR GridConv down
layers -- grids 64^3 / 32^3 / 16^3 / 8^3 over [0,2)^3, kernel 3, O = 16384 / 4096 / 1024 / 256,
channels 0 -> 64 -> 128 -> 256 -> 512 with the per-edge point MLP [C/2, C/2, C] and the
segmentation network's attention MLP (10 -> C/4 -> C).  The layers are the segmentation GridConv
(gridconv.SubGUpdate, gcn_module_g_att.py:172-287) chained as in get_symbol_seg_ggcn's down path
(ggcn_models_g.py:152-187); a global max over the last layer's centres and one linear layer give
class scores so that a training step has a loss.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import synth
from .gridconv import SubGUpdate
from .model import HipIndexOps, call_seed, _is_hip, _mlp_macs, release_packs

SYNTH_200K = dict(
    grid=synth.SYNTH_200K, inputDim=[0, 64, 128, 256],
    pt_ele_dim=[[32, 32, 64], [64, 64, 128], [128, 128, 256], [256, 256, 512]],
    localfdim=3, relu=False, num_classes=40, bn_decay=0.9)


class GGCNSynth(nn.Module):
    def __init__(self, cfg=SYNTH_200K, index_ops=HipIndexOps, seed=0, fixed_seed=False):
        super().__init__()
        self.cfg, self.ix, self.seed, self.fixed_seed = cfg, index_ops, seed, fixed_seed
        self.forward_no = 0
        self.register_forward_hook(release_packs)
        self.seed_dev = None      # see GGCNSeg
        self.down = nn.ModuleList(
            SubGUpdate(cfg["inputDim"][i], cfg["pt_ele_dim"][i], cfg["localfdim"], cfg["relu"],
                       bn_decay=cfg["bn_decay"]) for i in range(len(cfg["grid"]["down"])))
        self.fc = nn.Linear(self.down[-1].out_channels, cfg["num_classes"])
        nn.init.xavier_uniform_(self.fc.weight)
        nn.init.zeros_(self.fc.bias)
        self._take_kw = dict(neighbour_index=True) if _is_hip(index_ops) else {}

    edge_kernel = True

    def forward(self, data_xyz, actual_centnum, return_layers=False):
        """data_xyz [B,N,3] f32, actual_centnum [B,1] i32 -> class scores [B,num_classes]."""
        g, ix = self.cfg["grid"], self.ix
        data = torch.cat([data_xyz, torch.ones_like(data_xyz[..., :1])], dim=2)
        fwd_no = self.forward_no
        if self.training:
            self.forward_no += 1
            if data_xyz.is_cuda and torch.is_grad_enabled():
                from .train import common as tcommon, head as thead
                tcommon.PACKS.prepack(self)   # all weight layouts of the step, one launch
        data_loc, data_layer, num = data, data, actual_centnum
        outs = []
        for i, layer in enumerate(self.down):
            seed = self.seed if (self.fixed_seed or not self.training) else \
                call_seed(self.seed, fwd_no, i)
            seed_dev = self.seed_dev if (self.training and not self.fixed_seed) else None
            sd = dict(seed_dev=seed_dev) if (seed_dev is not None and _is_hip(ix)) else {}
            nebidx, _, cent, centmsk, num = ix.Gridify(
                data_loc.detach().contiguous(), num, **synth.gridify_kwargs(g, i, seed), **sd)
            data_loc = cent
            if _is_hip(ix) and self.edge_kernel:
                cf = layer.forward_src(cent, data_layer, nebidx, centmsk)
            else:
                nb = ix.batch_take_g(data_layer.contiguous(), nebidx, **self._take_kw)
                cf = layer(cent[..., 0:3], nb, centmsk)
            data_layer = torch.cat([cent, cf], dim=2)
            outs.append(cf)
        # masked global max over the centres (features of empty slots are zero, real ones may be
        # negative with relu=False: mask with -inf)
        neg = torch.finfo(cf.dtype).min
        pooled = torch.where(centmsk[..., None] > 0, cf, torch.full_like(cf, neg)).max(dim=1).values
        if pooled.is_cuda and _is_hip(ix):
            from .train import common as tcommon, head as thead
            logits = thead.linear_mm(pooled, self.fc)
        else:
            logits = self.fc(pooled)
        return (logits, outs) if return_layers else logits


def synth_loss(logits, label):
    return F.cross_entropy(logits, label.long(), reduction="mean")


def forward_flops(net, B):
    """Algorithmic flops of one forward (2 * edges * sum(Cin*Cout))."""
    g = net.cfg["grid"]
    fl = 0.0
    for i, layer in enumerate(net.down):
        L = g["down"][i]
        e = B * L["max_o_grid"] * L["max_p_grid"]
        fl += 2.0 * e * (_mlp_macs(layer.pt_mlp) + _mlp_macs(layer.att1) + _mlp_macs(layer.att2))
    return fl + 2.0 * B * net.fc.in_features * net.fc.out_features
