"""Time the fused GridConv kernel per layer of BASELINE configs[3] (ScanNet 81920-pt, B=8, eval
mode) against the stock-PyTorch path, and report algorithmic TFLOP/s (2*B*O*P*sum
(Cin*Cout) over the per-edge 1x1 convs; element-wise work not counted).
Usage: python tools/prof_gridconv.py [--B 8] [--iters 10] [--only fused|torch]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import model, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--points", type=int, default=81920)
ap.add_argument("--only", default="both")
a = ap.parse_args()
dev = "cuda:0"
cfg = model.SEG_81920 if a.points > 8192 else model.SEG_8192
torch.manual_seed(0)
net = model.GGCNSeg(cfg).to(dev).eval()
data, npn = synth.make_batch(a.B, a.points, "planes")
x = torch.from_numpy(data).to(dev)
n = torch.from_numpy(npn).to(dev)
g = cfg["grid"]


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


def edge_macs(layer):
    m = 0
    for seq in (layer.pt_mlp, layer.att1, layer.att2):
        for l in seq:
            m += l.lin.in_features * l.lin.out_features
    return m


with torch.no_grad():
    locs, feats, nums, masks = [x], [x], [n], []
    data_layer = x
    jobs = []
    for i, layer in enumerate(net.down):
        nebidx, _, cent, cmsk, cn = ops.Gridify(locs[-1], nums[-1], **synth.gridify_kwargs(g, i))
        jobs.append(("down%d" % i, layer, cent, data_layer, nebidx, cmsk, None))
        cf = layer.forward_fused(cent, data_layer, nebidx, cmsk)
        data_layer = torch.cat([cent, cf], 2)
        locs.append(cent); feats.append(data_layer); nums.append(cn); masks.append(cmsk)
    f_last = feats[-1]
    for i, layer in enumerate(net.up):
        down, upl = locs[-i - 1], locs[-i - 2]
        U = g["up"][i]
        r = U["voxel_size"][0] * U["kernel_size"] * 1.7 / 2
        nebidx = ops.BallKNN(upl[..., :3].contiguous(), down[..., :3].contiguous(), nums[-i - 1],
                             nums[-i - 2], k=5, radius=r)
        cmask = masks[-i - 2] if i != 2 else None
        jobs.append(("up%d" % i, layer, upl, f_last, nebidx, cmask, feats[-i - 2]))
        cf = layer.forward_fused(upl, f_last, nebidx, cmask, feats[-i - 2])
        f_last = torch.cat([upl, cf], 2)
    tot_f = tot_t = 0.0
    for name, layer, cent, src, nebidx, cmask, cori in jobs:
        B, O, P = nebidx.shape
        flops = 2.0 * B * O * P * edge_macs(layer)
        pt, att = layer.packed_layers()
        line = "%-6s O=%6d P=%3d cin=%3d C=%3d  %7.2f GFLOP" % (
            name, O, P, layer.cin, pt[-1][4], flops / 1e9)
        if a.only in ("both", "fused"):
            ms = timeit(lambda: ops.gridconv_forward(src.contiguous(), nebidx, cent, pt, att,
                                                     has_feats=layer.has_feats,
                                                     localfdim=layer.localfdim))
            tot_f += ms
            line += "  fused %8.3f ms %6.1f TFLOP/s (%4.1f%% of 157.3)" % (
                ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100)
        if a.only in ("both", "torch"):
            def torch_path():
                nb = ops.batch_take_g(src.contiguous(), nebidx)
                nf, att_vec = layer.edge_inputs(nb, cent[..., :3])
                return (layer.att2(layer.att1(att_vec)) * layer.pt_mlp(nf)).max(dim=2).values
            ms = timeit(torch_path)
            tot_t += ms
            line += "  torch %8.3f ms" % ms
        print(line)
    print("total fused %.3f ms   torch %.3f ms" % (tot_f, tot_t))
