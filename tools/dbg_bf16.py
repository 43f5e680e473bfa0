import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import train_ops
from grid_gcn_amd.gridconv import mlp
DEV = "cuda:0"
torch.manual_seed(1)
for cin, dims in ((136, [128]), (128, [128]), (128, [256]), (256, [128]), (64, [64]), (32, [128]), (136, [128, 128, 256]), (8, [32]), (24, [64])):
    layers = mlp(cin, dims).to(DEV).train()
    x = torch.randn(20000, cin, device=DEV)
    out = {}
    for mode in ("fp32", "bf16"):
        train_ops.set_mlp_precision(mode)
        for l in layers: l.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = train_ops.mlp_bn_relu_train(xi, list(layers))
        (y * y).sum().backward()
        out[mode] = (y.detach(), xi.grad, layers[0].lin.weight.grad.clone())
    train_ops.set_mlp_precision("fp32")
    def rel(a, b): return float((a - b).abs().max() / b.abs().max())
    print(cin, dims, "y %.3g gx %.3g gw %.3g" % tuple(rel(out["bf16"][i], out["fp32"][i]) for i in range(3)),
          "nan:", [bool(torch.isnan(t).any()) for t in out["bf16"]])

# whole segmentation network, fp32 vs bf16 contraction mode (same weights, same sampling seeds)
import copy
from grid_gcn_amd import model, synth
torch.manual_seed(3)
cfg = dict(model.SEG_81920, dropout=0.0)
net = model.GGCNSeg(cfg, fixed_seed=True).to(DEV).train()
state = copy.deepcopy(net.state_dict())
data, npn = synth.make_batch(2, 16384, "planes")
x = torch.from_numpy(data[..., :3].copy()).to(DEV); n = torch.from_numpy(npn).to(DEV)
lab = torch.randint(0, 21, (2, 16384), device=DEV)
res = {}
for mode in ("fp32", "bf16"):
    train_ops.set_mlp_precision(mode)
    net.load_state_dict(state); net.zero_grad()
    loss = model.seg_loss(net(x, n), lab); loss.backward()
    res[mode] = (float(loss), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double())
train_ops.set_mlp_precision("fp32")
a, b = res["fp32"], res["bf16"]
print("model loss fp32 %.6f bf16 %.6f; grad rel L2 %.4f cos %.6f nan %s" % (
    a[0], b[0], float((a[1] - b[1]).norm() / a[1].norm()), float((a[1] * b[1]).sum() / (a[1].norm() * b[1].norm())),
    bool(torch.isnan(b[1]).any())))
