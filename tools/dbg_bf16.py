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
