"""Whole-network fwd+bwd: HIP edge-kernel path vs stock-op path, per-parameter gradient differences,
both against the stock path in float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grid_gcn_amd import model, synth
DEV = "cuda:0"
torch.manual_seed(0)
net = model.GGCNSeg(model.SEG_81920, fixed_seed=True).to(DEV).train()
data, npn = synth.make_batch(2, 4096, "planes")
x = torch.from_numpy(data[..., :3].copy()).to(DEV)
n = torch.from_numpy(npn).to(DEV)
lab = torch.randint(0, 21, (2, 4096), device=DEV)
names = [k for k, _ in net.named_parameters()]
res = []
for ek in (True, False):
    net.zero_grad(); net.edge_kernel = ek
    torch.manual_seed(5)
    loss = model.seg_loss(net(x, n), lab); loss.backward()
    res.append((loss.item(), [p.grad.detach().double().cpu().clone() for p in net.parameters()]))
print("loss", res[0][0], res[1][0])
worst = []
for k, a, b in zip(names, res[0][1], res[1][1]):
    s = float(b.abs().max()) + 1e-30
    worst.append((float((a - b).abs().max()) / s, k, s))
worst.sort(reverse=True)
for w in [w for w in worst if w[2] > 1e-6][:14]:
    print("%.3e  %-50s scale %.3e" % w)
