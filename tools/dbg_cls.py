import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from grid_gcn_amd import model_cls, synth
torch.manual_seed(0)
cfg = dict(model_cls.CLS_MN40, dropout=0.0)
net = model_cls.GGCNCls(cfg).to("cuda:0").train()
data, npn = synth.make_batch(8, 1024, "ball")
x, n = torch.from_numpy(data[..., :3].copy()).to("cuda:0"), torch.from_numpy(npn).to("cuda:0")
lab = torch.randint(0, 40, (8,), device="cuda:0")
state = {k: v.clone() for k, v in net.state_dict().items()}
res = []
for mfma in (True, False):
    net.load_state_dict(state); net.zero_grad(set_to_none=True)
    for l in net.layers: l.mfma_train = mfma
    loss = model_cls.cls_loss(net(x, n), lab); loss.backward()
    res.append({k: p.grad.clone() for k, p in net.named_parameters()})
for k in res[0]:
    a, b = res[0][k], res[1][k]
    d = float((a - b).abs().max()); s = float(b.abs().max())
    if d > 2e-4 * max(s, 1e-6): print("%-40s diff %.3e scale %.3e shape %s" % (k, d, s, tuple(a.shape)))
print("---- stock path, input perturbed by 1e-7")
res = []
for eps in (0.0, 1e-7):
    net.load_state_dict(state); net.zero_grad(set_to_none=True)
    for l in net.layers: l.mfma_train = False
    loss = model_cls.cls_loss(net(x * (1 + eps), n), lab); loss.backward()
    res.append({k: p.grad.clone() for k, p in net.named_parameters()})
worst = 0
for k in res[0]:
    a, b = res[0][k], res[1][k]
    d = float((a - b).abs().max()); s = float(b.abs().max())
    if s > 1e-5: worst = max(worst, d / s)
    if d > 2e-3 * max(s, 1e-6) and s > 1e-5: print("%-40s diff %.3e scale %.3e" % (k, d, s))
print("worst relative diff", worst)
