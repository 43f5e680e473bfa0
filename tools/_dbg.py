import copy, sys, torch
sys.path.insert(0, ".")
from grid_gcn_amd import graph, model, synth, train_ops
DEV = "cuda:0"
torch.manual_seed(5)
cfg = model.SEG_8192
net_e = model.GGCNSeg(cfg, seed=11).to(DEV).train()
net_g = model.GGCNSeg(cfg, seed=11).to(DEV).train()
net_g.load_state_dict(copy.deepcopy(net_e.state_dict()))
data, npn = synth.make_batch(2, 8192, "planes", first_id=70)
x = torch.from_numpy(data[..., :3].copy()).to(DEV)
n = torch.from_numpy(npn).to(DEV)
lab = torch.randint(0, 21, (2, 8192), device=DEV)
mk = lambda net: torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
opt_e, opt_g = mk(net_e), mk(net_g)
W = 2
gs = graph.GraphedTrainStep(net_g, opt_g, model.seg_loss, (x, n), lab, warmup=W)
net_e.seed_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
def eager(f=None):
    if f is not None:
        net_e.forward_no = f
    net_e.seed_dev.add_(graph._GOLDEN)
    opt_e.zero_grad(set_to_none=True)
    loss = model.seg_loss(net_e(x, n), lab)
    loss.backward()
    opt_e.step()
    return float(loss)
for _ in range(W):
    eager()
def diff(tag):
    worst = []
    for (k, a), (_, b) in zip(net_e.named_parameters(), net_g.named_parameters()):
        worst.append((float((a - b).abs().max()), k))
    worst.sort(reverse=True)
    print(tag, worst[:6])
    gw = []
    for (k, a), (_, b) in zip(net_e.named_parameters(), net_g.named_parameters()):
        if a.grad is not None and b.grad is not None:
            gw.append((float((a.grad - b.grad).abs().max() / (a.grad.abs().max() + 1e-12)), k))
    gw.sort(reverse=True)
    print(tag, "grads", gw[:8])
diff("after warmup")
for i in range(2):
    le = eager(W); lg = float(gs())
    print("loss", le, lg)
    diff("step %d" % i)
