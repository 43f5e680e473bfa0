"""Training step as a hipGraph (grid_gcn_amd/graph.py): eager vs replay time, fresh random draws
per replay, and replay after interleaved eager launches (the round-1 open issue).
usage: python tools/graph_step_probe.py [cfg4|cfg3] [--world-emul]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import graph, model, ops, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
B, N, cfg = (8, 81920, model.SEG_81920) if which == "cfg4" else (16, 8192, model.SEG_8192)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = model.GGCNSeg(cfg).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, fused=True, capturable=True)
data, npn = synth.make_batch(B, N, "planes")
x = torch.from_numpy(data[..., :3].copy()).to(dev)
n = torch.from_numpy(npn).to(dev)
lab = torch.randint(0, 21, (B, N), device=dev)


def eager():
    opt.zero_grad(set_to_none=True)
    loss = model.seg_loss(net(x, n), lab)
    loss.backward()
    opt.step()
    return loss


def timed(fn, K=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for _ in range(3):
    eager()
print(which, "eager  %.3f ms/step" % timed(eager), flush=True)
gs = graph.GraphedTrainStep(net, opt, model.seg_loss, (x, n), lab)
print(which, "captured", flush=True)
losses = []
for _ in range(5):
    losses.append(float(gs()))
print(which, "replay losses", ["%.5f" % l for l in losses], flush=True)
print(which, "graph  %.3f ms/step" % timed(gs), flush=True)
# the random draws move on: centres of layer 0 picked with the current device seed
d4 = torch.from_numpy(data).to(dev)
kw = synth.gridify_kwargs(cfg["grid"], 0, 0)
a = ops.Gridify(d4, n, seed_dev=net.seed_dev, **kw)[2].clone()
gs()
b = ops.Gridify(d4, n, seed_dev=net.seed_dev, **kw)[2].clone()
print(which, "centres differ between replays:", bool((a != b).any()), flush=True)
# interleave eager launches of the same model with replays (hung in round 1 after ~1000 launches)
net.eval()
for rnd in range(6):
    with torch.no_grad():
        for _ in range(10):
            net(x, n)
    net.train()
    l = float(gs())
    net.eval()
    print(which, "round %d: 10 eager forwards + replay ok, loss %.5f" % (rnd, l), flush=True)
net.train()
print(which, "graph after interleaving %.3f ms/step" % timed(gs), flush=True)
