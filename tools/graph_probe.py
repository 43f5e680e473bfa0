"""Which part of the training step survives capture + replay as a hipGraph (torch.cuda.CUDAGraph)?
usage: python tools/graph_probe.py <index|fwd|fwdbwd|step> [B] [N]   (run each mode in its own process)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import model, ops, synth  # noqa: E402

mode = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 81920
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = model.SEG_81920
net = model.GGCNSeg(cfg).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, fused=True, capturable=True)
data, npn = synth.make_batch(B, N, "planes")
x = torch.from_numpy(data[..., :3].copy()).to(dev)
d4 = torch.from_numpy(data).to(dev)
n = torch.from_numpy(npn).to(dev)
lab = torch.randint(0, 21, (B, N), device=dev)
kw = synth.gridify_kwargs(cfg["grid"], 0, 0)


def work():
    if mode == "index":
        return ops.Gridify(d4, n, **kw)[2].sum()
    if mode == "fwd":
        with torch.no_grad():
            return net(x, n).sum()
    opt.zero_grad(set_to_none=True)
    loss = model.seg_loss(net(x, n), lab)
    loss.backward()
    if mode == "step":
        opt.step()
    return loss.detach()


def timed(fn, K=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        work()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print(mode, "eager %.3f ms" % timed(work), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = work()
torch.cuda.synchronize()
print(mode, "captured", flush=True)
g.replay()
torch.cuda.synchronize()
print(mode, "replayed once, out %.5f" % float(out), flush=True)
print(mode, "graph %.3f ms" % timed(g.replay), flush=True)
