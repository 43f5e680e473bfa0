"""Is the training step host-bound?  Times K steps of the bench workload twice: the time the host
needs to ENQUEUE them (no sync inside) and the time until the GPU has finished them.
usage: python tools/host_vs_gpu.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import model, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = model.GGCNSeg(model.SEG_81920).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
data, npn = synth.make_batch(8, 81920, "planes")
x = torch.from_numpy(data[..., :3].copy()).to(dev)
n = torch.from_numpy(npn).to(dev)
lab = torch.randint(0, 21, (8, 81920), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.seg_loss(net(x, n), lab)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, until GPU done %.2f ms/step" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
# forward / backward split of the host time
torch.cuda.synchronize()
tf = tb = to = 0.0
for _ in range(K):
    torch.cuda.synchronize()
    a = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = model.seg_loss(net(x, n), lab)
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    opt.step()
    d = time.perf_counter()
    tf += b - a; tb += c - b; to += d - c
print("host: forward %.2f ms, backward %.2f ms, optimizer %.2f ms (GPU idle at start of each)" % (
    tf / K * 1e3, tb / K * 1e3, to / K * 1e3))
