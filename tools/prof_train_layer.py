"""Time one training-mode MLP (fwd + bwd) on the MFMA kernels vs stock PyTorch.
usage: python tools/prof_train_layer.py [--E 3276800] [--cin 131] [--dims 128] [--iters 5] [--nograd-x]"""
import argparse
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd.train import mlp as tmlp  # noqa: E402
from grid_gcn_amd.gridconv import mlp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--E", type=int, default=3276800)
ap.add_argument("--cin", type=int, default=131)
ap.add_argument("--dims", default="128")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--nograd-x", action="store_true")
ap.add_argument("--only", default="both")
a = ap.parse_args()
dims = [int(d) for d in a.dims.split(",")]
dev = "cuda:0"
torch.manual_seed(0)
ref = mlp(a.cin, dims).to(dev).train()
new = copy.deepcopy(ref)
x = torch.randn(a.E, a.cin, device=dev, requires_grad=not a.nograd_x)
g = torch.randn(a.E, dims[-1], device=dev)
macs = sum(l.lin.in_features * l.lin.out_features for l in ref)
flops_f = 2.0 * a.E * macs


def run(fn):
    def step():
        y = fn(x)
        y.backward(g)
        return y
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ef0, ef1, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tf = tb = 0.0
    for _ in range(a.iters):
        ef0.record()
        y = fn(x)
        ef1.record()
        y.backward(g)
        eb1.record()
        torch.cuda.synchronize()
        tf += ef0.elapsed_time(ef1)
        tb += ef1.elapsed_time(eb1)
    return tf / a.iters, tb / a.iters


nb = 3.0 if not a.nograd_x else 2.0 + (len(dims) - 1) / len(dims)
if a.only in ("both", "mfma"):
    tf, tb = run(lambda t: tmlp.mlp_bn_relu_train(t, list(new)))
    print("mfma : fwd %7.3f ms (%5.1f TF/s)  bwd %7.3f ms (%5.1f TF/s)" % (
        tf, flops_f / tf / 1e9, tb, 2 * flops_f / tb / 1e9))
if a.only in ("both", "torch"):
    tf, tb = run(lambda t: ref(t))
    print("torch: fwd %7.3f ms (%5.1f TF/s)  bwd %7.3f ms (%5.1f TF/s)" % (
        tf, flops_f / tf / 1e9, tb, 2 * flops_f / tb / 1e9))
