"""Evaluation forward of the bench workload a few times (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import model, synth  # noqa: E402

dev = "cuda:0"
net = model.GGCNSeg(model.SEG_81920).to(dev).eval()
data, npn = synth.make_batch(8, 81920, "planes")
x = torch.from_numpy(data[..., :3].copy()).to(dev)
n = torch.from_numpy(npn).to(dev)
with torch.no_grad():
    for _ in range(13):
        net(x, n)
torch.cuda.synchronize()
