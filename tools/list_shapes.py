"""Print the (E, cin, couts) of every conv+BN+ReLU chain of one training step of the bench
workload (which layer shapes the MFMA kernels see).  usage: python tools/list_shapes.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import model, synth  # noqa: E402
from grid_gcn_amd.train import common as tcommon

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
orig = tcommon._chain_forward


def logged(lib, x, params, bns, eps, rot=0, ndx0=0, **kw):
    L = len(params) // 4
    print("chain E=%d cin=%d couts=%s rot=%d ndx0=%d%s" % (
        x.shape[0], x.shape[1], [params[4 * l].shape[0] for l in range(L)], rot, ndx0,
        " (after a source-side first conv)" if kw.get("prev_bn") is not None else ""))
    return orig(lib, x, params, bns, eps, rot, ndx0, **kw)


tcommon._chain_forward = logged
dev = torch.device("cuda", 0)
net = model.GGCNSeg(model.SEG_81920).to(dev).train()
data, npn = synth.make_batch(B, 81920, "planes")
x = torch.from_numpy(data[..., :3].copy()).to(dev)
n = torch.from_numpy(npn).to(dev)
lab = torch.randint(0, 21, (B, 81920), device=dev)
loss = model.seg_loss(net(x, n), lab)
print("loss", float(loss))
for name, layer in list(zip(["down%d" % i for i in range(3)], net.down)) + \
        list(zip(["up%d" % i for i in range(3)], net.up)):
    pass
