"""Where does the training step still issue stock PyTorch ops?  One eager cfg4 step under a
TorchDispatchMode: every non-view aten op with its shapes and the innermost frames of this package
(forward and backward: the mode follows the autograd thread).
usage: python tools/aten_sites.py [cfg4|cfg3]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import model, optim, synth  # noqa: E402

VIEWS = ("view", "reshape", "slice", "select", "detach", "t.", "transpose", "permute", "expand", "as_strided",
         "alias", "unsqueeze", "squeeze", "_unsafe_view", "empty", "new_empty", "split", "unbind", "narrow",
         "_local_scalar_dense", "is_", "stride", "size", "numel", "dim", "_has_", "lift_fresh", "unfold",
         "record_stream", "set_", "resize_", "sym_")

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N = 81920 if cfg == "cfg4" else 8192
B = 8 if cfg == "cfg4" else 32
net = model.GGCNSeg(model.SEG_81920 if cfg == "cfg4" else model.SEG_8192).to(dev).train()
opt = optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)        # (what bench.py times)
data, npn = synth.make_batch(B, N, "planes")
x = torch.from_numpy(data[..., :3].copy()).to(dev)
n = torch.from_numpy(npn).to(dev)
lab = torch.randint(0, 21, (B, N), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.seg_loss(net(x, n), lab)
    loss.backward()
    opt.step()


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.acc = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        full = str(func).replace("aten.", "")
        if not any(full.startswith(v) for v in VIEWS):
            fr = [f for f in traceback.extract_stack() if "grid_gcn_amd/" in f.filename or "aten_sites" in f.filename]
            site = " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1]) or "?"
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:3]
            self.acc[(full, site, str(shp)[:70])] += 1
        return func(*args, **(kwargs or {}))


for _ in range(2):
    step()
torch.cuda.synchronize()
with Log() as lg:
    step()
torch.cuda.synchronize()
tot = 0
for (op, site, shp), c in sorted(lg.acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%3d %-28s %-70s %s" % (c, op, site, shp))
    tot += c
print("total non-view stock ops in one step: %d" % tot)
