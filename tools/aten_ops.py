"""Which stock PyTorch kernels are still in the training step?  Runs the bench workload under
torch.profiler and lists the non-gg device kernels grouped by (aten op, input shapes).
usage: python tools/aten_ops.py [steps]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import model, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = model.GGCNSeg(model.SEG_81920).to(dev).train()
from grid_gcn_amd import optim  # noqa: E402
opt = optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(K):
        step()
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_input_shape=True):
    dt = getattr(ev, "self_device_time_total", 0.0)
    if dt <= 0 or ev.key.startswith("gg_k") or "gg_k_" in ev.key:
        continue
    agg[(ev.key, str(ev.input_shapes)[:90])][0] += ev.count
    agg[(ev.key, str(ev.input_shapes)[:90])][1] += dt
tot = 0.0
for (k, shp), (cnt, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.1f us/step n=%5.1f %-40s %s" % (dt / K, cnt / K, k[:40], shp))
    tot += dt
print("listed total %.1f us/step" % (tot / K))
# the small stock ops by name: each costs ~4 us of GPU time however little it does
byop = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages():
    dt = getattr(ev, "self_device_time_total", 0.0)
    if dt > 0 and ev.key.startswith("aten::"):
        byop[ev.key][0] += ev.count
        byop[ev.key][1] += dt
print("---- aten ops with device time")
n = t = 0
for k, (cnt, dt) in sorted(byop.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us/step n=%5.1f %s" % (dt / K, cnt / K, k))
    n += cnt; t += dt
print("aten total: %.1f launches, %.1f us per step" % (n / K, t / K))

# (where each of them is issued: tools/aten_sites.py)
