"""Forward of the attention pair of an up layer at the cfg4 up2 shape (655 360 centres x 5 neighbours, 10 -> 32 -> 128):
the path that writes and reads Z2 [E, 128] (gridgcn_linear_fwd_direct_fin + gridgcn_pairmax_fwd_src) against the one
that does not (gridgcn_att_bn2_moments + gridgcn_att_pairmax_fwd, csrc/gridgcn_attfwd.hip).  Per piece, HIP events."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grid_gcn_amd import _lib
from grid_gcn_amd.train import common as tcommon
from grid_gcn_amd.ops import _ptr, _stream
from grid_gcn_amd.train.edge import _att_fwd_noz

lib = _lib.load()
dev = torch.device("cuda:0")
B, Nsrc, O, P, cin, C = 8, 20480, 81920, 5, 32, 128
if len(sys.argv) > 1:
    O = int(sys.argv[1])
ncent, E, R = B * O, B * O * P, B * Nsrc
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
Ysrc = rnd(R, C)
nebidx = torch.randint(0, Nsrc, (B, O, P), device=dev, dtype=torch.int32, generator=g)
# neighbours of a centre are spatially close in the real layer: sources near o * Nsrc / O
base = (torch.arange(O, device=dev) * Nsrc // O)[None, :, None]
nebidx = ((base + torch.randint(-40, 40, (B, O, P), device=dev, generator=g)) % Nsrc).int().contiguous()
att16 = rnd(E, 16)
Wg, b = rnd(3, C) * 0.3, rnd(C) * 0.1
W1, b1 = torch.nn.Parameter(rnd(cin, 16) * 0.3), torch.nn.Parameter(rnd(cin) * 0.1)
W2, b2 = torch.nn.Parameter(rnd(C, cin) * 0.2), torch.nn.Parameter(rnd(C) * 0.1)
bn1, bn2 = torch.nn.BatchNorm1d(cin).to(dev), torch.nn.BatchNorm1d(C).to(dev)
pa = [W1, b1, bn1.weight, bn1.bias, W2, b2, bn2.weight, bn2.bias]
scp, shp = rnd(C).abs() + 0.5, rnd(C) * 0.3
agg = torch.empty(ncent, C, device=dev)
amax = torch.empty(ncent, C, dtype=torch.uint8, device=dev)
zsel = torch.empty(2, ncent, C, device=dev)
st = _stream(att16)


def old():
    sa = tcommon._chain_forward(lib, att16, pa, [bn1, bn2], 1e-5)
    rc = lib.gridgcn_pairmax_fwd_src_z(_ptr(Ysrc), _ptr(nebidx), _ptr(att16), _ptr(Wg), _ptr(b), B, Nsrc, O,
                                       _ptr(sa.Z[-1]), 0, _ptr(scp), _ptr(shp), _ptr(sa.scale[-1]), _ptr(sa.shift[-1]),
                                       ncent, P, C, _ptr(agg), C, _ptr(amax), _ptr(zsel), st)
    assert rc == 0


def new():
    sa = _att_fwd_noz(lib, att16, pa, [bn1, bn2], 1e-5, st)
    rc = lib.gridgcn_att_pairmax_fwd(_ptr(Ysrc), _ptr(nebidx), _ptr(att16), _ptr(Wg), _ptr(b), B, Nsrc, O,
                                     _ptr(sa.Z[0]), _ptr(sa.scale[0]), _ptr(sa.shift[0]), _ptr(W2.detach()),
                                     _ptr(b2.detach()), _ptr(scp), _ptr(shp), _ptr(sa.scale[1]), _ptr(sa.shift[1]),
                                     ncent, P, cin, C, _ptr(agg), C, _ptr(amax), _ptr(zsel), st)
    assert rc == 0


def first_layer():
    tcommon._chain_forward(lib, att16, pa[:4], [bn1], 1e-5)


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for rep in range(2):
    t1 = timeit(first_layer)
    to, tn = timeit(old), timeit(new)
    print("ncent %d: first conv alone %.3f ms | Z2 path %.3f ms (second conv + max %.3f) | Z2-free %.3f ms "
          "(moments + max %.3f)" % (ncent, t1, to, to - t1, tn, tn - t1))
