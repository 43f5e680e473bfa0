"""Time gridgcn_linear_fwd (LDS-staged) against gridgcn_linear_fwd_direct on edge-layer shapes.
usage: python tools/time_fwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import _lib  # noqa: E402
from grid_gcn_amd.train import common as tcommon
from grid_gcn_amd.ops import _ptr, _stream  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
SHAPES = [(3276800, 128, 128, True), (3276800, 136, 128, False), (3276800, 16, 32, False),
          (3276800, 32, 128, True), (1310720, 72, 64, False), (1310720, 64, 64, True),
          (655360, 128, 128, False), (655360, 128, 256, True), (327680, 264, 128, False)]
for E, cin, C, act in SHAPES:
    W = torch.randn(C, cin, device=dev) * 0.1
    b = torch.randn(C, device=dev)
    X = torch.randn(E, cin, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5 if act else None
    sh = torch.randn(cin, device=dev) * 0.1 if act else None
    K, ldw, nwp, nwb = tcommon.packed_sizes(C, cin)
    Wp, Bp = torch.empty(nwp, device=dev), torch.empty(ldw, device=dev)
    Wq = torch.empty(cin * ldw, device=dev)
    lib.gridgcn_pack_linear(_ptr(W), _ptr(b), C, cin, 0, cin, 0, _ptr(Wp), _ptr(Bp), None, None,
                            _ptr(Wq), None, _stream(W))
    Z1, Z2 = torch.empty(E, C, device=dev), torch.empty(E, C, device=dev)
    s1 = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    s2 = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    ps = lambda t: _ptr(t) if t is not None else None  # noqa: E731

    def old():
        return lib.gridgcn_linear_fwd(_ptr(X), E, cin, _ptr(Wp), _ptr(Bp), K, ldw, C, ps(sc), ps(sh),
                                      _ptr(Z1), _ptr(s1), _stream(X))

    def new():
        return lib.gridgcn_linear_fwd_direct(_ptr(X), E, cin, cin, _ptr(Wq), _ptr(Bp), ldw, C, ps(sc),
                                             ps(sh), _ptr(Z2), _ptr(s2), _stream(X))
    assert old() == 0 and new() == 0
    torch.cuda.synchronize()
    err = float((Z1 - Z2).abs().max())
    serr = float((s1 - s2).abs().max() / s1.abs().max())
    res = []
    for f in (old, new):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * E * cin * C
    by = 4.0 * E * (cin + C)
    print("E=%8d %3d->%3d act=%d  lds %.3f ms  direct %.3f ms (%.1f TF/s, %.2f TB/s)  maxerr %.2e sums %.1e"
          % (E, cin, C, act, res[0], res[1], fl / res[1] / 1e9, by / res[1] / 1e9, err, serr))
