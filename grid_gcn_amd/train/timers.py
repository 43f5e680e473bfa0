"""Micro-benchmarks of single library calls on synthetic tensors (bench.py's roofline lines)."""
import ctypes

import torch

from .. import _lib
from ..ops import _ptr, _stream
from .options import OPT
from .common import (pack_groups, pack_tiles, packed_sizes)

def median_ms(call, iters=50, warm=5, device="cuda:0"):
    """Median device time of `call()` over `iters` launches, each bracketed by its own pair of events
    on the current stream (a mean over a handful of back-to-back launches moved by 10-15 % from box
    to box and with the clock state the previous benchmark left behind)."""
    with torch.cuda.device(device):
        for _ in range(warm):
            call()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(iters)]
        for e0, e1 in ev:
            e0.record()
            call()
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2]


def time_linear_fwd(E, cin, C, iters=50, device="cuda:0"):
    """Time one gridgcn_linear_fwd_direct launch (previous layer's BatchNorm+ReLU applied on the fly,
    statistics epilogue) on synthetic tensors.  Returns the median ms/launch."""
    lib = _lib.load()
    g = torch.Generator(device=device).manual_seed(0)
    X = torch.randn(E, cin, device=device, generator=g)
    W = torch.randn(C, cin, device=device, generator=g) * 0.1
    b = torch.randn(C, device=device, generator=g)
    sc = torch.rand(cin, device=device, generator=g) + 0.5
    sh = torch.randn(cin, device=device, generator=g) * 0.1
    K, ldw, nwp, nwb = packed_sizes(C, cin)
    Bp, Wq = torch.empty(ldw, device=device), torch.empty(cin * ldw, device=device)
    _lib.check(lib.gridgcn_pack_linear(_ptr(W), _ptr(b), C, cin, 0, cin, 0, None, _ptr(Bp), None,
                                       None, _ptr(Wq), None, _stream(W)), "pack")
    Z = torch.empty(E, C, device=device)
    sums = torch.zeros(2 * C, dtype=torch.float64, device=device)

    def call():
        _lib.check(lib.gridgcn_linear_fwd_direct(_ptr(X), E, cin, cin, _ptr(Wq), _ptr(Bp), ldw, C,
                                                 _ptr(sc), _ptr(sh), _ptr(Z), _ptr(sums),
                                                 _stream(X)), "gridgcn_linear_fwd_direct")
    return median_ms(call, iters, device=device)


def time_linear_bwd(ncent, P, cin, C, iters=50, device="cuda:0", ndx=0, prev_bn=False, dense=False):
    """Time one gridgcn_linear_bwd call (dX kernel + dW kernel + dW reduce) on synthetic tensors
    shaped like the last pt layer of a GridConv edge block (sparse upstream gradient, input gradient
    for the first `ndx` columns; cin = padded row length).  Used by bench.py for the roofline of the
    dominant kernels of the training step.  prev_bn: the layer's input is the raw output of a
    BatchNorm'd layer (as the second attention conv's is): its BatchNorm+ReLU is applied on the fly
    and its BatchNorm-backward sums are accumulated.  dense: a dense upstream gradient [E, C]
    instead of the max-pool's sparse one (the per-point layers of the head).  Returns the median
    ms/call."""
    lib = _lib.load()
    E = ncent * P
    g = torch.Generator(device=device).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=device, generator=g)  # noqa: E731
    Z, X = rnd(E, C), rnd(E, cin)
    scale, shift = rnd(C).abs() + 0.5, rnd(C) * 0.1
    mean, rstd = rnd(C) * 0.1, rnd(C).abs() + 0.5
    m1, m2 = rnd(C) * 1e-3, rnd(C) * 1e-3
    amax = torch.randint(0, P, (ncent, C), device=device, dtype=torch.int32, generator=g).to(torch.uint8)
    gval = rnd(ncent, C)
    dY = rnd(E, C) if dense else None
    Wt = rnd(C, cin)
    Wb, Wg = pack_tiles(Wt), pack_groups(Wt)
    ndx = (ndx or min(cin, 256)) if (OPT.DIRECT_DX and C % 8 == 0) else 0
    Wdx = torch.empty(C * 32 * 8, device=device)
    if ndx:
        _lib.check(lib.gridgcn_pack_linear(_ptr(Wt), None, C, cin, 0, cin, ndx, None, None, None,
                                           None, None, _ptr(Wdx), _stream(Wt)), "pack")
    dX = torch.empty(E, cin, device=device)
    dW = torch.empty(C, cin, device=device)
    nbytes = ctypes.c_size_t(0)
    lib.gridgcn_linear_bwd_workspace_bytes(E, cin, C, ctypes.byref(nbytes))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
    pv = [rnd(cin).abs() + 0.5, rnd(cin) * 0.1, rnd(cin) * 0.1, rnd(cin).abs() + 0.5]
    pb = [_ptr(t) for t in pv] if prev_bn else [None] * 4
    psums = torch.zeros(2 * cin, dtype=torch.float64, device=device)

    def call():
        rc = lib.gridgcn_linear_bwd(_ptr(dY) if dense else None, _ptr(Z), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd),
                                    _ptr(m1), _ptr(m2), _ptr(X), pb[0], pb[1], pb[2], pb[3], _ptr(Wb),
                                    _ptr(Wg), _ptr(Wdx) if ndx else None, ndx, E, C, cin, cin, 0, C if dense else 0,
                                    _ptr(dX), _ptr(dW), _ptr(psums) if prev_bn else None,
                                    None if dense else _ptr(amax), None if dense else _ptr(gval), P,
                                    _ptr(ws), nbytes.value, _stream(Z))
        _lib.check(rc, "gridgcn_linear_bwd")
    return median_ms(call, iters, device=device)


def time_att_bwd_noz(ncent, P, cin, C, iters=50, device="cuda:0"):
    """Time one gridgcn_att_bwd_noz call (backward of an up layer's second attention conv without its [E, C]
    pre-activation: gg_k_att_bwd_nz + its reduce and finish launches) on synthetic tensors; bench.py's
    roofline of the dominant kernel of the step.  Returns the median ms/call."""
    lib = _lib.load()
    E = ncent * P
    g = torch.Generator(device=device).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=device, generator=g)  # noqa: E731
    Z1 = rnd(E, cin)
    s1v, h1v, m1v, r1v = rnd(cin).abs() + 0.5, rnd(cin) * 0.1, rnd(cin) * 0.1, rnd(cin).abs() + 0.5
    W2, b2 = rnd(C, cin) * 0.2, rnd(C) * 0.1
    s2v, m2v, r2v = rnd(C).abs() + 0.5, rnd(C) * 0.1, rnd(C).abs() + 0.5
    sums_a = torch.zeros(2 * C, dtype=torch.float64, device=device)
    amax = torch.randint(0, P, (ncent, C), device=device, dtype=torch.int32, generator=g).to(torch.uint8)
    ga = rnd(ncent, C)
    dA1 = torch.empty(E, cin, device=device)
    dW2 = torch.empty(C, cin, device=device)
    v = torch.empty(4, C, device=device)
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.gridgcn_att_bwd_noz_workspace_bytes(E, cin, C, ctypes.byref(nbytes)), "att_bwd_noz_workspace")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)

    def call():
        acc = torch.zeros(3 * cin, dtype=torch.float64, device=device)
        rc = lib.gridgcn_att_bwd_noz(_ptr(Z1), _ptr(s1v), _ptr(h1v), _ptr(m1v), _ptr(r1v), _ptr(W2), _ptr(b2),
                                     _ptr(s2v), _ptr(m2v), _ptr(r2v), _ptr(sums_a), _ptr(amax), _ptr(ga), int(P),
                                     E, cin, C, _ptr(dA1), _ptr(dW2), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]),
                                     _ptr(v[3]), _ptr(acc[:2 * cin]), _ptr(acc[2 * cin:]), _ptr(ws), nbytes.value,
                                     _stream(Z1))
        _lib.check(rc, "gridgcn_att_bwd_noz")
    return median_ms(call, iters, device=device)
