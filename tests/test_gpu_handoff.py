"""Stress test of the inter-workgroup hand-offs.

The folded BatchNorm finalisation (last workgroup of gg_k_linear_fwd_direct), the 16-slot finish of the loss and
column-sum kernels, the slice tickets of gg_k_gemm_tn / gg_k_dw_reduce_direct and the Z2-free attention backward's
reduce chain hand data from many workgroups to ONE reader without an agent-scope release fence: drained agent-scope
atomics / write-through stores on the producer side, a relaxed ticket, sc1 loads on the consumer side
(tests/test_isa_handoff.py pins those instructions).  MI355X_MICROARCH asks that such a hand-off be tested "under
UNEVEN load, consumer L1-warm, checking every word".  Here every hand-off runs ROUNDS (2000) times

  * while a second stream keeps the chip busy with kernels of very different sizes (1 MB .. 256 MB element-wise
    passes and small GEMM-like reductions), so that workgroups of the tested launch start and finish unevenly;
  * on a ring of 4 buffer sets that are re-zeroed and reused -- the reader's CU has seen the previous contents of
    the very lines it reads (L1-warm) -- with the inputs alternating so that a stale line holds a DIFFERENT value;
  * with every word the last arriver produced compared, bit for bit, against the same quantity formed AFTER the
    launch from what the producers left in memory (a second launch behind a kernel boundary: the unfolded form), or
    against the result of a quiet run where the operation is bit-reproducible.

All checks are accumulated on the device (a mismatch counter per hand-off); the host reads them once at the end."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROUNDS = int(os.environ.get("GG_HANDOFF_ROUNDS", "2000"))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _Load:
    """Uneven background load on a second stream: every `every` rounds a handful of kernels whose sizes span
    1 MB .. 256 MB, plus a small strided reduction that occupies a few CUs for a long time."""

    def __init__(self):
        self.s = torch.cuda.Stream(device=DEV)
        self.bufs = [torch.empty(n, device=DEV) for n in (1 << 18, 1 << 22, 1 << 26)]
        self.outs = [torch.empty_like(b) for b in self.bufs]
        self.m = torch.randn(2048, 2048, device=DEV)
        self.k = 0

    def kick(self):
        with torch.cuda.stream(self.s):
            for j in range(4):
                i = (self.k + j * j) % 3
                torch.mul(self.bufs[i], 1.0001, out=self.outs[i])
            self.m.sum(dim=0)
            self.outs[0][: 1 << 12].add_(1.0)
        self.k += 1

    def done(self):
        torch.cuda.current_stream(DEV).wait_stream(self.s)


def _mismatch(flag, a, b):
    """flag += number of words of a that differ from b (bit compare through int views)"""
    ia = a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int64)
    ib = b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int64)
    flag += (ia != ib).sum()


@pytest.mark.parametrize("E,cin,C", [(655360, 128, 128), (40000, 64, 64), (2100, 32, 128), (200000, 16, 32)])
def test_folded_batchnorm_finalisation_under_load(E, cin, C):
    """gridgcn_linear_fwd_direct_fin: scale / shift / mean / rstd and the running statistics written by the LAST
    workgroup == gridgcn_bn_finalize_tail run afterwards on the sums the kernel left in memory."""
    from grid_gcn_amd import _lib
    from grid_gcn_amd.train import common as tcommon
    lib = _lib.load()
    st = torch.cuda.current_stream(DEV).cuda_stream
    g = torch.Generator(device=DEV).manual_seed(E + C)
    Xs = [torch.randn(E, cin, device=DEV, generator=g) * (1.0 + 0.5 * k) + 0.1 * k for k in range(2)]
    W = torch.randn(C, cin, device=DEV, generator=g) * 0.1
    b = torch.randn(C, device=DEV, generator=g)
    gamma, beta = torch.rand(C, device=DEV, generator=g) + 0.5, torch.randn(C, device=DEV, generator=g)
    K, ldw, nwp, nwb = tcommon.packed_sizes(C, cin)
    Bp, Wq = torch.empty(ldw, device=DEV), torch.empty(cin * ldw, device=DEV)
    assert lib.gridgcn_pack_linear(_p(W), _p(b), C, cin, 0, cin, 0, None, _p(Bp), None, None, _p(Wq), None, st) == 0
    Z = torch.empty(E, C, device=DEV)
    RING = 4
    rounds = ROUNDS if E < 300000 else max(200, ROUNDS // 8)     # (the big shape is 0.2 ms per launch)
    # per ring slot: sums [2C] + ticket (fp64) | running mean / var for the folded and the reference form
    state = [dict(z=torch.zeros(2 * C + 1, dtype=torch.float64, device=DEV),
                  run=torch.zeros(4, C, device=DEV), vec=torch.empty(4, C, device=DEV),
                  ref=torch.empty(4, C, device=DEV), nbt=torch.zeros(1, dtype=torch.int64, device=DEV))
             for _ in range(RING)]
    flag = torch.zeros((), dtype=torch.int64, device=DEV)
    nz = torch.zeros((), dtype=torch.int64, device=DEV)
    load = _Load()
    for r in range(rounds):
        if r % 8 == 0:
            load.kick()
        s = state[r % RING]
        s["z"].zero_()
        s["run"].zero_()
        fin = _lib.BnFin()
        fin.gamma, fin.beta = gamma.data_ptr(), beta.data_ptr()
        fin.scale, fin.shift, fin.mean, fin.rstd = [s["vec"][k].data_ptr() for k in range(4)]
        fin.running_mean, fin.running_var = s["run"][0].data_ptr(), s["run"][1].data_ptr()
        fin.num_batches_tracked = s["nbt"].data_ptr()
        fin.ticket = s["z"][2 * C:].data_ptr()
        fin.eps, fin.momentum, fin.tail = 1e-3, 0.1, 0
        rc = lib.gridgcn_linear_fwd_direct_fin(_p(Xs[r % 2]), E, cin, cin, _p(Wq), _p(Bp), ldw, C, None, None,
                                               _p(Z), _p(s["z"]), 0, 0, ctypes.byref(fin), st)
        assert rc == 0
        rc = lib.gridgcn_bn_finalize_tail(_p(s["z"]), _p(gamma), _p(beta), E, 1e-3, 0.1, C, 0,
                                          _p(s["ref"][0]), _p(s["ref"][1]), _p(s["ref"][2]), _p(s["ref"][3]),
                                          _p(s["run"][2]), _p(s["run"][3]), None, st)
        assert rc == 0
        _mismatch(flag, s["vec"], s["ref"])
        _mismatch(flag, s["run"][:2], s["run"][2:])
        nz += (s["vec"][3] > 0).sum()
    load.done()
    torch.cuda.synchronize()
    assert int(flag) == 0, "%d words of the folded finalisation differ from the two-launch form" % int(flag)
    assert int(nz) == rounds * C                 # (the comparison did see real statistics)
    assert all(int(s["nbt"]) > 0 for s in state)


@pytest.mark.parametrize("E", [655360, 70000, 300])
def test_loss_and_colsum_finish_under_load(E):
    """gridgcn_softmax_ce_loss / gridgcn_colsum_f32: the totals formed by the last arriver from the 16 slots ==
    the slots left in memory, added in slot order afterwards; the valid-row count is exact."""
    from grid_gcn_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream(DEV).cuda_stream
    g = torch.Generator(device=DEV).manual_seed(E)
    ld, ncls = 24, 21
    logits = [torch.zeros(E, ld, device=DEV) for _ in range(2)]
    for k, t in enumerate(logits):
        t[:, :ncls] = torch.randn(E, ncls, device=DEV, generator=g) * (1.0 + k)
    label = torch.randint(0, ncls, (E,), device=DEV, generator=g)
    nvalid = int((label != 0).sum())
    lse = torch.empty(E, device=DEV)
    RING = 4
    accs = [torch.zeros(544, dtype=torch.float64, device=DEV) for _ in range(RING)]
    cacc = [torch.zeros(784, dtype=torch.float64, device=DEV) for _ in range(RING)]
    loss = [torch.empty((), device=DEV) for _ in range(RING)]
    col = [torch.empty(ncls, device=DEV) for _ in range(RING)]
    flag = torch.zeros((), dtype=torch.int64, device=DEV)
    load = _Load()
    rounds = ROUNDS if E < 300000 else max(300, ROUNDS // 4)
    for r in range(rounds):
        if r % 8 == 0:
            load.kick()
        k = r % RING
        accs[k].zero_()
        cacc[k].zero_()
        x = logits[r % 2]
        assert lib.gridgcn_softmax_ce_loss(_p(x), ld, ncls, _p(label), E, 0, _p(lse), _p(accs[k]), _p(loss[k]),
                                           st) == 0
        assert lib.gridgcn_colsum_f32(_p(x), E, ld, ncls, _p(cacc[k]), _p(col[k]), st) == 0
        # the unfolded form: the slots, added in slot order behind the kernel boundary
        sl = accs[k][:256].view(16, 16)
        s0, s1 = sl[0, 0].clone(), sl[0, 1].clone()
        cs = cacc[k][:512].view(16, 32)
        ct = cs[0, :ncls].clone()
        for q in range(1, 16):
            s0 += sl[q, 0]
            s1 += sl[q, 1]
            ct += cs[q, :ncls]
        _mismatch(flag, accs[k][256:258], torch.stack([s0, s1]))
        _mismatch(flag, loss[k].reshape(1), (s0 / torch.clamp(s1, min=1.0)).float().reshape(1))
        _mismatch(flag, col[k], ct.float())
        flag += (s1 != float(nvalid)).long()
    load.done()
    torch.cuda.synchronize()
    assert int(flag) == 0, "%d words of the slotted finish differ from the slots' sum" % int(flag)


@pytest.mark.parametrize("R,m,n", [(8192, 128, 128), (40000, 64, 36), (655360, 24, 128)])
def test_gemm_tn_ticket_under_load(R, m, n):
    """gg_k_gemm_tn (slice partials -> ticket -> last arriver; release / acquire): bit-identical to a quiet run
    in every round -- the summation order is fixed, so any difference is a stale or missing slice."""
    from grid_gcn_amd.train import common as tcommon
    g = torch.Generator(device=DEV).manual_seed(R)
    As = [torch.randn(R, m, device=DEV, generator=g) for _ in range(2)]
    Bs = [torch.randn(R, n, device=DEV, generator=g) for _ in range(2)]
    quiet = [tcommon._tn_matmul(a, b).clone() for a, b in zip(As, Bs)]
    ref64 = As[0].double().t() @ Bs[0].double()
    assert float((quiet[0] - ref64).abs().max()) <= 1e-5 * float(ref64.abs().max()) * R ** 0.5
    torch.cuda.synchronize()
    flag = torch.zeros((), dtype=torch.int64, device=DEV)
    outs = [torch.empty(m, n, device=DEV) for _ in range(4)]
    load = _Load()
    rounds = ROUNDS if R < 300000 else max(300, ROUNDS // 4)
    for r in range(rounds):
        if r % 8 == 0:
            load.kick()
        o = outs[r % 4]
        tcommon._tn_matmul(As[r % 2], Bs[r % 2], out=o)
        _mismatch(flag, o, quiet[r % 2])
    load.done()
    torch.cuda.synchronize()
    assert int(flag) == 0, "%d words differ from the quiet run" % int(flag)


@pytest.mark.parametrize("ncent,P,cin,C,dense", [(8192, 5, 32, 128, False), (65536, 1, 128, 128, True),
                                                 (2048, 1, 136, 128, True)])
def test_dw_reduce_and_att_nz_chain_under_load(ncent, P, cin, C, dense):
    """gridgcn_linear_bwd (dX, dW partials, gg_k_dw_reduce_direct's slice ticket) and gridgcn_att_bwd_noz (partial
    tiles -> gg_k_att_nz_reduce -> gg_k_att_nz_finish): dW of every round == dW of a quiet run, bit for bit
    (both reduce in a fixed order)."""
    from grid_gcn_amd import _lib
    from grid_gcn_amd.train import common as tcommon
    lib = _lib.load()
    E = ncent * P
    st = torch.cuda.current_stream(DEV).cuda_stream
    g = torch.Generator(device=DEV).manual_seed(ncent + C)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    Zs, Xs = [rnd(E, C) for _ in range(2)], [rnd(E, cin) for _ in range(2)]
    scale, shift = rnd(C).abs() + 0.5, rnd(C) * 0.1
    mean, rstd = rnd(C) * 0.1, rnd(C).abs() + 0.5
    amax = torch.randint(0, P, (ncent, C), device=DEV, dtype=torch.int32, generator=g).to(torch.uint8)
    gval = rnd(ncent, C)
    dY = rnd(E, C) if dense else None
    Wt = rnd(C, cin)
    Wb, Wg = tcommon.pack_tiles(Wt), tcommon.pack_groups(Wt)
    ndx = min(cin, 256)
    Wdx = torch.empty(C * 32 * 8, device=DEV)
    assert lib.gridgcn_pack_linear(_p(Wt), None, C, cin, 0, cin, ndx, None, None, None, None, None, _p(Wdx), st) == 0
    nbytes = ctypes.c_size_t(0)
    lib.gridgcn_linear_bwd_workspace_bytes(E, cin, C, ctypes.byref(nbytes))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=DEV)
    dX = torch.empty(E, cin, device=DEV)
    sums = [torch.randn(2 * C, device=DEV, generator=g).double() for _ in range(2)]

    def run(k, dW, v):
        rc = lib.gridgcn_linear_bwd_fin(
            _p(dY), _p(Zs[k]), _p(scale), _p(shift), _p(mean), _p(rstd), _p(sums[k]), _p(v[0]), _p(v[1]), _p(v[2]),
            _p(v[3]), _p(Xs[k]), None, None, None, None, _p(Wb), _p(Wg), _p(Wdx), ndx, E, C, cin, cin, 0,
            C if dense else 0, 0, 0, 0, _p(dX), _p(dW), None, None if dense else _p(amax),
            None if dense else _p(gval), P, _p(ws), nbytes.value, st)
        assert rc == 0

    quiet = []
    for k in range(2):
        dW, v = torch.empty(C, cin, device=DEV), torch.empty(4, C, device=DEV)
        run(k, dW, v)
        quiet.append((dW, v))
    torch.cuda.synchronize()
    flag = torch.zeros((), dtype=torch.int64, device=DEV)
    outs = [(torch.empty(C, cin, device=DEV), torch.empty(4, C, device=DEV)) for _ in range(4)]
    load = _Load()
    rounds = max(200, ROUNDS // 4)
    for r in range(rounds):
        if r % 8 == 0:
            load.kick()
        dW, v = outs[r % 4]
        run(r % 2, dW, v)
        _mismatch(flag, dW, quiet[r % 2][0])
        _mismatch(flag, v, quiet[r % 2][1])
    if cin == 32 and C == 128 and not dense:
        # the Z2-free attention backward on the same shape
        Z1 = Xs
        s1v, h1v, m1v, r1v = rnd(cin).abs() + 0.5, rnd(cin) * 0.1, rnd(cin) * 0.1, rnd(cin).abs() + 0.5
        W2, b2 = rnd(C, cin) * 0.2, rnd(C) * 0.1
        nb2 = ctypes.c_size_t(0)
        assert lib.gridgcn_att_bwd_noz_workspace_bytes(E, cin, C, ctypes.byref(nb2)) == 0
        ws2 = torch.empty(nb2.value, dtype=torch.uint8, device=DEV)
        dA1 = torch.empty(E, cin, device=DEV)

        def run_nz(k, dW, v, acc):
            acc.zero_()
            rc = lib.gridgcn_att_bwd_noz(_p(Z1[k]), _p(s1v), _p(h1v), _p(m1v), _p(r1v), _p(W2), _p(b2), _p(scale),
                                         _p(mean), _p(rstd), _p(sums[k]), _p(amax), _p(gval), int(P), E, cin, C,
                                         _p(dA1), _p(dW), _p(v[0]), _p(v[1]), _p(v[2]), _p(v[3]),
                                         _p(acc[:2 * cin]), _p(acc[2 * cin:]), _p(ws2), nb2.value, st)
            assert rc == 0

        accs = [torch.zeros(3 * cin, dtype=torch.float64, device=DEV) for _ in range(4)]
        quiet = []
        for k in range(2):
            dW, v = torch.empty(C, cin, device=DEV), torch.empty(4, C, device=DEV)
            run_nz(k, dW, v, accs[k])
            quiet.append((dW, v))
        torch.cuda.synchronize()
        for r in range(rounds):
            if r % 8 == 0:
                load.kick()
            dW, v = outs[r % 4]
            run_nz(r % 2, dW, v, accs[r % 4])
            _mismatch(flag, dW, quiet[r % 2][0])
            _mismatch(flag, v, quiet[r % 2][1])
    load.done()
    torch.cuda.synchronize()
    assert int(flag) == 0, "%d words differ from the quiet run" % int(flag)
