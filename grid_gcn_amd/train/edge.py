"""The GridConv edge block of the segmentation nets in training mode (gcn_module_g_att.py:120-287):
source-side first conv, attention chain, product + max, and the fused backward."""
import ctypes

import torch

from .. import _lib
from ..ops import _ptr, _stream
from .options import OPT
from .common import (  # noqa: F401
    PACKS, _Chain, _chain_backward, _chain_forward, _gemm_small, _mm_nn, _mm_nt, _momentum, _small_ok,
    _stats_written, _tn_matmul, _zeros, supported,)

class _EdgeBlockTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nf, att_vec, meta, *params):
        """nf [E,cin], att_vec [E,10]; params = pt chain params + att chain params (4 per layer);
        meta = (eps, pt bns, att bns, ncent, P).  Returns agg [ncent, C]."""
        lib = _lib.load()
        eps, bns_p, bns_a, ncent, P, rot = meta
        Lp, La = len(bns_p), len(bns_a)
        nf, att_vec = nf.contiguous(), att_vec.contiguous()
        dev = nf.device
        with torch.cuda.device(dev):
            # only the feature columns of nf (the leading ones) need a gradient
            nfeat = params[0].shape[1] - rot
            sp = _chain_forward(lib, nf, params[:4 * Lp], bns_p, eps, rot,
                                nfeat if ctx.needs_input_grad[0] else 0)
            sa = _chain_forward(lib, att_vec, params[4 * Lp:], bns_a, eps)
            C = sp.Z[-1].shape[1]
            agg = torch.empty((ncent, C), dtype=torch.float32, device=dev)
            amax = torch.empty((ncent, C), dtype=torch.uint8, device=dev)
            zsel = torch.empty((2, ncent, C), dtype=torch.float32, device=dev)
            rc = lib.gridgcn_pairmax_fwd(_ptr(sp.Z[-1]), _ptr(sa.Z[-1]), _ptr(sp.scale[-1]),
                                         _ptr(sp.shift[-1]), _ptr(sa.scale[-1]), _ptr(sa.shift[-1]),
                                         ncent, P, C, _ptr(agg), C, _ptr(amax), _ptr(zsel),
                                         _stream(nf))
            _lib.check(rc, "gridgcn_pairmax_fwd")
        ctx.dims = (Lp, La, ncent, P, rot, params[0].shape[1], params[4 * Lp].shape[1])
        ctx.ndx = (sp.ndx, sa.ndx)
        ctx.save_for_backward(
            nf, att_vec, amax, zsel,
            *sp.Z, *sp.scale, *sp.shift, *sp.mean, *sp.rstd, *sp.Wb, *sp.Wg, *sp.Wdx,
            *sa.Z, *sa.scale, *sa.shift, *sa.mean, *sa.rstd, *sa.Wb, *sa.Wg, *sa.Wdx)
        ctx.mark_non_differentiable(amax)
        return agg

    @staticmethod
    def backward(ctx, dagg):
        lib = _lib.load()
        Lp, La, ncent, P, rot, cwp, cwa = ctx.dims
        t = ctx.saved_tensors
        nf, att_vec, amax, zsel = t[0], t[1], t[2], t[3]
        o = 4
        pZ, pS, pH, pM, pR, pWb, pWg, pWx = (t[o + k * Lp:o + (k + 1) * Lp] for k in range(8))
        o += 8 * Lp
        aZ, aS, aH, aM, aR, aWb, aWg, aWx = (t[o + k * La:o + (k + 1) * La] for k in range(8))
        dev = nf.device
        dagg = dagg.contiguous()
        C = pZ[-1].shape[1]
        with torch.cuda.device(dev):
            gp = torch.empty((ncent, C), dtype=torch.float32, device=dev)
            ga = torch.empty((ncent, C), dtype=torch.float32, device=dev)
            sums_pa = _zeros((2, 2 * C), torch.float64, dev)
            sums_p, sums_a = sums_pa[0], sums_pa[1]
            rc = lib.gridgcn_pairmax_bwd(_ptr(pZ[-1]), _ptr(aZ[-1]), _ptr(pS[-1]), _ptr(pH[-1]),
                                         _ptr(pM[-1]), _ptr(pR[-1]), _ptr(aS[-1]), _ptr(aH[-1]),
                                         _ptr(aM[-1]), _ptr(aR[-1]), _ptr(dagg), _ptr(amax), ncent,
                                         P, C, C, _ptr(gp), _ptr(ga), _ptr(sums_p), _ptr(sums_a),
                                         _ptr(zsel), _stream(nf))
            _lib.check(rc, "gridgcn_pairmax_bwd")
            dnf, grads_p = _chain_backward(lib, nf, pZ, pS, pH, pM, pR, pWb, pWg, pWx, ctx.ndx[0],
                                           sums_p, None, (amax, gp, P), ctx.needs_input_grad[0],
                                           cwp, rot)
            _, grads_a = _chain_backward(lib, att_vec, aZ, aS, aH, aM, aR, aWb, aWg, aWx, ctx.ndx[1],
                                         sums_a, None, (amax, ga, P), False, cwa, 0)
        return (dnf, None, None) + tuple(grads_p) + tuple(grads_a)


def _att_bwd_noz(lib, att16, Z1, aS, aH, aM, aR, aWb, aWg, aWx, ndxs, W2, b2, sums_a, amax, ga, P, cwa, st, mom=None):
    """backward of the attention chain (10 -> 32 -> 128) of an up layer without the second conv's [E, 128]
    pre-activation: gridgcn_att_bwd_noz for the second conv (dA1, dW2, its BatchNorm vectors, the BatchNorm-
    backward sums of the first layer), then the ordinary chain backward for the first conv.  Returns the
    chain's gradient list [dW, db, dgamma, dbeta] * 2.
    mom: the [17 * 64] fp64 moments (S2, S1) of the forward (_att_fwd_noz), or None: accumulated here."""
    E, dev = att16.shape[0], att16.device
    C, cin = W2.shape
    dA1 = torch.empty((E, cin), dtype=torch.float32, device=dev)
    dW2 = torch.empty((C, cin), dtype=torch.float32, device=dev)
    v = torch.empty((4, C), dtype=torch.float32, device=dev)          # m1, m2, dgamma, dbeta
    acc = _zeros(3 * cin, torch.float64, dev)
    psums, s1 = acc[:2 * cin], acc[2 * cin:]
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.gridgcn_att_bwd_noz_workspace_bytes(E, cin, C, ctypes.byref(nbytes)), "att_bwd_noz_workspace")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    t_end = OPT.TIMERS.bracket(("linear_bwd", E, cin, C)) if OPT.TIMERS is not None else None
    if mom is not None:
        rc = lib.gridgcn_att_bwd_noz_mom(_ptr(Z1), _ptr(aS[0]), _ptr(aH[0]), _ptr(aM[0]), _ptr(aR[0]),
                                         _ptr(W2.detach()), _ptr(b2.detach()), _ptr(aS[1]), _ptr(aM[1]), _ptr(aR[1]),
                                         _ptr(sums_a), _ptr(amax), _ptr(ga), int(P), E, cin, C, _ptr(mom), _ptr(dA1),
                                         _ptr(dW2), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(v[3]), _ptr(psums),
                                         _ptr(ws), nbytes.value, st)
    else:
        rc = lib.gridgcn_att_bwd_noz(_ptr(Z1), _ptr(aS[0]), _ptr(aH[0]), _ptr(aM[0]), _ptr(aR[0]),
                                     _ptr(W2.detach()), _ptr(b2.detach()), _ptr(aS[1]), _ptr(aM[1]), _ptr(aR[1]),
                                     _ptr(sums_a), _ptr(amax), _ptr(ga), int(P), E, cin, C, _ptr(dA1), _ptr(dW2),
                                     _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(v[3]), _ptr(psums), _ptr(s1),
                                     _ptr(ws), nbytes.value, st)
    if t_end is not None:
        t_end.record()
    _lib.check(rc, "gridgcn_att_bwd_noz")
    _, g0 = _chain_backward(lib, att16, [Z1], aS[:1], aH[:1], aM[:1], aR[:1], aWb[:1], aWg[:1], aWx[:1],
                            ndxs[:1], psums, dA1, None, False, cwa, 0)
    db2 = _zeros(C, torch.float32, dev)          # a bias in front of a BatchNorm: sum(dZ) == 0
    return list(g0) + [dW2, db2, v[2], v[3]]


def _att_fwd_noz(lib, att16, pa, bns_a, eps, st):
    """forward of the attention chain (10 -> 32 -> 128) of an up layer without the second conv's [E, 128] output:
    the first conv as ever, the second BatchNorm's vectors from the moments of the 32-wide activation
    (gridgcn_att_bn2_moments; csrc/gridgcn_attfwd.hip).  Returns the chain record with an EMPTY second Z."""
    sa = _chain_forward(lib, att16, pa[:4], bns_a[:1], eps)
    E, dev = att16.shape[0], att16.device
    W2, b2, g2, be2 = pa[4:8]
    C, cin = W2.shape
    vec = torch.empty((4, C), dtype=torch.float32, device=dev)
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.gridgcn_att_fwd_noz_workspace_bytes(E, cin, C, ctypes.byref(nbytes)), "att_fwd_noz_workspace")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    bn = bns_a[1]
    track = bn is not None and bn.track_running_stats
    rc = lib.gridgcn_att_bn2_moments(
        _ptr(sa.Z[0]), _ptr(sa.scale[0]), _ptr(sa.shift[0]), _ptr(W2.detach()), _ptr(b2.detach()),
        _ptr(g2.detach()), _ptr(be2.detach()), E, cin, C, eps, _momentum(bn) if track else 0.0,
        _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]), _ptr(vec[3]),
        _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None,
        _ptr(bn.num_batches_tracked) if track else None, None, _ptr(ws), nbytes.value, st)
    _lib.check(rc, "gridgcn_att_bn2_moments")
    if track:
        _stats_written(bn)
    none = torch.empty(0, dtype=torch.float32, device=dev)
    sa.Z.append(none); sa.scale.append(vec[0]); sa.shift.append(vec[1])
    sa.mean.append(vec[2]); sa.rstd.append(vec[3])
    sa.Wb.append(none); sa.Wg.append(none); sa.Wdx.append(none); sa.ndx.append(0)
    # the moments S2 / S1 of the 32-wide activation (fp64, [17 * 64]) stay in the workspace: the backward takes them
    # from there instead of accumulating them again (OPT.NOZ_BWD_MOMENTS; a view: it keeps the workspace alive)
    sa.mom = None
    if OPT.NOZ_BWD_MOMENTS and lib.gridgcn_att_bwd_noz_mom_supported(E, cin, C, 5) == 1:
        off = ctypes.c_size_t(0)
        _lib.check(lib.gridgcn_att_moments_offset(E, cin, C, ctypes.byref(off)), "att_moments_offset")
        sa.mom = ws[off.value:off.value + 17 * 64 * 8].view(torch.float64)
    return sa


class _EdgeBlockSrcTrain(torch.autograd.Function):
    """The whole GridConv edge block from (src, nebidx, cent): the first conv of the point MLP is
    applied to the SOURCE points (Ysrc = features * Wf^T, [B*Nsrc, C0]) and gathered, instead of
    being applied to the gathered [E, 3+Cf] tensor (csrc/gridgcn_edgelin.hip); the remaining pt
    layers, the att MLP and the product/max run as in _EdgeBlockTrain."""

    @staticmethod
    def forward(ctx, src, nebidx, cent, meta, *params):
        lib = _lib.load()
        eps, bns_p, bns_a, geo, out = meta
        Lp, La = len(bns_p), len(bns_a)
        B, Nsrc, Cs = src.shape
        _, O, P = nebidx.shape
        E, R, Cf = B * O * P, B * Nsrc, Cs - 4
        dev = src.device
        W0, b0, g0, be0 = params[:4]
        C0 = W0.shape[0]
        rot = 3 if geo else 0
        with torch.cuda.device(dev):
            st = _stream(src)
            feat = src.detach()[..., 4:].reshape(R, Cf)
            # [R, C0]: once per source point
            if _small_ok(R, C0) and Cf % 8 == 0 and Cf <= 512:
                Ysrc = _gemm_small(0, feat, W0.detach()[:, rot:], torch.empty((R, C0), dtype=torch.float32,
                                                                             device=dev), R, C0, Cf)
            else:
                Ysrc = _mm_nt(feat, W0.detach()[:, rot:])
            # rows 0..2: geo_vec weights [3][C0] (zeros without geo_vec), row 3: bias
            wgb = PACKS.get_wgb(lib, W0, b0, geo)
            Wg = wgb if geo else None
            # a single-layer point MLP never materialises Z0: its consumers recompute it
            noz = Lp == 1 and OPT.NO_Z0 and C0 % 4 == 0
            Z0 = None if noz else torch.empty((E, C0), dtype=torch.float32, device=dev)
            att16 = torch.empty((E, 16), dtype=torch.float32, device=dev)
            sums0 = _zeros(2 * C0, torch.float64, dev)
            gsum = gg = None
            if noz and OPT.SRC_STATS and (Nsrc + 1) * 28 <= 150 * 1024 and C0 <= 1024 and E >= OPT.SRC_STATS_MIN_EDGES:
                # statistics of the never-stored Z0 from per-source counts and geo_vec sums: no edge x
                # channel pass (csrc/gridgcn_edgelin.hip, gg_k_edge_geo_fwd)
                nbytes = ctypes.c_size_t(0)
                lib.gridgcn_edge_geo_forward_workspace_bytes(B, Nsrc, O, P, ctypes.byref(nbytes))
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
                gsum = torch.empty((R, 4), dtype=torch.float32, device=dev)
                gg = _zeros(12, torch.float64, dev)
                rc = lib.gridgcn_edge_geo_forward(
                    _ptr(Ysrc), _ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B, Nsrc, Cs, O, P,
                    C0, _ptr(Wg) if geo else None, _ptr(wgb[3]), _ptr(att16), _ptr(gsum), _ptr(gg),
                    _ptr(sums0), _ptr(ws), nbytes.value, st)
            else:
                rc = lib.gridgcn_edge_lin0_forward(
                    _ptr(Ysrc), _ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B, Nsrc, Cs, O, P,
                    C0, _ptr(Wg) if geo else None, _ptr(wgb[3]), None if noz else _ptr(Z0),
                    _ptr(att16), _ptr(sums0), st)
            _lib.check(rc, "gridgcn_edge_lin0_forward")
            vec0 = torch.empty((4, C0), dtype=torch.float32, device=dev)
            bn = bns_p[0]
            track = bn.track_running_stats
            rc = lib.gridgcn_bn_finalize(
                _ptr(sums0), _ptr(g0.detach()), _ptr(be0.detach()), E, eps,
                _momentum(bn) if track else 0.0, C0, _ptr(vec0[0]), _ptr(vec0[1]), _ptr(vec0[2]),
                _ptr(vec0[3]), _ptr(bn.running_mean) if track else None,
                _ptr(bn.running_var) if track else None,
                _ptr(bn.num_batches_tracked) if track else None, st)
            _lib.check(rc, "gridgcn_bn_finalize")
            if track:
                _stats_written(bn)
            if Lp > 1:
                sp = _chain_forward(lib, Z0, params[4:4 * Lp], bns_p[1:], eps, 0, C0,
                                    prev_bn=(vec0[0], vec0[1]))
                Zl, scl, shl = sp.Z[-1], sp.scale[-1], sp.shift[-1]
            else:
                sp = _Chain()
                Zl, scl, shl = Z0, vec0[0], vec0[1]
            pa = params[4 * Lp:]
            C = pa[4 * (La - 1)].shape[0]
            A0 = pa[0].shape[0]
            ncent = B * O
            agg = out if out is not None else torch.empty((ncent, C), dtype=torch.float32,
                                                           device=dev)
            lda = agg.stride(0)
            amax = torch.empty((ncent, C), dtype=torch.uint8, device=dev)
            zsel = torch.empty((2, ncent, C), dtype=torch.float32, device=dev)
            # bf16 mode: the [E, C] pre-activation of the second attention conv -- the largest tensor of
            # the step, written once and read twice -- is STORED as bf16 (its writer's fp32
            # accumulators are rounded once; BatchNorm statistics from the fp32 values).  Only where
            # both readers take it: the source-side max kernel and the fused attention backward.
            # (E >= 32: below that the fused backward declines and nothing else reads a bf16 Z)
            nz_shape = noz and La == 2 and A0 == 32 and C == 128 and E >= 32 and att16.shape[1] == 16
            # the limits of the Z2-free FORWARD kernel, asked of the library (ONE statement of them: gg_att_fwd_ok)
            nzf_shape = nz_shape and lib.gridgcn_att_pairmax_fwd_supported(ncent, O, P, A0, C, lda, R) == 1
            # (bf16 mode, OPT.NOZ_IN_BF16: where the Z2-free PAIR of kernels applies -- both of them, or the step would
            #  write an fp32 Z2 from the bf16 chain and run the fp32 recompute backward on it -- it is taken in bf16 mode
            #  too: it is fp32-exact and moves fewer bytes than the bf16-stored tensor does)
            nz16 = (OPT.NOZ_IN_BF16 and OPT.NOZ_ATT_BWD and OPT.NOZ_ATT_FWD and nzf_shape
                    and lib.gridgcn_get_mlp_precision() == 1)
            z16 = (OPT.Z16_STORAGE and noz and La == 2 and A0 in (16, 32) and C in (64, 128) and E >= 32
                   and lib.gridgcn_get_mlp_precision() == 1 and not nz16
                   and lib.gridgcn_get_option(_lib.OPT_ATT_BWD_FUSED) == 1)
            # the backward of the second attention conv needs no Z2 (gridgcn_att_bwd_noz): decided HERE, because
            # the tensor is then not saved ...
            nz = (OPT.NOZ_ATT_BWD and nz_shape and not z16 and (lib.gridgcn_get_mlp_precision() == 0 or nz16))
            # ... and neither does the forward (gridgcn_att_bn2_moments, gridgcn_att_pairmax_fwd): the [E, 128]
            # tensor is then never written at all
            nzf = nz and OPT.NOZ_ATT_FWD and nzf_shape
            if nzf:
                sa = _att_fwd_noz(lib, att16, pa, bns_a, eps, st)
                rc = lib.gridgcn_att_pairmax_fwd(
                    _ptr(Ysrc), _ptr(nebidx), _ptr(att16), _ptr(Wg) if geo else None, _ptr(wgb[3]), B, Nsrc, O,
                    _ptr(sa.Z[0]), _ptr(sa.scale[0]), _ptr(sa.shift[0]), _ptr(pa[4].detach()), _ptr(pa[5].detach()),
                    _ptr(scl), _ptr(shl), _ptr(sa.scale[1]), _ptr(sa.shift[1]), ncent, P, A0, C, _ptr(agg), lda,
                    _ptr(amax), _ptr(zsel), st)
            else:
                sa = _chain_forward(lib, att16, pa, bns_a, eps, z16_last=z16)
            if nzf:
                pass
            elif noz:
                rc = lib.gridgcn_pairmax_fwd_src_z(
                    _ptr(Ysrc), _ptr(nebidx), _ptr(att16), _ptr(Wg) if geo else None, _ptr(wgb[3]),
                    B, Nsrc, O, _ptr(sa.Z[-1]), 1 if z16 else 0, _ptr(scl), _ptr(shl),
                    _ptr(sa.scale[-1]), _ptr(sa.shift[-1]), ncent, P, C, _ptr(agg), lda, _ptr(amax),
                    _ptr(zsel), st)
            else:
                rc = lib.gridgcn_pairmax_fwd(_ptr(Zl), _ptr(sa.Z[-1]), _ptr(scl), _ptr(shl),
                                             _ptr(sa.scale[-1]), _ptr(sa.shift[-1]), ncent, P, C,
                                             _ptr(agg), lda, _ptr(amax), _ptr(zsel), st)
            _lib.check(rc, "gridgcn_pairmax_fwd")
        ctx.dims = (Lp, La, B, Nsrc, Cs, O, P, C0, rot, params[4 * Lp].shape[1], noz)
        ctx.ndx = (sp.ndx, sa.ndx)
        ctx.nz = nz
        mom = getattr(sa, "mom", None) if nzf else None
        ctx.mom = mom is not None
        ctx.geo = gsum is not None      # (per-source geo sums of the forward: the backward's geo pass is skipped)
        saZ = list(sa.Z)
        if nz:
            saZ[-1] = torch.empty(0, dtype=torch.float32, device=dev)     # Z2: read by nobody any more
        ctx.save_for_backward(
            src, nebidx, att16, amax, Ysrc if noz else Z0, vec0, W0, zsel, wgb,
            *sp.Z, *sp.scale, *sp.shift, *sp.mean, *sp.rstd, *sp.Wb, *sp.Wg, *sp.Wdx,
            *saZ, *sa.scale, *sa.shift, *sa.mean, *sa.rstd, *sa.Wb, *sa.Wg, *sa.Wdx,
            *((pa[4], pa[5]) if nz else ()), *((gsum, gg) if gsum is not None else ()),
            *((mom,) if mom is not None else ()))
        ctx.mark_non_differentiable(amax)
        return agg if out is not None else agg.reshape(B, O, C)

    @staticmethod
    def backward(ctx, dagg):
        lib = _lib.load()
        Lp, La, B, Nsrc, Cs, O, P, C0, rot, cwa, noz = ctx.dims
        t = ctx.saved_tensors
        src, nebidx, att16, amax, Z0, vec0, W0, zsel, wgb = t[:9]
        Ysrc = None
        if noz:
            Ysrc, Z0 = Z0, None
        o = 9
        L1 = Lp - 1
        pZ, pS, pH, pM, pR, pWb, pWg, pWx = (t[o + k * L1:o + (k + 1) * L1] for k in range(8))
        o += 8 * L1
        aZ, aS, aH, aM, aR, aWb, aWg, aWx = (t[o + k * La:o + (k + 1) * La] for k in range(8))
        o += 8 * La
        nz = ctx.nz
        W2, b2 = (t[o], t[o + 1]) if nz else (None, None)
        o += 2 if nz else 0
        gsum_f, gg_f = (t[o], t[o + 1]) if ctx.geo else (None, None)
        o += 2 if ctx.geo else 0
        mom = t[o] if ctx.mom else None
        dev = src.device
        E, R, Cf, ncent = B * O * P, B * Nsrc, Cs - 4, B * O
        Zl = pZ[-1] if L1 else Z0
        lS, lH, lM, lR = (pS[-1], pH[-1], pM[-1], pR[-1]) if L1 else (vec0[0], vec0[1], vec0[2],
                                                                        vec0[3])
        C = amax.shape[1]
        if not (dagg.dim() == 2 and dagg.stride(1) == 1):       # (a concat half: used in place)
            dagg = dagg.contiguous().reshape(ncent, C)
        with torch.cuda.device(dev):
            st = _stream(src)
            gp = torch.empty((ncent, C), dtype=torch.float32, device=dev)
            ga = torch.empty((ncent, C), dtype=torch.float32, device=dev)
            sums_pa = _zeros((2, 2 * C), torch.float64, dev)
            sums_p, sums_a = sums_pa[0], sums_pa[1]
            # (the arg-max pre-activations come from zsel: Zl may not exist)
            if nz:
                # ga with the attention ReLU mask applied: its consumer has no pre-activation to mask with
                rc = lib.gridgcn_pairmax_bwd_masked(_ptr(lS), _ptr(lH), _ptr(lM), _ptr(lR), _ptr(aS[-1]),
                                                    _ptr(aH[-1]), _ptr(aM[-1]), _ptr(aR[-1]), _ptr(dagg),
                                                    _ptr(amax), ncent, P, C, dagg.stride(0), _ptr(gp), _ptr(ga),
                                                    _ptr(sums_p), _ptr(sums_a), _ptr(zsel), st)
            else:
                rc = lib.gridgcn_pairmax_bwd(_ptr(Zl) if Zl is not None else None,
                                             # (a bf16-stored attention tensor: the values at the arg
                                             #  max come from zsel)
                                             _ptr(aZ[-1]) if aZ[-1].dtype == torch.float32 else None,
                                             _ptr(lS), _ptr(lH), _ptr(lM),
                                             _ptr(lR), _ptr(aS[-1]), _ptr(aH[-1]), _ptr(aM[-1]),
                                             _ptr(aR[-1]), _ptr(dagg), _ptr(amax), ncent, P, C,
                                             dagg.stride(0), _ptr(gp),
                                             _ptr(ga), _ptr(sums_p), _ptr(sums_a), _ptr(zsel), st)
            _lib.check(rc, "gridgcn_pairmax_bwd")
            if nz:
                grads_a = _att_bwd_noz(lib, att16, aZ[0], aS, aH, aM, aR, aWb, aWg, aWx, ctx.ndx[1], W2, b2,
                                       sums_a, amax, ga, P, cwa, st, mom=mom)
            else:
                _, grads_a = _chain_backward(lib, att16, aZ, aS, aH, aM, aR, aWb, aWg, aWx,
                                             ctx.ndx[1], sums_a, None, (amax, ga, P), False, cwa, 0)
            if L1:
                dY0, grads_rest, sums0 = _chain_backward(
                    lib, Z0, pZ, pS, pH, pM, pR, pWb, pWg, pWx, ctx.ndx[0], sums_p, None,
                    (amax, gp, P), True, None, 0, prev_bn=(vec0[0], vec0[1], vec0[2], vec0[3]))
                sparse0 = (None, None)
            else:
                dY0, grads_rest, sums0 = None, [], sums_p
                sparse0 = (_ptr(amax), _ptr(gp))
            v = torch.empty((4, C0), dtype=torch.float32, device=dev)
            rc = lib.gridgcn_bn_bwd_finalize(_ptr(sums0), E, C0, _ptr(v[0]), _ptr(v[1]), _ptr(v[2]),
                                             _ptr(v[3]), st)
            _lib.check(rc, "gridgcn_bn_bwd_finalize")
            if noz and OPT.SPARSE_L0 and (Nsrc + 1) * 144 <= 150 * 1024:
                # single-layer point MLP: only the arg-max entries are scattered; the dense
                # BatchNorm terms collapse onto per-source counts and geo_vec sums
                dYsrc = torch.empty((R, C0), dtype=torch.float32, device=dev)
                acc64 = _zeros(3 * C0 + 12, torch.float64, dev)
                wgs, gg = acc64[:3 * C0].view(3, C0), acc64[3 * C0:]
                nbytes = ctypes.c_size_t(0)
                lib.gridgcn_edge_lin0_backward_sparse_workspace_bytes(B, Nsrc, C0,
                                                                      ctypes.byref(nbytes))
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
                if gsum_f is not None:
                    Gsum, gg = gsum_f, gg_f
                    rc = lib.gridgcn_edge_lin0_backward_sparse_geo(
                        _ptr(nebidx), _ptr(att16), _ptr(amax), _ptr(gp), _ptr(zsel[0]), _ptr(Ysrc),
                        _ptr(wgb) if rot else None, _ptr(wgb[3]), _ptr(vec0[0]), _ptr(vec0[1]),
                        _ptr(vec0[2]), _ptr(vec0[3]), _ptr(v[0]), _ptr(v[1]), B, Nsrc, O, P, C0,
                        _ptr(dYsrc), _ptr(Gsum), _ptr(wgs), _ptr(ws), nbytes.value, st)
                else:
                    Gsum = torch.empty((R, 4), dtype=torch.float32, device=dev)
                    rc = lib.gridgcn_edge_lin0_backward_sparse(
                        _ptr(nebidx), _ptr(att16), _ptr(amax), _ptr(gp), _ptr(zsel[0]), _ptr(Ysrc),
                        _ptr(wgb) if rot else None, _ptr(wgb[3]), _ptr(vec0[0]), _ptr(vec0[1]),
                        _ptr(vec0[2]), _ptr(vec0[3]), _ptr(v[0]), _ptr(v[1]), B, Nsrc, O, P, C0,
                        _ptr(dYsrc), _ptr(Gsum), _ptr(wgs), _ptr(gg), _ptr(ws), nbytes.value, st)
                _lib.check(rc, "gridgcn_edge_lin0_backward_sparse")
                dWg = None
                if rot:
                    # geo_vec columns of dW0, written in place by one kernel
                    dW0 = torch.empty((C0, rot + Cf), dtype=torch.float32, device=dev)
                    ytg = _tn_matmul(Ysrc, Gsum)          # (a name: the product must outlive the call that reads it)
                    rc = lib.gridgcn_edge_lin0_dwg(
                        _ptr(wgs), _ptr(gg), _ptr(ytg), _ptr(wgb),
                        _ptr(vec0[0]), _ptr(vec0[2]), _ptr(vec0[3]), _ptr(v[0]), _ptr(v[1]), C0,
                        _ptr(dW0), rot + Cf, st)
                    _lib.check(rc, "gridgcn_edge_lin0_dwg")
            else:
                zb = torch.zeros(R * C0 * 4 + 3 * C0 * 8, dtype=torch.uint8, device=dev)
                dYsrc = zb[:R * C0 * 4].view(torch.float32).view(R, C0)
                dWg = zb[R * C0 * 4:].view(torch.float64).view(3, C0)
                nbytes = ctypes.c_size_t(0)
                lib.gridgcn_take_backward_workspace_bytes(B, Nsrc, O * P, ctypes.byref(nbytes))
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
                rc = lib.gridgcn_edge_lin0_backward(
                    _ptr(Z0) if Z0 is not None else None, _ptr(Ysrc) if noz else None,
                    _ptr(wgb) if (noz and rot) else None, _ptr(wgb[3]) if noz else None,
                    _ptr(dY0) if dY0 is not None else None, sparse0[0], sparse0[1],
                    _ptr(vec0[0]), _ptr(vec0[1]), _ptr(vec0[2]), _ptr(vec0[3]), _ptr(v[0]),
                    _ptr(v[1]), _ptr(att16), _ptr(nebidx), B, Nsrc, O, P, C0, _ptr(dYsrc),
                    _ptr(dWg) if rot else None, _ptr(ws), nbytes.value, st)
                _lib.check(rc, "gridgcn_edge_lin0_backward")
            # the two small GEMMs on the source points
            feat = src.detach()[..., 4:].reshape(R, Cf)
            if rot and dWg is None:
                _tn_matmul(dYsrc, feat, out=dW0[:, rot:])             # [C0, Cf] beside dWg
            elif rot:
                dW0 = torch.empty((C0, rot + Cf), dtype=torch.float32, device=dev)
                _tn_matmul(dYsrc, feat, out=dW0[:, rot:])
                dW0[:, :rot].copy_(dWg.t())                           # (fp64 sums -> the three geo columns)
            else:
                dW0 = _tn_matmul(dYsrc, feat)
            gsrc = None
            if ctx.needs_input_grad[0]:
                # gradient of the source rows [xyz w | features]: the four leading columns are zero
                # rows of the (transposed) weight, so the product IS the full row
                if _small_ok(R, Cf) and C0 % 8 == 0:
                    gsrc = torch.empty((R, Cs), dtype=torch.float32, device=dev)
                    _gemm_small(1, dYsrc, W0.detach()[:, rot:], gsrc[:, 4:], R, Cf, C0, zero_left=4)
                    gsrc = gsrc.view(B, Nsrc, Cs)
                else:
                    gsrc = torch.zeros((R, Cs), dtype=torch.float32, device=dev)
                    _mm_nn(dYsrc, W0.detach()[:, rot:], out=gsrc[:, 4:])
                    gsrc = gsrc.view(B, Nsrc, Cs)
            db0 = _zeros(C0, torch.float32, dev)
        grads0 = [dW0, db0, v[2], v[3]]
        return (gsrc, None, None, None) + tuple(grads0) + tuple(grads_rest) + tuple(grads_a)


def edge_block_src_supported(pt_layers, att_layers, src, has_feats, P=None):
    """the source-side first conv needs neighbour features with a width the kernels can vector-load"""
    if not (has_feats and src.is_cuda and src.dtype == torch.float32 and OPT.SRC_FIRST_CONV):
        return False
    C0 = pt_layers[0].lin.out_features
    if C0 % 4 or C0 > 256 or (src.shape[2] % 4):
        return False
    return edge_block_supported(pt_layers, att_layers, src, P)


def edge_block_src_train(src, nebidx, cent, pt_layers, att_layers, localfdim, out=None):
    """[B,O,C] = max_p att_mlp(att_vec) * pt_mlp(concat(geo_vec, gathered features)) from
    (src [B,Nsrc,4+Cf], nebidx [B,O,P], cent [B,O,>=3]) -- sub_g_update up to the pooling.
    out: optional [B*O, C] destination (alias_columns); the 2-D result is then returned."""
    params = []
    for l in list(pt_layers) + list(att_layers):
        params += [l.lin.weight, l.lin.bias, l.bn.weight, l.bn.bias]
    meta = (pt_layers[0].bn.eps, [l.bn for l in pt_layers], [l.bn for l in att_layers],
            localfdim != 0, out)
    return _EdgeBlockSrcTrain.apply(src, nebidx, cent, meta, *params)


def edge_block_supported(pt_layers, att_layers, nf, P=None):
    """P: neighbours per centre.  The arg max of the neighbour max-pool is stored in ONE byte
    (uint8 amax, four of them per 32-bit store), so the kernels take P <= 256 (include/gridgcn.h);
    wider neighbour lists run on the stock modules."""
    if P is not None and P > 256:
        return False
    C = pt_layers[-1].lin.out_features
    return (supported(pt_layers, nf) and supported(att_layers, nf)
            and att_layers[-1].lin.out_features == C)


def edge_block_train(nf, att_vec, pt_layers, att_layers, rot=0):
    """nf [B,O,P,cin], att_vec [B,O,P,10] -> [B,O,C] = max_p att_mlp(att_vec) * pt_mlp(nf).
    nf / att_vec may come zero padded (and nf with its first `rot` channels moved behind the
    others) from ops.edge_inputs_rows."""
    B, O, P, cin = nf.shape
    params = []
    for l in list(pt_layers) + list(att_layers):
        params += [l.lin.weight, l.lin.bias, l.bn.weight, l.bn.bias]
    meta = (pt_layers[0].bn.eps, [l.bn for l in pt_layers], [l.bn for l in att_layers], B * O, P,
            rot)
    agg = _EdgeBlockTrain.apply(nf.reshape(-1, cin), att_vec.reshape(-1, att_vec.shape[-1]), meta,
                                *params)
    return agg.reshape(B, O, -1)
