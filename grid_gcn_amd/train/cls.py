"""The classification GridConv edge block (classification/models/gcn_module_g.py:64-223) in training mode."""
import ctypes

import torch

from .. import _lib
from ..ops import _ptr, _stream
from .common import (  # noqa: F401
    _chain_backward, _chain_forward, _dw_direct_ok, _gemm_small, _mm_nn, _mm_nt, _momentum, _small_ok,
    _stats_written, _tn_matmul, _zeros, packed_sizes, supported,)

# ------------------------------------------------------------------------------------------------
# classification edge block (classification/models/gcn_module_g.py:64-114 verts_pair_func with
# att_full='next' and the context vector of :212-223)
def _pack_bwd_part(lib, Wpart, stream):
    """backward operand layouts (Wb, Wg, Wdx) of a column slice of a weight, W [C, cin] contiguous"""
    C, cin = Wpart.shape
    _, _, _, nwb = packed_sizes(C, cin)
    nt = (cin + 31) // 32
    nwdx = C * 32 * (1 if nt <= 1 else 2 if nt <= 2 else 4 if nt <= 4 else 8)
    pk = torch.empty(2 * nwb + nwdx, dtype=torch.float32, device=Wpart.device)
    Wb, Wg, Wdx = pk[:nwb], pk[nwb:2 * nwb], pk[2 * nwb:]
    _lib.check(lib.gridgcn_pack_linear(_ptr(Wpart), None, C, cin, 0, cin, cin, None, None, _ptr(Wb),
                                       _ptr(Wg), None, _ptr(Wdx), stream), "gridgcn_pack_linear")
    return Wb, Wg, Wdx


class _EdgeBlockClsTrain(torch.autograd.Function):
    """The classifier's GridConv edge block from (src, nebidx, cent):

        nf0  = geo_vec | gathered features             ctx = max_p nf0  (per centre)
        nf   = pt_mlp(nf0)                             a1  = att1(dist | geo_vec)
        att  = att2(concat(a1, nf, tile(ctx)))         agg = max_p att * nf

    Nothing of the concat exists: the first att2 conv reads the raw outputs of att1 and of the last
    pt conv (their BatchNorm+ReLU applied on load: the two-source kernel
    gridgcn_linear_fwd_direct2) and receives the context term ctx Wc^T + b as a per-centre bias.  Its
    backward is two gridgcn_linear_bwd calls (one per source, each with the matching slice of the
    weight), a segmented sum for the per-centre bias, and the dense gradient it leaves on the last
    pt conv's activation is merged with the sparse one of the product/max before the pt chain's
    backward runs.  The first pt conv of a layer WITH neighbour features runs on the source points
    (as _EdgeBlockSrcTrain), of a layer without on the [E, 8] geo rows."""

    @staticmethod
    def forward(ctx, src, nebidx, cent, meta, *params):
        lib = _lib.load()
        eps, bns_p, bn_a1, bns_a2 = meta
        Lp, La = len(bns_p), len(bns_a2)
        B, Nsrc, Cs = src.shape
        _, O, P = nebidx.shape
        E, R, Cf, ncent = B * O * P, B * Nsrc, Cs - 4, B * O
        has_feats = Cf > 0
        dev = src.device
        pp, pa1, pa2 = params[:4 * Lp], params[4 * Lp:4 * Lp + 4], params[4 * Lp + 4:]
        with torch.cuda.device(dev):
            st = _stream(src)
            srcd = src.detach()
            ctxv = torch.empty((ncent, 3 + Cf), dtype=torch.float32, device=dev)
            cidx = torch.empty((ncent, max(Cf, 1)), dtype=torch.int32, device=dev)
            _lib.check(lib.gridgcn_ctx_max(_ptr(srcd), _ptr(nebidx), _ptr(cent), cent.shape[2], B,
                                           Nsrc, Cs, O, P, _ptr(ctxv), _ptr(cidx), st),
                       "gridgcn_ctx_max")
            att16 = torch.empty((E, 16), dtype=torch.float32, device=dev)
            if has_feats:
                W0, b0, g0, be0 = pp[:4]
                C0 = W0.shape[0]
                feat = srcd[..., 4:].reshape(R, Cf)
                if _small_ok(R, C0) and Cf % 8 == 0 and Cf <= 512:
                    Ysrc = _gemm_small(0, feat, W0.detach()[:, 3:],
                                       torch.empty((R, C0), dtype=torch.float32, device=dev), R, C0, Cf)
                else:
                    Ysrc = _mm_nt(feat, W0.detach()[:, 3:])
                wgb = torch.cat([W0.detach()[:, :3].t(), b0.detach()[None]])
                x0 = torch.empty((E, C0), dtype=torch.float32, device=dev)          # Z0
                sums0 = _zeros(2 * C0, torch.float64, dev)
                rc = lib.gridgcn_edge_lin0_forward(
                    _ptr(Ysrc), _ptr(srcd), _ptr(nebidx), _ptr(cent), cent.shape[2], B, Nsrc, Cs, O,
                    P, C0, _ptr(wgb), _ptr(wgb[3]), _ptr(x0), _ptr(att16), _ptr(sums0), st)
                _lib.check(rc, "gridgcn_edge_lin0_forward")
                vec0 = torch.empty((4, C0), dtype=torch.float32, device=dev)
                bn = bns_p[0]
                track = bn.track_running_stats
                rc = lib.gridgcn_bn_finalize(
                    _ptr(sums0), _ptr(g0.detach()), _ptr(be0.detach()), E, eps,
                    _momentum(bn) if track else 0.0, C0, _ptr(vec0[0]), _ptr(vec0[1]),
                    _ptr(vec0[2]), _ptr(vec0[3]), _ptr(bn.running_mean) if track else None,
                    _ptr(bn.running_var) if track else None,
                    _ptr(bn.num_batches_tracked) if track else None, st)
                _lib.check(rc, "gridgcn_bn_finalize")
                if track:
                    _stats_written(bn)
                sp = _chain_forward(lib, x0, pp[4:], bns_p[1:], eps, 0, C0,
                                    prev_bn=(vec0[0], vec0[1]))
            else:
                x0 = torch.empty((E, 8), dtype=torch.float32, device=dev)           # geo_vec | 0
                rc = lib.gridgcn_edge_inputs_rows(_ptr(srcd), _ptr(nebidx), _ptr(cent),
                                                  cent.shape[2], B, Nsrc, Cs, O, P, 0, 0, 8,
                                                  _ptr(x0), _ptr(att16), st)
                _lib.check(rc, "gridgcn_edge_inputs_rows")
                vec0 = wgb = W0 = None
                sp = _chain_forward(lib, x0, pp, bns_p, eps, 0, 0)
            Zl, scl, shl = sp.Z[-1], sp.scale[-1], sp.shift[-1]
            C = Zl.shape[1]
            s1 = _chain_forward(lib, att16, pa1, [bn_a1], eps)
            Za1 = s1.Z[0]
            A0 = Za1.shape[1]
            # first att2 conv: two row sources + the context term as a per-centre bias
            W2, b2, g2, be2 = pa2[:4]
            W2d = W2.detach()
            N0, K12 = W2d.shape[0], A0 + C
            rowb = _mm_nt(ctxv, W2d[:, K12:], bias=b2)                              # [ncent, N0]
            _, ldw, _, _ = packed_sizes(N0, K12)
            Wq = torch.empty(K12 * ldw, dtype=torch.float32, device=dev)
            W12 = W2d[:, :K12].contiguous()       # (a name: the copy must outlive the call that reads it)
            _lib.check(lib.gridgcn_pack_linear(_ptr(W12), None, N0, K12, 0,
                                               K12, 0, None, None, None, None, _ptr(Wq), None, st),
                       "gridgcn_pack_linear")
            part1 = _pack_bwd_part(lib, W2d[:, :A0].contiguous(), st)
            part2 = _pack_bwd_part(lib, W2d[:, A0:K12].contiguous(), st)
            Z20 = torch.empty((E, N0), dtype=torch.float32, device=dev)
            sums20 = _zeros(2 * N0, torch.float64, dev)
            psc = torch.cat([s1.scale[0], scl])
            psh = torch.cat([s1.shift[0], shl])
            rc = lib.gridgcn_linear_fwd_direct2(_ptr(Za1), A0, A0, _ptr(Zl), C, C, E, _ptr(Wq), None,
                                                _ptr(rowb), P, ldw, N0, _ptr(psc), _ptr(psh),
                                                _ptr(Z20), _ptr(sums20), st)
            _lib.check(rc, "gridgcn_linear_fwd_direct2")
            vecA = torch.empty((4, N0), dtype=torch.float32, device=dev)
            bn = bns_a2[0]
            track = bn.track_running_stats
            rc = lib.gridgcn_bn_finalize(
                _ptr(sums20), _ptr(g2.detach()), _ptr(be2.detach()), E, eps,
                _momentum(bn) if track else 0.0, N0, _ptr(vecA[0]), _ptr(vecA[1]), _ptr(vecA[2]),
                _ptr(vecA[3]), _ptr(bn.running_mean) if track else None,
                _ptr(bn.running_var) if track else None,
                _ptr(bn.num_batches_tracked) if track else None, st)
            _lib.check(rc, "gridgcn_bn_finalize")
            if track:
                _stats_written(bn)
            sa = _chain_forward(lib, Z20, pa2[4:], bns_a2[1:], eps, 0, N0,
                                prev_bn=(vecA[0], vecA[1]))
            agg = torch.empty((ncent, C), dtype=torch.float32, device=dev)
            amax = torch.empty((ncent, C), dtype=torch.uint8, device=dev)
            zsel = torch.empty((2, ncent, C), dtype=torch.float32, device=dev)
            rc = lib.gridgcn_pairmax_fwd(_ptr(Zl), _ptr(sa.Z[-1]), _ptr(scl), _ptr(shl),
                                         _ptr(sa.scale[-1]), _ptr(sa.shift[-1]), ncent, P, C,
                                         _ptr(agg), C, _ptr(amax), _ptr(zsel), st)
            _lib.check(rc, "gridgcn_pairmax_fwd")
        ctx.dims = (Lp, La, B, Nsrc, Cs, O, P, A0, N0, has_feats, len(sp.Z))
        ctx.ndx = (sp.ndx, sa.ndx)
        none = torch.empty(0, device=dev)
        ctx.save_for_backward(
            src, nebidx, att16, amax, zsel, ctxv, cidx, x0, Z20, vecA, W2,
            vec0 if has_feats else none, wgb if has_feats else none, W0 if has_feats else none,
            *part1, *part2, Za1, s1.scale[0], s1.shift[0], s1.mean[0], s1.rstd[0], s1.Wb[0],
            s1.Wg[0], s1.Wdx[0],
            *sp.Z, *sp.scale, *sp.shift, *sp.mean, *sp.rstd, *sp.Wb, *sp.Wg, *sp.Wdx,
            *sa.Z, *sa.scale, *sa.shift, *sa.mean, *sa.rstd, *sa.Wb, *sa.Wg, *sa.Wdx)
        ctx.mark_non_differentiable(amax)
        return agg.reshape(B, O, C)

    @staticmethod
    def backward(ctx, dagg):
        lib = _lib.load()
        Lp, La, B, Nsrc, Cs, O, P, A0, N0, has_feats, nsp = ctx.dims
        t = ctx.saved_tensors
        src, nebidx, att16, amax, zsel, ctxv, cidx, x0, Z20, vecA, W2, vec0, wgb, W0 = t[:14]
        part1, part2 = t[14:17], t[17:20]
        Za1, a1S, a1H, a1M, a1R, a1Wb, a1Wg, a1Wx = t[20:28]
        o = 28
        pZ, pS, pH, pM, pR, pWb, pWg, pWx = (t[o + k * nsp:o + (k + 1) * nsp] for k in range(8))
        o += 8 * nsp
        L2 = La - 1
        aZ, aS, aH, aM, aR, aWb, aWg, aWx = (t[o + k * L2:o + (k + 1) * L2] for k in range(8))
        dev = src.device
        E, R, Cf, ncent = B * O * P, B * Nsrc, Cs - 4, B * O
        Zl = pZ[-1]
        C = Zl.shape[1]
        K12 = A0 + C
        dagg = dagg.contiguous().reshape(ncent, C)
        with torch.cuda.device(dev):
            st = _stream(src)
            gp = torch.empty((ncent, C), dtype=torch.float32, device=dev)
            ga = torch.empty((ncent, C), dtype=torch.float32, device=dev)
            sums_pa = _zeros((2, 2 * C), torch.float64, dev)
            sums_p, sums_a = sums_pa[0], sums_pa[1]
            rc = lib.gridgcn_pairmax_bwd(_ptr(Zl), _ptr(aZ[-1]), _ptr(pS[-1]), _ptr(pH[-1]),
                                         _ptr(pM[-1]), _ptr(pR[-1]), _ptr(aS[-1]), _ptr(aH[-1]),
                                         _ptr(aM[-1]), _ptr(aR[-1]), _ptr(dagg), _ptr(amax), ncent,
                                         P, C, C, _ptr(gp), _ptr(ga), _ptr(sums_p), _ptr(sums_a),
                                         _ptr(zsel), st)
            _lib.check(rc, "gridgcn_pairmax_bwd")
            # att2 layers 1..: leaves dA = gradient of relu(bn(Z20)) and Z20's BatchNorm sums
            dA, grads_a2, sums20 = _chain_backward(
                lib, Z20, aZ, aS, aH, aM, aR, aWb, aWg, aWx, ctx.ndx[1], sums_a, None,
                (amax, ga, P), True, None, 0, prev_bn=(vecA[0], vecA[1], vecA[2], vecA[3]))
            v = torch.empty((4, N0), dtype=torch.float32, device=dev)
            rc = lib.gridgcn_bn_bwd_finalize(_ptr(sums20), E, N0, _ptr(v[0]), _ptr(v[1]), _ptr(v[2]),
                                             _ptr(v[3]), st)
            _lib.check(rc, "gridgcn_bn_bwd_finalize")

            def part_bwd(prev, pbn, pk, cin):
                dX = torch.empty((E, cin), dtype=torch.float32, device=dev)
                dW = torch.empty((N0, cin), dtype=torch.float32, device=dev)
                ps = _zeros(2 * cin, torch.float64, dev)
                nbytes = ctypes.c_size_t(0)
                lib.gridgcn_linear_bwd_workspace_bytes(E, cin, N0, ctypes.byref(nbytes))
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
                rc = lib.gridgcn_linear_bwd(
                    _ptr(dA), _ptr(Z20), _ptr(vecA[0]), _ptr(vecA[1]), _ptr(vecA[2]), _ptr(vecA[3]),
                    _ptr(v[0]), _ptr(v[1]), _ptr(prev), _ptr(pbn[0]), _ptr(pbn[1]), _ptr(pbn[2]),
                    _ptr(pbn[3]), _ptr(pk[0]), _ptr(pk[1]), _ptr(pk[2]), cin, E, N0, cin, cin, 0,
                    dA.stride(0), _ptr(dX), _ptr(dW), _ptr(ps), None, None, 0, _ptr(ws),
                    nbytes.value, st)
                _lib.check(rc, "gridgcn_linear_bwd")
                return dX, dW, ps

            dX1, dW1, ps1 = part_bwd(Za1, (a1S, a1H, a1M, a1R), part1, A0)
            dX2, dW2, ps2 = part_bwd(Zl, (pS[-1], pH[-1], pM[-1], pR[-1]), part2, C)
            # per-centre bias: context term
            dcb = torch.empty((ncent, N0), dtype=torch.float32, device=dev)
            rc = lib.gridgcn_bn_dz_segsum(_ptr(dA), _ptr(Z20), _ptr(vecA[0]), _ptr(vecA[1]),
                                          _ptr(vecA[2]), _ptr(vecA[3]), _ptr(v[0]), _ptr(v[1]),
                                          ncent, P, N0, _ptr(dcb), st)
            _lib.check(rc, "gridgcn_bn_dz_segsum")
            Wc = W2.detach()[:, K12:]
            dW20 = torch.cat([dW1, dW2, _tn_matmul(dcb, ctxv)], dim=1)
            grads_a20 = [dW20, _zeros(N0, torch.float32, dev), v[2], v[3]]
            # last pt conv: dense gradient through att2 + sparse gradient of the product/max
            _lib.check(lib.gridgcn_sparse_add(_ptr(amax), _ptr(gp), ncent, P, C, _ptr(dX2), st),
                       "gridgcn_sparse_add")
            sums_l = sums_p + ps2
            _, grads_a1 = _chain_backward(lib, att16, [Za1], [a1S], [a1H], [a1M], [a1R], [a1Wb],
                                          [a1Wg], [a1Wx], [0], ps1, dX1, None, False, 4, 0)
            gsrc = None
            if not has_feats:
                _, grads_p = _chain_backward(lib, x0, pZ, pS, pH, pM, pR, pWb, pWg, pWx, ctx.ndx[0],
                                             sums_l, dX2, None, False, 3, 0)
            else:
                C0 = x0.shape[1]
                dY0, grads_rest, sums0 = _chain_backward(
                    lib, x0, pZ, pS, pH, pM, pR, pWb, pWg, pWx, ctx.ndx[0], sums_l, dX2, None, True,
                    None, 0, prev_bn=(vec0[0], vec0[1], vec0[2], vec0[3]))
                v0 = torch.empty((4, C0), dtype=torch.float32, device=dev)
                rc = lib.gridgcn_bn_bwd_finalize(_ptr(sums0), E, C0, _ptr(v0[0]), _ptr(v0[1]),
                                                 _ptr(v0[2]), _ptr(v0[3]), st)
                _lib.check(rc, "gridgcn_bn_bwd_finalize")
                zb = torch.zeros(R * C0 * 4 + 3 * C0 * 8, dtype=torch.uint8, device=dev)
                dYsrc = zb[:R * C0 * 4].view(torch.float32).view(R, C0)
                dWg = zb[R * C0 * 4:].view(torch.float64).view(3, C0)
                nbytes = ctypes.c_size_t(0)
                lib.gridgcn_take_backward_workspace_bytes(B, Nsrc, O * P, ctypes.byref(nbytes))
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
                rc = lib.gridgcn_edge_lin0_backward(
                    _ptr(x0), None, None, None, _ptr(dY0), None, None, _ptr(vec0[0]), _ptr(vec0[1]),
                    _ptr(vec0[2]), _ptr(vec0[3]), _ptr(v0[0]), _ptr(v0[1]), _ptr(att16),
                    _ptr(nebidx), B, Nsrc, O, P, C0, _ptr(dYsrc), _ptr(dWg), _ptr(ws), nbytes.value,
                    st)
                _lib.check(rc, "gridgcn_edge_lin0_backward")
                feat = src.detach()[..., 4:].reshape(R, Cf)
                dW0 = torch.cat([dWg.t().float(), _tn_matmul(dYsrc, feat)], dim=1)
                grads_p = [dW0, _zeros(C0, torch.float32, dev), v0[2], v0[3]] + list(grads_rest)
                if ctx.needs_input_grad[0]:
                    if _small_ok(R, Cf) and C0 % 8 == 0:
                        gsrc = torch.empty((R, Cs), dtype=torch.float32, device=dev)
                        _gemm_small(1, dYsrc, W0.detach()[:, 3:], gsrc[:, 4:], R, Cf, C0, zero_left=4)
                        gsrc = gsrc.view(B, Nsrc, Cs)
                    else:
                        gsrc = torch.zeros((R, Cs), dtype=torch.float32, device=dev)
                        _mm_nn(dYsrc, W0.detach()[:, 3:], out=gsrc[:, 4:])
                        gsrc = gsrc.view(B, Nsrc, Cs)
                    # the context vector's arg-max rows
                    if dcb.stride(1) == 1 and Wc.stride(1) == 1 and dcb.shape[1] % 8 == 0 and \
                            _small_ok(dcb.shape[0], Wc.shape[1], dcb.shape[1]):
                        dctx = _gemm_small(1, dcb, Wc, torch.empty((dcb.shape[0], Wc.shape[1]), dtype=torch.float32,
                                                                  device=dev), dcb.shape[0], Wc.shape[1], dcb.shape[1])
                    else:
                        dctx = _mm_nn(dcb, Wc)
                    _lib.check(lib.gridgcn_ctx_max_backward(_ptr(dctx), _ptr(cidx), ncent, Cf, Cs,
                                                            _ptr(gsrc), st),
                               "gridgcn_ctx_max_backward")
        return (gsrc, None, None, None) + tuple(grads_p) + tuple(grads_a1) + tuple(grads_a20) + \
            tuple(grads_a2)


def edge_block_cls_supported(pt_layers, att1_layers, att2_layers, src, P):
    """shapes _EdgeBlockClsTrain takes: every stack within the MFMA kernels' widths, att1 output a
    multiple of 32 (the two-source kernel switches sources on a 32-column chunk), whole 32-row
    tiles per centre, neighbour features (if any) vector-loadable"""
    if not (src.is_cuda and src.dtype == torch.float32) or len(att1_layers) != 1:
        return False
    if len(att2_layers) < 2 or len(pt_layers) < 2 or P % 32 or P > 256:
        return False
    Cf = src.shape[2] - 4
    if Cf < 0 or Cf % 4 or (Cf > 0 and pt_layers[0].lin.in_features != 3 + Cf) or \
            (Cf == 0 and pt_layers[0].lin.in_features != 3):
        return False
    A0, C = att1_layers[0].lin.out_features, pt_layers[-1].lin.out_features
    N0 = att2_layers[0].lin.out_features
    if A0 % 32 or C % 8 or att2_layers[-1].lin.out_features != C or N0 % 8:
        return False
    if att2_layers[0].lin.in_features != A0 + C + 3 + Cf or A0 + C > 1024:
        return False
    if not (_dw_direct_ok(N0, A0) and _dw_direct_ok(N0, C)):
        return False
    l0 = att2_layers[0]
    if l0.bn is None or not l0.use_relu or N0 > 256 or 256 % N0:
        return False
    return (supported(pt_layers, src) and supported(att1_layers, src)
            and supported(att2_layers[1:], src))


def edge_block_cls_train(src, nebidx, cent, pt_layers, att1_layers, att2_layers):
    """[B,O,C] = max_p att2(concat(att1(att_vec), pt_mlp(nf0), ctx)) * pt_mlp(nf0) from
    (src [B,Nsrc,4+Cf], nebidx [B,O,P], cent [B,O,>=3]); training mode."""
    params = []
    for l in list(pt_layers) + list(att1_layers) + list(att2_layers):
        params += [l.lin.weight, l.lin.bias, l.bn.weight, l.bn.bias]
    meta = (pt_layers[0].bn.eps, [l.bn for l in pt_layers], att1_layers[0].bn,
            [l.bn for l in att2_layers])
    return _EdgeBlockClsTrain.apply(src.contiguous(), nebidx, cent.contiguous(), meta, *params)
