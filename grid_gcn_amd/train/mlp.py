"""Per-point conv + BatchNorm + ReLU stacks in training mode (centre / update MLPs, fc1; wide layers in
256-column slices): autograd Functions over train/common.py's chain kernels."""
import ctypes

import torch

from .. import _lib
from ..ops import _ptr, _stream
from .options import OPT
from .common import (  # noqa: F401
    PACKS, _cached_zeros, _chain_backward, _chain_forward, _dw_direct_ok, _identity_consts, _mm_nn, _mm_nt,
    _momentum, _stats_written, _tn_matmul, _zeros, packed_sizes, supported,)

class _MLPTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, meta, *params):
        """x [E,cin]; params = (W, b, gamma, beta) per layer; meta = (eps, [bn modules], out, link,
        prev): out = None or a [E, C_last] tensor (row stride >= C_last) that receives the result;
        link (RawLink, with out): the result is the RAW output of the last layer -- its BatchNorm+ReLU
        is the consumer's business; prev (RawLink): x is such a buffer."""
        lib = _lib.load()
        eps, bns, out, link, prev = (tuple(meta) + (None, None))[:5]
        L = len(params) // 4
        x = x.contiguous()
        E, cin0 = x.shape
        if OPT.DIRECT_FWD and cin0 % 8 and prev is None:
            # rows padded with zero columns to a multiple of 8 floats: the register-direct kernels
            # then take the layer (a [E, 4] or [E, 4 + C] centre tensor of the up path; the LDS-staged
            # forward kernel these widths used to fall to ran at 4 % of the MFMA rate)
            x = torch.nn.functional.pad(x, (0, -cin0 % 8))        # (one launch)
        with torch.cuda.device(x.device):
            st = _chain_forward(lib, x, params, bns, eps, 0,
                                cin0 if ctx.needs_input_grad[0] else 0,
                                prev_bn=prev.prev_bn()[:2] if prev is not None else None,
                                out_raw=out if link is not None else None,
                                last_vec=link.vec if link is not None else None)
            if link is not None:
                Y = out
            else:
                Y = out if out is not None else torch.empty_like(st.Z[-1])
                rc = lib.gridgcn_bn_relu_apply(_ptr(st.Z[-1]), _ptr(st.scale[-1]), _ptr(st.shift[-1]),
                                               _ptr(Y), E, Y.shape[1], Y.stride(0), _stream(x))
                _lib.check(rc, "gridgcn_bn_relu_apply")
        ctx.link, ctx.prev = link, prev
        ctx.L = L
        ctx.ndx = st.ndx
        ctx.cin_w0 = params[0].shape[1]          # x may carry zero-padded columns beyond it
        ctx.cin0 = cin0
        ctx.save_for_backward(x, *st.Z, *st.scale, *st.shift, *st.mean, *st.rstd, *st.Wb, *st.Wg,
                              *st.Wdx)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.load()
        L = ctx.L
        t = ctx.saved_tensors
        x = t[0]
        Zs, scales, shifts = t[1:1 + L], t[1 + L:1 + 2 * L], t[1 + 2 * L:1 + 3 * L]
        means, rstds, Wbs = t[1 + 3 * L:1 + 4 * L], t[1 + 4 * L:1 + 5 * L], t[1 + 5 * L:1 + 6 * L]
        Wgs, Wdxs = t[1 + 6 * L:1 + 7 * L], t[1 + 7 * L:1 + 8 * L]
        E, dev = x.shape[0], x.device
        C = Zs[-1].shape[1]
        cin_last = Zs[-2].shape[1] if L > 1 else x.shape[1]
        need_dx_last = L > 1 or ctx.needs_input_grad[0]
        # a row-strided gradient (one half of a concat's gradient) is consumed in place when the
        # register-direct kernels take this layer; otherwise it is packed first
        if not (dY.dim() == 2 and dY.stride(1) == 1 and dY.stride(0) % 4 == 0
                and dY.storage_offset() % 4 == 0 and _dw_direct_ok(C, cin_last)
                and OPT.DIRECT_DX
                and (not need_dx_last or ctx.ndx[-1] > 0)):
            dY = dY.contiguous()
        link, prev = ctx.link, ctx.prev
        with torch.cuda.device(dev):
            if link is not None and link.sums is not None:
                # the consumer's input-gradient kernel has accumulated this layer's sums
                sums = link.sums[:, :C].contiguous()
            else:
                assert link is None, "RawLink: the consumer's backward has not run"
                sums = _zeros((2, C), torch.float64, dev)
                rc = lib.gridgcn_bn_relu_bwd_reduce(_ptr(dY), _ptr(Zs[-1]), _ptr(scales[-1]),
                                                    _ptr(shifts[-1]), _ptr(means[-1]), _ptr(rstds[-1]),
                                                    E, C, dY.stride(0), _ptr(sums), _stream(x))
                _lib.check(rc, "gridgcn_bn_relu_bwd_reduce")
            r = _chain_backward(lib, x, Zs, scales, shifts, means, rstds, Wbs, Wgs, Wdxs,
                                ctx.ndx, sums, dY, None, ctx.needs_input_grad[0],
                                ctx.cin_w0, 0, prev_bn=prev.prev_bn() if prev is not None else None,
                                nbn=prev.nbn() if prev is not None else 0)
            dX, grads = r[0], r[1]
            if prev is not None:
                prev.sums = prev.take_sums(r[2], x.shape[1])
            if dX is not None and ctx.cin0 != x.shape[1]:
                dX = dX[:, :ctx.cin0]
        return (dX, None) + tuple(grads)


def _wide_direct_ok(E, cin, C):
    """a layer of more than 256 output channels the register-direct kernels can take as 256-column slices"""
    return (OPT.DIRECT_FWD and OPT.DIRECT_DX and E >= 4096 and cin % 8 == 0 and cin <= 320 and C % 256 == 0
            and 256 < C <= 1024 and _dw_direct_ok(256, cin))


def _pack_tmp(lib, W, b, cout, cin, st):
    """operand layouts of a temporary weight (a slice, a transpose): packed per call, never cached"""
    K, ldw, nwp, nwb = packed_sizes(cout, cin)
    bufs = PACKS.get(lib, W, b, cout, cin, 0, cin, 0, True, (nwp, ldw, nwb, cin * ldw, 0), st)
    return ldw, bufs


class _WideLayerTrain(torch.autograd.Function):
    """conv + BatchNorm(batch statistics) + ReLU of a layer BEYOND the MFMA kernels' widths (> 256 output or
    > 384 input channels: the last layer of the classifier and of the 200k-point workload).  No rocBLAS:
      * large layers (>= 4096 rows, output a multiple of 256, input <= 320 columns): the register-direct kernels
        on 256-column SLICES -- forward per output slice (statistics in its epilogue), dW per output slice of the
        elementwise-formed dZ (identity BatchNorm constants), dX = dZ W as a plain forward product with W^T;
      * anything else (a handful of rows, inputs of 512 / 1027 columns): csrc/gridgcn_gemm.hip (_mm_nt / _mm_nn /
        _tn_matmul) + this library's BatchNorm kernels."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, bn):
        lib = _lib.load()
        x = x.contiguous()
        E, C = x.shape[0], W.shape[0]
        cin = x.shape[1]
        dev = x.device
        direct = _wide_direct_ok(E, cin, C)
        with torch.cuda.device(dev):
            st = _stream(x)
            vec = torch.empty((4, C), dtype=torch.float32, device=dev)
            track = bn.track_running_stats
            if direct:
                Z = torch.empty((E, C), dtype=torch.float32, device=dev)
                allsums = _zeros(2 * C, torch.float64, dev)
                Wbs = []
                for h in range(C // 256):
                    sl = slice(h * 256, (h + 1) * 256)
                    ldw, bufs = _pack_tmp(lib, W.detach()[sl], b.detach()[sl], 256, cin, st)
                    Wbs.append(bufs[2])
                    sums = allsums[h * 512:(h + 1) * 512]
                    rc = lib.gridgcn_linear_fwd_direct_ld(_ptr(x), E, cin, cin, _ptr(bufs[4]), _ptr(bufs[1]), ldw,
                                                          256, None, None, _ptr(Z[:, sl]), _ptr(sums), C, 0, st)
                    _lib.check(rc, "gridgcn_linear_fwd_direct")
                    rc = lib.gridgcn_bn_finalize(
                        _ptr(sums), _ptr(gamma.detach()[sl]), _ptr(beta.detach()[sl]), E, bn.eps,
                        _momentum(bn) if track else 0.0, 256, _ptr(vec[0][sl]), _ptr(vec[1][sl]), _ptr(vec[2][sl]),
                        _ptr(vec[3][sl]), _ptr(bn.running_mean[sl]) if track else None,
                        _ptr(bn.running_var[sl]) if track else None,
                        _ptr(bn.num_batches_tracked) if (track and h == 0) else None, st)
                    _lib.check(rc, "gridgcn_bn_finalize")
                    if track:
                        _stats_written(bn)
                saved_w = Wbs
            else:
                Z = _mm_nt(x.detach(), W.detach(), bias=b)
                sums = _zeros(2 * C, torch.float64, dev)
                _lib.check(lib.gridgcn_bn_stats(_ptr(Z), E, C, C, _ptr(sums), st), "gridgcn_bn_stats")
                rc = lib.gridgcn_bn_finalize(
                    _ptr(sums), _ptr(gamma.detach()), _ptr(beta.detach()), E, bn.eps,
                    _momentum(bn) if track else 0.0, C, _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]),
                    _ptr(vec[3]), _ptr(bn.running_mean) if track else None,
                    _ptr(bn.running_var) if track else None,
                    _ptr(bn.num_batches_tracked) if track else None, st)
                _lib.check(rc, "gridgcn_bn_finalize")
                if track:
                    _stats_written(bn)
                saved_w = []
            Y = torch.empty_like(Z)
            rc = lib.gridgcn_bn_relu_apply(_ptr(Z), _ptr(vec[0]), _ptr(vec[1]), _ptr(Y), E, C, C, st)
            _lib.check(rc, "gridgcn_bn_relu_apply")
        ctx.direct = direct
        ctx.save_for_backward(x, W, Z, vec, *saved_w)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.load()
        x, W, Z, vec = ctx.saved_tensors[:4]
        Wbs = ctx.saved_tensors[4:]
        E, C = Z.shape
        cin = x.shape[1]
        dev = x.device
        dY = dY.contiguous()
        with torch.cuda.device(dev):
            st = _stream(x)
            sums = _zeros(2 * C, torch.float64, dev)
            rc = lib.gridgcn_bn_relu_bwd_reduce(_ptr(dY), _ptr(Z), _ptr(vec[0]), _ptr(vec[1]),
                                                _ptr(vec[2]), _ptr(vec[3]), E, C, C, _ptr(sums), st)
            _lib.check(rc, "gridgcn_bn_relu_bwd_reduce")
            v = torch.empty((4, C), dtype=torch.float32, device=dev)
            rc = lib.gridgcn_bn_bwd_finalize(_ptr(sums), E, C, _ptr(v[0]), _ptr(v[1]), _ptr(v[2]),
                                             _ptr(v[3]), st)
            _lib.check(rc, "gridgcn_bn_bwd_finalize")
            dZ = torch.empty_like(Z)
            rc = lib.gridgcn_bn_relu_bwd_elemt(_ptr(dY), _ptr(Z), _ptr(vec[0]), _ptr(vec[1]),
                                               _ptr(vec[2]), _ptr(vec[3]), _ptr(v[0]), _ptr(v[1]),
                                               E, C, _ptr(dZ), st)
            _lib.check(rc, "gridgcn_bn_relu_bwd_elemt")
            if ctx.direct:
                dX = None
                if ctx.needs_input_grad[0]:
                    # dX = dZ W: a plain forward product over K = C with the rows of W^T as "output channels"
                    dX = torch.empty((E, cin), dtype=torch.float32, device=dev)
                    Wt = W.detach().t()
                    for c0 in range(0, cin, 256):
                        n = min(256, cin - c0)
                        ldw, bufs = _pack_tmp(lib, Wt[c0:c0 + n].contiguous(), _cached_zeros(n, dev), n, C, st)
                        rc = lib.gridgcn_linear_fwd_direct_ld(_ptr(dZ), E, C, C, _ptr(bufs[4]), _ptr(bufs[1]), ldw, n,
                                                              None, None, _ptr(dX[:, c0:]), None, cin, 0, st)
                        _lib.check(rc, "gridgcn_linear_fwd_direct")
                # dW = dZ^T x per 256-channel slice of dZ: the register-direct dW kernel with identity BatchNorm
                # constants (dz = dy), the "pre-activation" it asks for being dZ itself (never used: shift = inf)
                dW = torch.empty((C, cin), dtype=torch.float32, device=dev)
                ident = _identity_consts(256, dev)
                nbytes = ctypes.c_size_t(0)
                lib.gridgcn_linear_bwd_workspace_bytes(E, cin, 256, ctypes.byref(nbytes))
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
                for h in range(C // 256):
                    sl = slice(h * 256, (h + 1) * 256)
                    rc = lib.gridgcn_linear_bwd_ld(
                        _ptr(dZ[:, sl]), _ptr(dZ[:, sl]), _ptr(ident[0]), _ptr(ident[1]), _ptr(ident[2]),
                        _ptr(ident[3]), _ptr(ident[4]), _ptr(ident[5]), _ptr(x), None, None, None, None,
                        _ptr(Wbs[h]), None, None, 0, E, 256, cin, cin, 0, C, C, 0, 0, None, _ptr(dW[sl]), None,
                        None, None, 0, _ptr(ws), nbytes.value, st)
                    _lib.check(rc, "gridgcn_linear_bwd")
            else:
                dX = _mm_nn(dZ, W.detach()) if ctx.needs_input_grad[0] else None
                dW = _tn_matmul(dZ, x.detach())
            db = _zeros(C, torch.float32, dev)      # bias in front of a BatchNorm: sum(dZ) == 0
        return dX, dW, db, v[2], v[3], None


def wide_supported(layers, x):
    """any stack of conv + BatchNorm + ReLU on fp32 GPU rows (the fallback behind supported())"""
    # (the BatchNorm kernels: a divisor of 256 or a multiple of 256 channels)
    return (x.is_cuda and x.dtype == torch.float32 and
            all(l.bn is not None and l.use_relu and
                (l.lin.out_features % 256 == 0 or 256 % l.lin.out_features == 0) for l in layers))


def _padded_supported(layer, x):
    """a single layer whose input, zero-padded to a multiple of 8 columns, fits the MFMA kernels"""
    c, cin8 = layer.lin.out_features, (layer.lin.in_features + 7) & ~7
    return (layer.bn is not None and layer.use_relu and c <= 256 and 256 % c == 0 and cin8 <= 320
            and _dw_direct_ok(c, cin8) and c % 8 == 0)


def mlp_wide_train(x, layers):
    """x [..., cin] through `layers` in training mode: wide layers on _WideLayerTrain (256-column slices of the
    register-direct kernels or gridgcn_gemm + this library's BatchNorm kernels; no BLAS library); layers the
    MFMA kernels take still go through them."""
    shp = x.shape
    y = x.reshape(-1, shp[-1])
    i = 0
    while i < len(layers):
        # longest run of layers the MFMA chain takes, else one wide layer
        j = i
        while j < len(layers) and supported(layers[i:j + 1], y):
            j += 1
        if j > i:
            y = mlp_bn_relu_train(y, layers[i:j])
            i = j
        elif y.shape[1] % 8 and supported(layers[i:i + 1], y) is False and \
                _padded_supported(layers[i], y):
            # e.g. 259 -> 256: zero-padded to 264 columns the register-direct kernels take it
            y = torch.nn.functional.pad(y, (0, 8 - y.shape[1] % 8))
            y = mlp_bn_relu_train(y, layers[i:i + 1])
            i += 1
        else:
            l = layers[i]
            y = _WideLayerTrain.apply(y, l.lin.weight, l.lin.bias, l.bn.weight, l.bn.bias, l.bn)
            i += 1
    return y.reshape(shp[:-1] + (y.shape[-1],))


def mlp_bn_relu_train(x, layers, out=None, link=None, prev=None):
    """x [..., cin] -> [..., cout_last] through `layers` (gridconv.ConvBNReLU modules, training
    mode).  Callers check supported() first.  out: optional [E, cout_last] destination
    (alias_columns); the 2-D result is then returned as is.  link / prev: RawLink roles (producer of
    a raw output into `out` / consumer of such a buffer)."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    params = []
    for l in layers:
        params += [l.lin.weight, l.lin.bias, l.bn.weight, l.bn.bias]
    y = _MLPTrain.apply(x2, (layers[0].bn.eps, [l.bn for l in layers], out, link, prev), *params)
    if out is not None:
        return y
    return y.reshape(shp[:-1] + (y.shape[-1],))
