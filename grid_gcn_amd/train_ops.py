"""Training-mode per-edge MLPs on the hand-written gfx950 kernels (csrc/gridgcn_train.hip).

One autograd Function covers a whole stack of (1x1 conv -> BatchNorm(batch statistics) -> ReLU)
layers (mlp2d_c / mlp1d_c, utils/ops.py:236-260):

  forward   per layer ONE kernel: Z_l = act_{l-1} * W_l + b_l on fp32 MFMA, where act_{l-1} =
            relu(bn(Z_{l-1})) is applied while the tile is staged (never materialised) and the
            kernel's epilogue accumulates the batch statistics of Z_l; only the last layer's
            activation is materialised.
  backward  BatchNorm+ReLU backward in two passes per layer (reduce, element-wise); the two GEMMs
            of each layer (dW = act^T dZ, dX = dZ W^T) go through rocBLAS for now.

Numerics follow torch.nn.BatchNorm1d(eps, momentum) exactly as used by gridconv.ConvBNReLU (biased
variance for normalisation, unbiased for the running estimate).
"""
import ctypes

import torch

from . import _lib
from .ops import _ptr, _stream, pack_conv_layer


def supported(layers, x):
    if not (x.is_cuda and x.dtype == torch.float32):
        return False
    for l in layers:
        c = l.lin.out_features
        if l.bn is None or not l.use_relu or c > 256 or 256 % c != 0:
            return False
        if l.lin.in_features > 384:          # <= 12 column tiles in gg_k_linear_bwd
            return False
    return True


class _MLPTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, meta, *params):
        """x [E,cin]; params = (W, b, gamma, beta) per layer; meta = (eps, [bn modules])."""
        lib = _lib.load()
        eps, bns = meta
        L = len(params) // 4
        E, dev = x.shape[0], x.device
        x = x.contiguous()
        saved, scales, shifts, means, rstds = [], [], [], [], []
        prev, pscale, pshift = x, None, None
        with torch.cuda.device(dev):
            for l in range(L):
                W, b, gamma, beta = params[4 * l:4 * l + 4]
                cout, cin = W.shape
                Wp, Bp, K, ldw, _ = pack_conv_layer(W.detach().t(), b.detach())
                Z = torch.empty((E, cout), dtype=torch.float32, device=dev)
                sums = torch.zeros((2, cout), dtype=torch.float64, device=dev)
                rc = lib.gridgcn_linear_fwd(
                    _ptr(prev), E, cin, _ptr(Wp), _ptr(Bp), K, ldw, cout,
                    _ptr(pscale) if pscale is not None else None,
                    _ptr(pshift) if pshift is not None else None, _ptr(Z), _ptr(sums), _stream(x))
                _lib.check(rc, "gridgcn_linear_fwd")
                mean64 = sums[0] / E
                var64 = (sums[1] / E - mean64 * mean64).clamp_min(0.0)
                mean, var = mean64.float(), var64.float()
                rstd = torch.rsqrt(var + eps)
                scale = (gamma.detach() * rstd).contiguous()
                shift = (beta.detach() - mean * scale).contiguous()
                bn = bns[l]
                if bn is not None and bn.track_running_stats:
                    m = bn.momentum
                    bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                    bn.running_var.mul_(1 - m).add_(var * (E / max(E - 1, 1)), alpha=m)
                    bn.num_batches_tracked += 1
                saved.append(Z); scales.append(scale); shifts.append(shift)
                means.append(mean.contiguous()); rstds.append(rstd.contiguous())
                prev, pscale, pshift = Z, scale, shift
            Y = torch.empty_like(prev)
            rc = lib.gridgcn_bn_relu_apply(_ptr(prev), _ptr(pscale), _ptr(pshift), _ptr(Y), E,
                                           prev.shape[1], _stream(x))
            _lib.check(rc, "gridgcn_bn_relu_apply")
        ctx.L = L
        ctx.save_for_backward(x, *saved, *scales, *shifts, *means, *rstds,
                              *[params[4 * l] for l in range(L)])
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.load()
        L = ctx.L
        t = ctx.saved_tensors
        x = t[0]
        Zs, scales, shifts = t[1:1 + L], t[1 + L:1 + 2 * L], t[1 + 2 * L:1 + 3 * L]
        means, rstds, Ws = t[1 + 3 * L:1 + 4 * L], t[1 + 4 * L:1 + 5 * L], t[1 + 5 * L:1 + 6 * L]
        E, dev = x.shape[0], x.device
        grads = [None] * (4 * L)
        dY = dY.contiguous()
        with torch.cuda.device(dev):
            sums = None
            for l in range(L - 1, -1, -1):
                Z, C = Zs[l], Zs[l].shape[1]
                cin = Ws[l].shape[1]
                if sums is None:     # last layer: its BN-backward sums need their own pass
                    sums = torch.zeros((2, C), dtype=torch.float64, device=dev)
                    rc = lib.gridgcn_bn_relu_bwd_reduce(_ptr(dY), _ptr(Z), _ptr(scales[l]),
                                                        _ptr(shifts[l]), _ptr(means[l]),
                                                        _ptr(rstds[l]), E, C, _ptr(sums), _stream(x))
                    _lib.check(rc, "gridgcn_bn_relu_bwd_reduce")
                s1, s2 = sums[0], sums[1]
                grads[4 * l + 3] = s1.float()                       # d beta
                grads[4 * l + 2] = s2.float()                       # d gamma
                # the conv bias feeds a BatchNorm: its gradient is sum(dZ) == 0 analytically
                grads[4 * l + 1] = torch.zeros(C, dtype=torch.float32, device=dev)
                m1 = (s1 / E).float().contiguous()
                m2 = (s2 / E).float().contiguous()
                need_dx = l > 0 or ctx.needs_input_grad[0]
                dX = torch.empty((E, cin), dtype=torch.float32, device=dev) if need_dx else None
                psums = torch.zeros((2, cin), dtype=torch.float64, device=dev) if l > 0 else None
                dW = torch.empty((C, cin), dtype=torch.float32, device=dev)
                Wb = pack_tiles(Ws[l].detach())
                nbytes = ctypes.c_size_t(0)
                lib.gridgcn_linear_bwd_workspace_bytes(E, cin, C, ctypes.byref(nbytes))
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
                prev = Zs[l - 1] if l > 0 else x
                pn = (lambda t: _ptr(t)) if l > 0 else (lambda t: None)
                rc = lib.gridgcn_linear_bwd(
                    _ptr(dY), _ptr(Z), _ptr(scales[l]), _ptr(shifts[l]), _ptr(means[l]),
                    _ptr(rstds[l]), _ptr(m1), _ptr(m2), _ptr(prev),
                    pn(scales[l - 1]), pn(shifts[l - 1]), pn(means[l - 1]), pn(rstds[l - 1]),
                    _ptr(Wb), E, C, cin, _ptr(dX) if need_dx else None, _ptr(dW),
                    _ptr(psums) if psums is not None else None, _ptr(ws), nbytes.value, _stream(x))
                _lib.check(rc, "gridgcn_linear_bwd")
                grads[4 * l] = dW
                dY, sums = dX, psums
        return (dY, None) + tuple(grads)


def pack_tiles(W):
    """W [K, N] -> tile-major [ceil(N/32)][round4(K)][32] (B operand of gridgcn_linear_bwd)."""
    K, N = W.shape
    K4, nt = (K + 3) & ~3, (N + 31) // 32
    Wp = torch.zeros((K4, nt * 32), dtype=torch.float32, device=W.device)
    Wp[:K, :N] = W
    return Wp.reshape(K4, nt, 32).permute(1, 0, 2).contiguous()


def mlp_bn_relu_train(x, layers):
    """x [..., cin] -> [..., cout_last] through `layers` (gridconv.ConvBNReLU modules, training
    mode).  Falls back to nothing: callers check supported() first."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    params = []
    for l in layers:
        params += [l.lin.weight, l.lin.bias, l.bn.weight, l.bn.bias]
    y = _MLPTrain.apply(x2, (layers[0].bn.eps, [l.bn for l in layers]), *params)
    return y.reshape(shp[:-1] + (y.shape[-1],))
