"""Segmentation / classification heads in training mode: class scores, fc1 + Dropout + fc2 as one op,
SoftmaxOutput(use_ignore, normalization='valid') (segmentation/models/ggcn_models_g.py:30-43)."""
import ctypes

import torch

from .. import _lib
from ..ops import _ptr, _stream
from .options import OPT
from .common import (  # noqa: F401
    PACKS, _chain_backward, _chain_forward, _identity_consts, _mm_nn, _mm_nt, _tn_matmul, _zeros,
    packed_sizes, supported,)

def linear_plain_supported(x, lin):
    return (x.is_cuda and x.dtype == torch.float32 and OPT.DIRECT_FWD and OPT.DIRECT_DX
            and x.shape[-1] % 8 == 0 and x.shape[-1] <= 256 and lin.out_features <= 32
            and lin.bias is not None)


class _LinearPlain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        lib = _lib.load()
        x = x.contiguous()
        E, cin = x.shape
        C = W.shape[0]
        Cp = (C + 7) & ~7
        dev = x.device
        ndx = cin if ctx.needs_input_grad[0] else 0
        nt = (ndx + 31) // 32
        ntv = 1 if nt <= 1 else 2 if nt <= 2 else 4 if nt <= 4 else 8
        K, ldw, nwp, nwb = packed_sizes(C, cin)
        with torch.cuda.device(dev):
            st = _stream(x)
            _, Bp, Wb, _, Wq, Wdx = PACKS.get(lib, W, b, C, cin, 0, cin, ndx, True,
                                              (0, ldw, nwb, cin * ldw, Cp * 32 * ntv if ndx else 0), st)
            if Wdx is None:
                Wdx = Wb
            Z = torch.empty((E, Cp), dtype=torch.float32, device=dev)
            _lib.check(lib.gridgcn_linear_fwd_direct(_ptr(x), E, cin, cin, _ptr(Wq), _ptr(Bp), ldw,
                                                     Cp, None, None, _ptr(Z), None, st),
                       "gridgcn_linear_fwd_direct")
        ctx.save_for_backward(x, Z, Wb, Wdx)
        ctx.dims = (C, Cp, ndx)
        return Z[:, :C]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, Z, Wb, Wdx = ctx.saved_tensors
        C, Cp, ndx = ctx.dims
        E, cin = x.shape
        dev = x.device
        # the loss op (softmax_ce below) hands back the [:, :C] view of a zero-padded [E, Cp] buffer
        if g.stride() == (Cp, 1) and g.storage_offset() == 0 and \
                g.untyped_storage().nbytes() == E * Cp * 4:
            dL = g.as_strided((E, Cp), (Cp, 1))
        else:
            dL = torch.zeros((E, Cp), dtype=torch.float32, device=dev)
            dL[:, :C] = g
        ident = _identity_consts(Cp, dev)
        with torch.cuda.device(dev):
            st = _stream(x)
            dX = torch.empty((E, cin), dtype=torch.float32, device=dev) if ndx else None
            dW = torch.empty((Cp, cin), dtype=torch.float32, device=dev)
            nbytes = ctypes.c_size_t(0)
            lib.gridgcn_linear_bwd_workspace_bytes(E, cin, Cp, ctypes.byref(nbytes))
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            rc = lib.gridgcn_linear_bwd(
                _ptr(dL), _ptr(Z), _ptr(ident[0]), _ptr(ident[1]), _ptr(ident[2]), _ptr(ident[3]),
                _ptr(ident[4]), _ptr(ident[5]), _ptr(x), None, None, None, None, _ptr(Wb), None,
                _ptr(Wdx) if ndx else None, ndx, E, Cp, cin, cin, 0, 0,
                _ptr(dX) if ndx else None, _ptr(dW), None, None, None, 0, _ptr(ws), nbytes.value, st)
            _lib.check(rc, "gridgcn_linear_bwd")
            db64 = _zeros(784, torch.float64, dev)              # (16 slots of partial sums | tickets)
            db = torch.empty(C, dtype=torch.float32, device=dev)
            _lib.check(lib.gridgcn_colsum_f32(_ptr(dL), E, Cp, C, _ptr(db64), _ptr(db), st), "gridgcn_colsum")
        return dX, dW[:C], db


class _LinearMM(torch.autograd.Function):
    """x W^T + b of a torch.nn.Linear of any shape on csrc/gridgcn_gemm.hip (the classifier's 256 -> 40 scores on
    a batch of rows: nothing for a conv + BatchNorm kernel, and not worth a GEMM library)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        with torch.cuda.device(x.device):
            return _mm_nt(x.detach(), W.detach(), bias=b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous()
        with torch.cuda.device(x.device):
            dX = _mm_nn(g, W.detach()) if ctx.needs_input_grad[0] else None
            dW = _tn_matmul(g, x.detach())
        return dX, dW, g.sum(0)


def linear_mm(x, lin):
    """torch.nn.Linear on fp32 GPU rows through _LinearMM (stock module elsewhere)"""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and lin.bias is not None:
        return _LinearMM.apply(x.contiguous(), lin.weight, lin.bias)
    return lin(x)


def linear_plain_train(x, lin):
    """x [..., cin] -> [..., C] = x W^T + b of a torch.nn.Linear with C <= 32 (the class scores)."""
    shp = x.shape
    y = _LinearPlain.apply(x.reshape(-1, shp[-1]), lin.weight, lin.bias)
    return y.reshape(shp[:-1] + (y.shape[-1],))


class _HeadTrain(torch.autograd.Function):
    """conv+BN+ReLU chain -> Dropout(p) -> Linear (ggcn_models_g.py:33-38: fc1, fc1/dropout, fc2)
    as one op.  The Dropout mask is a hash of (seed, element index) that every reader of the dropped
    activation evaluates itself: fc2's forward while it loads the rows of Z_fc1 (BatchNorm + ReLU + mask
    in the prologue), fc2's weight-gradient kernel on its B operand, and the kernel that writes fc2's input
    gradient -- whose epilogue also accumulates fc1's BatchNorm-backward sums (FUSE_DROPOUT; fp32 mode,
    128 columns).  Otherwise the dropped activation is written once (gridgcn_bn_relu_dropout_apply) and
    only the backward regenerates the mask.  Against the separate ops this drops the dropout forward /
    backward passes, the dropped tensor and one reduce pass over the [E, C] gradient."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        lib = _lib.load()
        eps, bns, p, seed, seed_dev, prev = (tuple(meta) + (None,))[:6]
        L = (len(params) - 2) // 4
        W2, b2 = params[4 * L], params[4 * L + 1]
        x = x.contiguous()
        E, dev = x.shape[0], x.device
        C2 = W2.shape[0]
        Cp = (C2 + 7) & ~7
        with torch.cuda.device(dev):
            stream = _stream(x)
            st = _chain_forward(lib, x, params[:4 * L], bns, eps, 0,
                                x.shape[1] if ctx.needs_input_grad[0] else 0,
                                prev_bn=prev.prev_bn()[:2] if prev is not None else None)
            C = st.Z[-1].shape[1]
            # the dropped activation is never stored when fc2's kernels can evaluate the mask themselves
            # (fp32 mode, 128 columns: the segmentation head's shape)
            fuse = (OPT.FUSE_DROPOUT and 0.0 < float(p) < 1.0 and C == 128 and E * C < 2 ** 32
                    and lib.gridgcn_get_mlp_precision() == 0)
            sd = _ptr(seed_dev) if seed_dev is not None else None
            K, ldw, nwp, nwb = packed_sizes(C2, C)
            ntv = next(v for v in (1, 2, 4, 8) if v * 32 >= C)
            _, Bp, Wb, _, Wq, Wdx = PACKS.get(lib, W2, b2, C2, C, 0, C, C, True,
                                              (0, ldw, nwb, C * ldw, Cp * 32 * ntv), stream)
            Z2 = torch.empty((E, Cp), dtype=torch.float32, device=dev)
            if fuse:
                Hd = st.Z[-1].new_empty(0)
                _lib.check(lib.gridgcn_linear_fwd_direct_drop(
                    _ptr(st.Z[-1]), E, C, C, _ptr(Wq), _ptr(Bp), ldw, Cp, _ptr(st.scale[-1]),
                    _ptr(st.shift[-1]), _ptr(Z2), float(p), int(seed), sd, stream),
                    "gridgcn_linear_fwd_direct_drop")
            else:
                Hd = torch.empty((E, C), dtype=torch.float32, device=dev)
                _lib.check(lib.gridgcn_bn_relu_dropout_apply(
                    _ptr(st.Z[-1]), _ptr(st.scale[-1]), _ptr(st.shift[-1]), _ptr(Hd), E, C, C,
                    float(p), int(seed), sd, stream), "gridgcn_bn_relu_dropout_apply")
                _lib.check(lib.gridgcn_linear_fwd_direct(_ptr(Hd), E, C, C, _ptr(Wq), _ptr(Bp), ldw,
                                                         Cp, None, None, _ptr(Z2), None, stream),
                           "gridgcn_linear_fwd_direct")
        ctx.L = L
        ctx.prev = prev
        ctx.ndx = st.ndx
        ctx.drop = (float(p), int(seed), seed_dev)
        ctx.dims = (C2, Cp)
        ctx.save_for_backward(x, *st.Z, *st.scale, *st.shift, *st.mean, *st.rstd, *st.Wb, *st.Wg,
                              *st.Wdx, Hd, Z2, Wb, Wdx)
        return Z2[:, :C2]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        L = ctx.L
        t = ctx.saved_tensors
        x = t[0]
        Zs, scales, shifts = t[1:1 + L], t[1 + L:1 + 2 * L], t[1 + 2 * L:1 + 3 * L]
        means, rstds, Wbs = t[1 + 3 * L:1 + 4 * L], t[1 + 4 * L:1 + 5 * L], t[1 + 5 * L:1 + 6 * L]
        Wgs, Wdxs = t[1 + 6 * L:1 + 7 * L], t[1 + 7 * L:1 + 8 * L]
        Hd, Z2, Wb2, Wdx2 = t[1 + 8 * L:]
        C2, Cp = ctx.dims
        E, dev = x.shape[0], x.device
        C = Zs[-1].shape[1]
        if g.stride() == (Cp, 1) and g.storage_offset() == 0 and \
                g.untyped_storage().nbytes() == E * Cp * 4:
            dL = g.as_strided((E, Cp), (Cp, 1))      # softmax_ce's zero-padded gradient buffer
        else:
            dL = torch.zeros((E, Cp), dtype=torch.float32, device=dev)
            dL[:, :C2] = g
        ident = _identity_consts(Cp, dev)
        with torch.cuda.device(dev):
            st = _stream(x)
            dH = torch.empty((E, C), dtype=torch.float32, device=dev)
            acc = _zeros(2 * C + 784, torch.float64, dev)       # (db64: 16 slots of partial sums | tickets)
            sums, db64 = acc[:2 * C], acc[2 * C:]
            db2 = torch.empty(C2, dtype=torch.float32, device=dev)
            # gradient w.r.t. relu(bn(Z_fc1)) (dropout mask applied) + fc1's BatchNorm-backward sums
            _lib.check(lib.gridgcn_linear_dx(
                _ptr(dL), _ptr(Z2), _ptr(ident[0]), _ptr(ident[1]), _ptr(ident[2]), _ptr(ident[3]),
                _ptr(ident[4]), _ptr(ident[5]), _ptr(Zs[-1]), _ptr(scales[-1]), _ptr(shifts[-1]),
                _ptr(means[-1]), _ptr(rstds[-1]), _ptr(Wdx2), C, E, Cp, C, Cp, ctx.drop[0],
                ctx.drop[1], _ptr(ctx.drop[2]) if ctx.drop[2] is not None else None,
                _ptr(dH), _ptr(sums), st), "gridgcn_linear_dx")
            dW2 = torch.empty((Cp, C), dtype=torch.float32, device=dev)
            nbytes = ctypes.c_size_t(0)
            lib.gridgcn_linear_bwd_workspace_bytes(E, C, Cp, ctypes.byref(nbytes))
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            if Hd.numel() == 0:         # (fused Dropout: fc2's input is rebuilt from Z_fc1 and the mask)
                _lib.check(lib.gridgcn_linear_dw_drop(
                    _ptr(dL), _ptr(Z2), _ptr(ident[0]), _ptr(ident[1]), _ptr(ident[2]), _ptr(ident[3]),
                    _ptr(ident[4]), _ptr(ident[5]), _ptr(Zs[-1]), _ptr(scales[-1]), _ptr(shifts[-1]),
                    E, Cp, C, ctx.drop[0], ctx.drop[1],
                    _ptr(ctx.drop[2]) if ctx.drop[2] is not None else None,
                    _ptr(dW2), _ptr(ws), nbytes.value, st), "gridgcn_linear_dw_drop")
            else:
                _lib.check(lib.gridgcn_linear_bwd(
                    _ptr(dL), _ptr(Z2), _ptr(ident[0]), _ptr(ident[1]), _ptr(ident[2]), _ptr(ident[3]),
                    _ptr(ident[4]), _ptr(ident[5]), _ptr(Hd), None, None, None, None, _ptr(Wb2), None,
                    None, 0, E, Cp, C, C, 0, 0, None, _ptr(dW2), None, None, None, 0, _ptr(ws),
                    nbytes.value, st), "gridgcn_linear_bwd")
            _lib.check(lib.gridgcn_colsum_f32(_ptr(dL), E, Cp, C2, _ptr(db64), _ptr(db2), st), "gridgcn_colsum")
            prev = ctx.prev
            r = _chain_backward(lib, x, Zs, scales, shifts, means, rstds, Wbs, Wgs, Wdxs,
                                ctx.ndx, sums, dH, None, ctx.needs_input_grad[0],
                                prev_bn=prev.prev_bn() if prev is not None else None,
                                nbn=prev.nbn() if prev is not None else 0)
            dX, grads = r[0], r[1]
            if prev is not None:
                prev.sums = prev.take_sums(r[2], x.shape[1])
        return (dX, None) + tuple(grads) + (dW2[:C2], db2)


def head_supported(x, layers, lin):
    C = layers[-1].lin.out_features
    return (supported(layers, x) and OPT.DIRECT_FWD and OPT.DIRECT_DX and lin.bias is not None
            and lin.out_features <= 32 and lin.in_features == C and C % 32 == 0 and C <= 256)


def head_train(x, layers, p, lin, seed=None, seed_dev=None, prev=None):
    """x [..., cin] -> class scores [..., lin.out_features] through `layers` (ConvBNReLU modules in
    training mode), Dropout(p) and the Linear `lin`.  seed: dropout seed (None: drawn from torch's
    CPU generator, i.e. reproducible under torch.manual_seed).  seed_dev: optional int64 GPU scalar
    added to the seed inside the kernels (a captured hipGraph then drops a fresh mask per replay)."""
    if seed is None:
        # with a device-side seed the variation comes from that scalar (and a host draw would be
        # frozen into a captured graph anyway)
        seed = 0 if seed_dev is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
    shp = x.shape
    params = []
    for l in layers:
        params += [l.lin.weight, l.lin.bias, l.bn.weight, l.bn.bias]
    params += [lin.weight, lin.bias]
    y = _HeadTrain.apply(x.reshape(-1, shp[-1]), (layers[0].bn.eps, [l.bn for l in layers],
                                                  float(p), seed, seed_dev, prev), *params)
    return y.reshape(shp[:-1] + (y.shape[-1],))


def dropout_mask(E, C, p, seed, device):
    """The {0, 1/(1-p)} factors head_train applies for (p, seed) on an [E, C] activation (tests)."""
    lib = _lib.load()
    one = torch.ones((E, C), dtype=torch.float32, device=device)
    sc, sh = torch.ones(C, device=device), torch.zeros(C, device=device)
    m = torch.empty_like(one)
    with torch.cuda.device(one.device):
        _lib.check(lib.gridgcn_bn_relu_dropout_apply(_ptr(one), _ptr(sc), _ptr(sh), _ptr(m), E, C,
                                                     C, float(p), int(seed), None, _stream(one)), "drop")
    return m


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, ignore, cw=None):
        lib = _lib.load()
        E, C = logits.shape
        dev = logits.device
        ld = logits.stride(0)
        if not (logits.stride(1) == 1 and ld % 4 == 0 and C <= ld <= 32
                and logits.storage_offset() == 0 and (ld == C or _pad_is_zero(logits, ld))):
            ld = (C + 7) & ~7
            buf = torch.zeros((E, ld), dtype=torch.float32, device=dev)
            buf[:, :C] = logits
            logits = buf[:, :C]
        label = label.contiguous()
        lse = torch.empty(E, dtype=torch.float32, device=dev)
        acc = _zeros(544, torch.float64, dev)           # 16 slots | total, count at [256:258] | tickets
        loss = torch.empty((), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.gridgcn_softmax_ce_loss(_ptr(logits), ld, C, _ptr(label), E, ignore,
                                                   _ptr(lse), _ptr(acc), _ptr(loss), _stream(logits)),
                       "gridgcn_softmax_ce_loss")
        ctx.save_for_backward(logits, label, lse, acc[256:258])
        ctx.meta = (ld, ignore)
        ctx.cw = cw
        # SoftmaxOutput(normalization='valid'): the valid count is clamped to >= 1, so a batch
        # whose labels are all ignore_label gives loss 0 and gradient 0 instead of 0/0
        # (formed by the kernel's last workgroup)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        logits, label, lse, acc = ctx.saved_tensors
        ld, ignore = ctx.meta
        E, C = logits.shape
        dev = logits.device
        d = torch.empty((E, ld), dtype=torch.float32, device=dev)
        g = g.contiguous().float()
        with torch.cuda.device(dev):
            _lib.check(lib.gridgcn_softmax_ce_bwd(_ptr(logits), ld, C, _ptr(label), E, ignore,
                                                  _ptr(lse), _ptr(acc), _ptr(g),
                                                  _ptr(ctx.cw) if ctx.cw is not None else None,
                                                  _ptr(d), _stream(logits)),
                       "gridgcn_softmax_ce_bwd")
        return d[:, :C], None, None, None


def _pad_is_zero(logits, ld):
    # a [:, :C] view of an [E, ld] buffer produced by _LinearPlain: padding columns are exact zeros
    return logits.untyped_storage().nbytes() == logits.shape[0] * ld * 4


def softmax_ce(logits, label, ignore_index, class_weight=None):
    """mean over label != ignore_index of -log softmax(logits)[label]; logits [E, C <= 32] f32 on
    the GPU, label [E] int64 (torch.nn.functional.cross_entropy(..., ignore_index, 'mean')).
    class_weight [C] (optional): the 'weighted_gradient' op of the reference in front of the loss
    (custom_op/weighted_gradient.py): the loss VALUE is unchanged, the gradient of a row is
    multiplied by the weight of its label."""
    cw = None
    if class_weight is not None:
        cw = torch.as_tensor(class_weight, dtype=torch.float32, device=logits.device).contiguous()
        assert cw.numel() == logits.shape[1]
    return _SoftmaxCE.apply(logits, label.long(), int(ignore_index), cw)
