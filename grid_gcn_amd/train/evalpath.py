"""Evaluation-mode paths through the training kernels (running statistics applied on the fly) and their
caches of folded constants / packed weights."""
import weakref

import torch

from .. import _lib
from ..ops import _ptr, _stream
from .options import OPT
from .common import (_PARAM_GEN, _cached_zeros, _mm_nt, packed_sizes)
from .edge import (edge_block_src_supported)

# Evaluation constants of a layer -- BatchNorm folded to (scale, shift) with the running statistics, the packed
# forward operand of the weight -- cached per module and rebuilt when what they are made of may have changed.
# (They were recomputed at every call: five element-wise framework launches per BatchNorm and one pack launch per
# layer, ~200 launches = 1 ms of a 2.7-ms evaluation forward of the segmentation net.)  A key holds the
# Tensor._version of every source tensor -- whoever writes these through raw pointers moves the counter by hand:
# the training kernels for the running statistics (common._stats_written), optim.Adam for the parameters,
# graph.GraphedTrainStep after every replay -- AND the process-wide parameter generation (common._PARAM_GEN), which
# a global optimizer-step hook advances: torch's fused / foreach optimizers rewrite parameters without moving
# Tensor._version.  (`p.data.op_()` moves neither: call clear_eval_cache() or common.params_changed()
# after editing parameters that way.)
_EVAL_BN, _EVAL_W = {}, {}


def clear_eval_cache():
    _EVAL_BN.clear()
    _EVAL_W.clear()


def _bn_eval_vectors(bn):
    """(scale, shift) of a BatchNorm in evaluation mode: y = x * scale + shift"""
    key = (_PARAM_GEN[0], bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.eps)
    e = _EVAL_BN.get(id(bn)) if OPT.EVAL_CACHE else None
    if e is not None and e[0]() is bn and e[1] == key:
        return e[2], e[3]
    with torch.no_grad():
        sc = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
        sh = (bn.bias - bn.running_mean * sc).contiguous()
    if OPT.EVAL_CACHE:
        _EVAL_BN[id(bn)] = (weakref.ref(bn, lambda _r, k=id(bn): _EVAL_BN.pop(k, None)), key, sc, sh)
    return sc, sh


def _eval_packed(lib, lin, cout_p, cin, st):
    """(Bp, Wq, ldw): zero-padded bias and forward operand of lin.weight for a kernel that sees `cin` input
    columns and cout_p >= out_features output columns (gridgcn_pack_linear)"""
    W, b = lin.weight, lin.bias
    cout, cin_w = W.shape
    key = (_PARAM_GEN[0], cout_p, cin, W._version, b._version, W.data_ptr(), b.data_ptr())
    e = _EVAL_W.get(id(lin)) if OPT.EVAL_CACHE else None
    if e is not None and e[0]() is lin and e[1] == key:
        return e[2]
    K, ldw, nwp, nwb = packed_sizes(cout_p, cin)
    pk = torch.empty(ldw + cin * ldw, dtype=torch.float32, device=W.device)
    Bp, Wq = pk[:ldw], pk[ldw:]
    _lib.check(lib.gridgcn_pack_linear(_ptr(W.detach()), _ptr(b.detach()), cout, cin_w, 0, cin, 0, None,
                                       _ptr(Bp), None, None, _ptr(Wq), None, st), "pack")
    if OPT.EVAL_CACHE:
        _EVAL_W[id(lin)] = (weakref.ref(lin, lambda _r, k=id(lin): _EVAL_W.pop(k, None)), key, (Bp, Wq, ldw))
    return Bp, Wq, ldw


def _chain_eval_raw(lib, prev, layers, prev_bn=None):
    """prev [E, cin % 8 == 0] through `layers` with running statistics: returns the LAST layer's
    raw output Z and its BatchNorm (scale, shift) -- the caller applies them (or hands them to a
    kernel that does).  prev_bn = (scale, shift): prev is itself a raw layer output."""
    E, dev = prev.shape[0], prev.device
    sc, sh = prev_bn if prev_bn is not None else (None, None)
    with torch.cuda.device(dev):
        st = _stream(prev)
        for l in layers:
            W, b, bn = l.lin.weight, l.lin.bias, l.bn
            cout, cin_w = W.shape
            cin = prev.shape[1]
            Bp, Wq, ldw = _eval_packed(lib, l.lin, cout, cin, st)
            Z = torch.empty((E, cout), dtype=torch.float32, device=dev)
            _lib.check(lib.gridgcn_linear_fwd_direct(
                _ptr(prev), E, cin, cin, _ptr(Wq), _ptr(Bp), ldw, cout,
                _ptr(sc) if sc is not None else None, _ptr(sh) if sh is not None else None,
                _ptr(Z), None, st), "gridgcn_linear_fwd_direct")
            sc, sh = _bn_eval_vectors(bn)
            prev = Z
    return prev, sc, sh


def edge_block_src_eval_supported(pt_layers, att_layers, src, has_feats, P=None):
    """single-layer point MLP on neighbour features (every up layer): evaluation through the
    training path's forward kernels with running statistics"""
    if len(pt_layers) != 1 or not OPT.SRC_EVAL:
        return False
    if not edge_block_src_supported(pt_layers, att_layers, src, has_feats, P):
        return False
    return all(l.lin.out_features % 8 == 0 and l.lin.out_features <= 256 for l in att_layers)


@torch.no_grad()
def edge_block_src_eval(src, nebidx, cent, pt_layer, att_layers, localfdim, out=None):
    """GridConv edge block in evaluation mode, [B,O,C]: first (only) point conv on the SOURCE points
    (Ysrc = features W_f^T, gathered by the max-pool kernel), attention MLP on the forward MFMA
    kernel with running statistics, product + max over P in gg_k_pairmax_fwd4_src.  For the up
    layers this is faster than the one-launch kernel of csrc/gridgcn_conv.hip, which repeats the
    131 -> 128 conv for every edge (2.45 ms against ~1.2 ms at cfg4 up2)."""
    lib = _lib.load()
    B, Nsrc, Cs = src.shape
    _, O, P = nebidx.shape
    E, R, Cf = B * O * P, B * Nsrc, Cs - 4
    dev = src.device
    W0, b0, bn0 = pt_layer.lin.weight, pt_layer.lin.bias, pt_layer.bn
    C0 = W0.shape[0]
    geo = localfdim != 0
    rot = 3 if geo else 0
    src = src.contiguous()
    with torch.cuda.device(dev):
        st = _stream(src)
        feat = src[..., 4:].reshape(R, Cf)
        Ysrc = _mm_nt(feat, W0[:, rot:])
        wkey = (_PARAM_GEN[0], geo, W0._version, b0._version, W0.data_ptr(), b0.data_ptr())
        e = _EVAL_W.get(("wgb", id(pt_layer))) if OPT.EVAL_CACHE else None
        if e is not None and e[0]() is pt_layer and e[1] == wkey:
            wgb = e[2]
        else:
            wgb = torch.cat([W0[:, :3].t() if geo else _cached_zeros(3 * C0, dev).view(3, C0), b0[None]])
            if OPT.EVAL_CACHE:
                k_ = ("wgb", id(pt_layer))
                _EVAL_W[k_] = (weakref.ref(pt_layer, lambda _r, k=k_: _EVAL_W.pop(k, None)), wkey, wgb)
        att16 = torch.empty((E, 16), dtype=torch.float32, device=dev)
        rc = lib.gridgcn_edge_lin0_forward(
            _ptr(Ysrc), _ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B, Nsrc, Cs, O, P, C0,
            _ptr(wgb) if geo else None, _ptr(wgb[3]), None, _ptr(att16), None, st)
        _lib.check(rc, "gridgcn_edge_lin0_forward")
        sc_p, sh_p = _bn_eval_vectors(bn0)
        a2 = att_layers[-1]
        C = a2.lin.out_features
        ncent = B * O
        # out: [ncent, C] destination with its own row stride (one half of update_func's concat)
        agg = out if out is not None else torch.empty((ncent, C), dtype=torch.float32, device=dev)
        ldo = agg.stride(0)
        if OPT.ATT_MAX_EVAL and a2.lin.in_features == 32 and C in (64, 128) and P <= 128:
            # second attention conv + activations + product + max in one kernel: the [E, C]
            # attention tensor is never written (csrc/gridgcn_atteval.hip)
            Z1, s1, h1 = _chain_eval_raw(lib, att16, att_layers[:-1])
            sc_a, sh_a = _bn_eval_vectors(a2.bn)
            rc = lib.gridgcn_att_max_eval(
                _ptr(Z1), _ptr(s1), _ptr(h1), _ptr(a2.lin.weight), _ptr(a2.lin.bias), _ptr(sc_a),
                _ptr(sh_a), _ptr(Ysrc), _ptr(nebidx), _ptr(att16), _ptr(wgb) if geo else None,
                _ptr(wgb[3]), _ptr(sc_p), _ptr(sh_p), B, Nsrc, O, P, C, _ptr(agg), ldo, st)
            _lib.check(rc, "gridgcn_att_max_eval")
            return agg if out is not None else agg.view(B, O, C)
        Za, sc_a, sh_a = _chain_eval_raw(lib, att16, att_layers)
        amax = torch.empty((ncent, C), dtype=torch.uint8, device=dev)
        rc = lib.gridgcn_pairmax_fwd_src(
            _ptr(Ysrc), _ptr(nebidx), _ptr(att16), _ptr(wgb) if geo else None, _ptr(wgb[3]), B,
            Nsrc, O, _ptr(Za), _ptr(sc_p), _ptr(sh_p), _ptr(sc_a), _ptr(sh_a), ncent, P, C,
            _ptr(agg), ldo, _ptr(amax), None, st)
        _lib.check(rc, "gridgcn_pairmax_fwd_src")
    return agg if out is not None else agg.view(B, O, C)


@torch.no_grad()
def mlp_bn_relu_eval(x, layers, out=None):
    """(out: optional [E, cout] destination with its own row stride; returned as is.)
    Inference through the same forward kernel: layer l computes Z_l = act(Z_{l-1}) W_l^T + b_l
    with act = the previous layer's BatchNorm (running statistics) + ReLU applied while the rows are
    loaded; one BatchNorm+ReLU pass at the end.  x [..., cin] float32 on the GPU."""
    lib = _lib.load()
    shp = x.shape
    prev = x.reshape(-1, shp[-1]).contiguous()
    E, dev = prev.shape[0], prev.device
    if prev.shape[1] % 8:                      # the kernel reads rows in 32-byte pieces
        prev = torch.nn.functional.pad(prev, (0, 8 - prev.shape[1] % 8))
    Z, sc, sh = _chain_eval_raw(lib, prev, layers)
    with torch.cuda.device(dev):
        Y = out if out is not None else torch.empty_like(Z)
        _lib.check(lib.gridgcn_bn_relu_apply(_ptr(Z), _ptr(sc), _ptr(sh), _ptr(Y), E,
                                             Y.shape[1], Y.stride(0), _stream(Z)),
                   "gridgcn_bn_relu_apply")
    if out is not None:
        return Y
    return Y.reshape(shp[:-1] + (Y.shape[1],))


@torch.no_grad()
def head_eval(x, layers, lin):
    """Evaluation of conv+BN+ReLU `layers` followed by the Linear `lin` (dropout is the identity):
    the last BatchNorm+ReLU is applied while `lin`'s kernel loads its rows, so no activation pass and
    no stock GEMM remain.  x [..., cin] -> [..., lin.out_features] (a view of class-padded rows)."""
    lib = _lib.load()
    shp = x.shape
    prev = x.reshape(-1, shp[-1]).contiguous()
    if prev.shape[1] % 8:
        prev = torch.nn.functional.pad(prev, (0, 8 - prev.shape[1] % 8))
    Z, sc, sh = _chain_eval_raw(lib, prev, layers)
    E, cin = Z.shape
    dev = Z.device
    C = lin.out_features
    Cp = (C + 7) & ~7
    with torch.cuda.device(dev):
        st = _stream(Z)
        Bp, Wq, ldw = _eval_packed(lib, lin, C, cin, st)
        Y = torch.empty((E, Cp), dtype=torch.float32, device=dev)
        _lib.check(lib.gridgcn_linear_fwd_direct(_ptr(Z), E, cin, cin, _ptr(Wq), _ptr(Bp), ldw, Cp,
                                                 _ptr(sc), _ptr(sh), _ptr(Y), None, st),
                   "gridgcn_linear_fwd_direct")
    return Y[:, :C].reshape(shp[:-1] + (C,))


@torch.no_grad()
def edge_block_cls_eval(src, nebidx, cent, pt_layers, att1_layers, att2_layers):
    """The classification edge block in evaluation mode (running statistics) on the same forward
    kernels as _EdgeBlockClsTrain: [B,O,C]."""
    lib = _lib.load()
    src, cent = src.contiguous(), cent.contiguous()
    B, Nsrc, Cs = src.shape
    _, O, P = nebidx.shape
    E, R, Cf, ncent = B * O * P, B * Nsrc, Cs - 4, B * O
    dev = src.device
    with torch.cuda.device(dev):
        st = _stream(src)
        ctxv = torch.empty((ncent, 3 + Cf), dtype=torch.float32, device=dev)
        _lib.check(lib.gridgcn_ctx_max(_ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B, Nsrc,
                                       Cs, O, P, _ptr(ctxv), None, st), "gridgcn_ctx_max")
        att16 = torch.empty((E, 16), dtype=torch.float32, device=dev)
        if Cf > 0:
            l0 = pt_layers[0]
            W0, C0 = l0.lin.weight, l0.lin.out_features
            Ysrc = _mm_nt(src[..., 4:].reshape(R, Cf), W0[:, 3:])
            wgb = torch.cat([W0[:, :3].t(), l0.lin.bias[None]])
            Z0 = torch.empty((E, C0), dtype=torch.float32, device=dev)
            rc = lib.gridgcn_edge_lin0_forward(
                _ptr(Ysrc), _ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B, Nsrc, Cs, O, P,
                C0, _ptr(wgb), _ptr(wgb[3]), _ptr(Z0), _ptr(att16), None, st)
            _lib.check(rc, "gridgcn_edge_lin0_forward")
            Zl, scl, shl = _chain_eval_raw(lib, Z0, pt_layers[1:], _bn_eval_vectors(l0.bn))
        else:
            x0 = torch.empty((E, 8), dtype=torch.float32, device=dev)
            rc = lib.gridgcn_edge_inputs_rows(_ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B,
                                              Nsrc, Cs, O, P, 0, 0, 8, _ptr(x0), _ptr(att16), st)
            _lib.check(rc, "gridgcn_edge_inputs_rows")
            Zl, scl, shl = _chain_eval_raw(lib, x0, pt_layers)
        C = Zl.shape[1]
        Za1, sc1, sh1 = _chain_eval_raw(lib, att16, att1_layers)
        A0 = Za1.shape[1]
        a20 = att2_layers[0]
        W2, N0, K12 = a20.lin.weight, a20.lin.out_features, A0 + C
        rowb = _mm_nt(ctxv, W2[:, K12:], bias=a20.lin.bias)
        _, ldw, _, _ = packed_sizes(N0, K12)
        Wq = torch.empty(K12 * ldw, dtype=torch.float32, device=dev)
        W12 = W2[:, :K12].contiguous()            # (a name: the copy must outlive the call that reads it)
        _lib.check(lib.gridgcn_pack_linear(_ptr(W12), None, N0, K12, 0, K12, 0,
                                           None, None, None, None, _ptr(Wq), None, st),
                   "gridgcn_pack_linear")
        Z20 = torch.empty((E, N0), dtype=torch.float32, device=dev)
        psc, psh = torch.cat([sc1, scl]), torch.cat([sh1, shl])
        rc = lib.gridgcn_linear_fwd_direct2(_ptr(Za1), A0, A0, _ptr(Zl), C, C, E, _ptr(Wq), None,
                                            _ptr(rowb), P, ldw, N0, _ptr(psc), _ptr(psh),
                                            _ptr(Z20), None, st)
        _lib.check(rc, "gridgcn_linear_fwd_direct2")
        Za, sca, sha = _chain_eval_raw(lib, Z20, att2_layers[1:], _bn_eval_vectors(a20.bn))
        agg = torch.empty((ncent, C), dtype=torch.float32, device=dev)
        amax = torch.empty((ncent, C), dtype=torch.uint8, device=dev)
        rc = lib.gridgcn_pairmax_fwd(_ptr(Zl), _ptr(Za), _ptr(scl), _ptr(shl), _ptr(sca), _ptr(sha),
                                     ncent, P, C, _ptr(agg), C, _ptr(amax), None, st)
        _lib.check(rc, "gridgcn_pairmax_fwd")
    return agg.view(B, O, C)
