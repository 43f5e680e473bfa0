"""Training path on the hand-written gfx950 kernels: the host logic of one training step between
the index ops and the optimizer (kernels: csrc/gridgcn_direct.hip, gridgcn_edgelin.hip,
gridgcn_pairmax.hip, gridgcn_scatter.hip, gridgcn_head.hip; LDS-staged fallbacks for odd shapes in
gridgcn_train.hip).  Everything goes through the C ABI of include/gridgcn.h.

A "chain" is a stack of (1x1 conv -> BatchNorm(batch statistics) -> ReLU) layers (mlp2d_c /
mlp1d_c, utils/ops.py:236-260):

  forward   per layer: pack the weight layouts (1 launch), Z_l = act_{l-1} * W_l + b_l on fp32 MFMA
            where act_{l-1} = relu(bn(Z_{l-1})) is applied while the rows are loaded (never
            materialised) and the epilogue accumulates the batch statistics of Z_l, then the
            BatchNorm bookkeeping (1 launch).
  backward  per layer: dZ formed in registers, a dX kernel (with the previous layer's
            BatchNorm-backward sums in its epilogue) and a dW kernel + reduce.

autograd Functions:
  _MLPTrain           one chain, dense output Y = relu(bn(Z_L))     (centre / update MLPs, fc1)
  _EdgeBlockSrcTrain  the GridConv edge block from (src, nebidx, cent): first point conv on the
                      source points + gather-add, remaining pt layers and the att chain on [E, .]
                      tensors, then agg[o,c] = max_p relu(bn(Zpt)) * relu(bn(Zatt)) without
                      materialising the two activations, their product, or (in backward) any dense
                      gradient of them (segmentation/models/gcn_module_g_att.py:135-167, 57-59)
  _EdgeBlockTrain     the same block on already gathered edge rows (layers without neighbour
                      features)
  _LinearPlain, _SoftmaxCE   the class scores and SoftmaxOutput(use_ignore, 'valid')
  _Cat2               update_func's concat, written in place by its two producers

Numerics follow torch.nn.BatchNorm1d(eps, momentum) exactly as used by gridconv.ConvBNReLU (biased
variance for normalisation, unbiased for the running estimate).
"""
import ctypes
import weakref

import torch
from torch.optim import optimizer as _torch_optimizer_mod

from . import _lib
from .ops import _ptr, _stream


# Path switches.  Plain module constants (tests flip them with monkeypatch to compare two paths on
# the same inputs); nothing here, and nothing in the C library, reads the process environment.
# register-direct forward / dX kernels (csrc/gridgcn_direct.hip) for row widths that are a multiple of 8
DIRECT_FWD = True
DIRECT_DX = True
# first conv of the point MLP applied to the source points and gathered (csrc/gridgcn_edgelin.hip)
SRC_FIRST_CONV = True
# ... and, for single-layer point MLPs, recomputed by its consumers instead of stored
NO_Z0 = True
# ... and its backward reduced to the sparse arg-max entries (gg_k_edge_lin0_bwd_sparse)
SPARSE_L0 = True
# bf16 mode (set_mlp_precision("bf16")): bf16 STORAGE of the attention pre-activation of the up layers
Z16_STORAGE = True
# fp32 mode, up layers: backward of the second attention conv without its [E, 128] pre-activation
# (csrc/gridgcn_attbwd_nz.hip): the tensor is not kept for the backward at all
NOZ_ATT_BWD = True




def set_mlp_precision(mode):
    """'fp32' (default): exact fp32 MFMA, the parity path.  'bf16': the register-direct GEMM kernels
    (forward, dX, dW of every conv of the training path) round their operands to bf16 in registers
    and use v_mfma_f32_32x32x16_bf16 with fp32 accumulation; tensors in HBM, BatchNorm statistics
    and all epilogues stay fp32 (BASELINE configs[2]: 'bf16 MLP / fp32 indices').  Process-wide,
    read when a kernel is launched (include/gridgcn.h: gridgcn_set_mlp_precision)."""
    assert mode in ("fp32", "bf16")
    _lib.check(_lib.load().gridgcn_set_mlp_precision(1 if mode == "bf16" else 0), "set_mlp_precision")


def get_mlp_precision():
    return "bf16" if _lib.load().gridgcn_get_mlp_precision() else "fp32"

def _momentum(bn):
    """BatchNorm momentum handed to gg_k_bn_finalize.  momentum=None (torch's cumulative moving
    average) has no counterpart in the reference (mx.sym.BatchNorm(momentum=bn_decay)) nor in the
    kernel: refuse it instead of passing None through ctypes."""
    if bn.momentum is None:
        raise RuntimeError("BatchNorm momentum=None (cumulative average) is not supported by the "
                           "training kernels; use momentum = 1 - bn_decay (gridconv.ConvBNReLU)")
    return float(bn.momentum)

def supported(layers, x):
    if not (x.is_cuda and x.dtype == torch.float32):
        return False
    for l in layers:
        c = l.lin.out_features
        if l.bn is None or not l.use_relu or c > 256 or 256 % c != 0:
            return False
        cin = l.lin.in_features
        if cin > 384:                        # <= 12 column tiles in gg_k_linear_bwd
            return False
        # input widths the register-direct dW kernel does not take go to the LDS-staged backward,
        # which holds at most 48 (input tile, output tile) pairs
        if not _dw_direct_ok(c, cin) and ((cin + 31) // 32) * ((c + 31) // 32) > 48:
            return False
    return True


def pack_tiles(W):
    """W [K, N] -> tile-major [ceil(N/32)][round4(K)][32] (B operand of gridgcn_linear_bwd)."""
    K, N = W.shape
    K4, nt = (K + 3) & ~3, (N + 31) // 32
    Wp = torch.zeros((K4, nt * 32), dtype=torch.float32, device=W.device)
    Wp[:K, :N] = W
    return Wp.reshape(K4, nt, 32).permute(1, 0, 2).contiguous()


def pack_groups(W):
    """W [K, N] -> column blocks of 4/2/1 tiles, each [round4(K)][32][nt] (B operand of
    gg_k_linear_dx: one vector load per k-step feeds nt MFMAs)."""
    K, N = W.shape
    K4, ntile = (K + 3) & ~3, (N + 31) // 32
    Wp = torch.zeros((K4, ntile * 32), dtype=torch.float32, device=W.device)
    Wp[:K, :N] = W
    blocks, done = [], 0
    while done < ntile:
        rem = ntile - done
        nt = 4 if rem >= 4 else (2 if rem >= 2 else 1)
        blk = Wp[:, done * 32:(done + nt) * 32].reshape(K4, nt, 32).permute(0, 2, 1)
        blocks.append(blk.contiguous().reshape(-1))
        done += nt
    return torch.cat(blocks).contiguous()


class _Chain:
    """forward state of one chain: Z_l and the BatchNorm vectors of every layer."""

    def __init__(self):
        self.Z, self.scale, self.shift, self.mean, self.rstd = [], [], [], [], []
        self.Wb, self.Wg, self.Wdx, self.ndx = [], [], [], []


def packed_sizes(C, cin):
    """(K, ldw, floats of Wp, floats of Wb == floats of Wg) of gridgcn_pack_linear."""
    K = (cin + 3) & ~3
    ldw = next(x for x in (32, 64, 128, 256) if x >= C)
    return K, ldw, K * ldw, ((cin + 31) // 32) * ((C + 3) & ~3) * 32


class _PackCache:
    """Operand layouts of every conv layer (gridgcn_pack_linear) in persistent buffers, one entry per
    (weight Parameter, layout request).  prepack(module) -- called by the models at the start of a
    training forward -- rebuilds the layouts of ALL of the module's entries in ONE launch
    (gridgcn_pack_linear_batch over a device-side descriptor table), and each entry then serves
    exactly one lookup without a launch of its own: the ~27 pack launches of a step become one.
    Any other lookup (no prepack before it, a second use in the same forward, a weight whose
    version counter has moved since) packs its layer alone, as before.  (Freshness cannot be read
    off Tensor._version alone: the fused optimizers update weights without moving it.)

    Ordering contract: forward -> backward -> optimizer step.  The buffers are shared by every forward
    that uses the Parameter and the views of them are what save_for_backward keeps, so a re-pack between
    a forward and ITS backward would silently change that backward's operands.  The kernels write
    through raw pointers, which autograd cannot see -- so every re-pack moves the buffer's version
    counter by hand (torch.autograd.graph.increment_version) and autograd's own saved-tensor check
    turns such an interleaving into its "modified by an inplace operation" error instead of wrong
    gradients (tests/test_model_cpu.py::test_pack_cache_*)."""

    def __init__(self):
        self.entries = {}          # key -> dict(W, b, pk, bufs, fresh, ver, desc)
        self.tables = {}           # id(module) -> (weakref(module), keys, device table, max_n)

    def _drop(self, key):
        self.entries.pop(key, None)
        for m in [m for m, t in self.tables.items() if key in t[1]]:
            del self.tables[m]

    def get(self, lib, W, b, cout, cin_w, rot, cin, ndx, direct, sizes, stream):
        if not (isinstance(W, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter)):
            # a temporary (a slice, a product): nothing to key a cache entry on -- packed per call
            bufs = self._alloc(W.device, direct, ndx, sizes)[1]
            self._pack_one(lib, W, b, cout, cin_w, rot, cin, ndx, bufs, stream)
            return bufs
        key = (id(W), id(b), cout, cin_w, rot, cin, ndx, direct, W.data_ptr(), b.data_ptr())
        e = self.entries.get(key)
        if e is not None and (e["W"]() is not W or e["b"]() is not b):
            self._drop(key)        # the id was recycled by another tensor
            e = None
        if e is None:
            # the Parameter got new storage (net.to(dev), param.data = ...): the entries made for its
            # old storage hold dead pointers in their descriptors and must not reach a device table
            self.drop_stale(W, b)
            pk, bufs = self._alloc(W.device, direct, ndx, sizes)
            d = _lib.PackDesc()
            d.W, d.b = W.data_ptr(), b.data_ptr()
            for name, t in zip(("Wp", "Bp", "Wb", "Wg", "Wq", "Wdx"), bufs):
                setattr(d, name, t.data_ptr() if t is not None else None)
            d.C, d.cin_w, d.rot, d.cin, d.ndx = cout, cin_w, rot, cin, ndx
            _lib.check(lib.gridgcn_pack_desc_fill(ctypes.byref(d)), "gridgcn_pack_desc_fill")
            e = dict(W=weakref.ref(W, lambda _r, k=key: self._drop(k)), b=weakref.ref(b), pk=pk,
                     bufs=bufs, fresh=False, ver=None, desc=d)
            self.entries[key] = e
            self.tables.clear()
        if not (e["fresh"] and e["ver"] == (W._version, b._version)):
            self._pack_one(lib, W, b, cout, cin_w, rot, cin, ndx, e["bufs"], stream)
            torch.autograd.graph.increment_version(e["pk"])
        e["fresh"] = False
        return e["bufs"]

    def get_wgb(self, lib, W, b, geo):
        """[4, C0] table of a first point conv whose feature columns are applied on the source points
        (gridgcn_edge_lin0_*): rows 0..2 = W[:, :3]^T (the geo_vec weights; zeros without geo_vec), row 3 =
        bias.  An entry of the module's prepack table like the layer layouts: built by that ONE launch when
        the table has it, by a concat otherwise."""
        C0, cin_w = W.shape

        def build(out=None):
            rows = W.detach()[:, :3].t() if geo else _cached_zeros(3 * C0, W.device).view(3, C0)
            return torch.cat([rows, b.detach()[None]], out=out)

        if not (WGB_PREPACK and isinstance(W, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter)):
            return build()
        key = (id(W), id(b), C0, cin_w, "wgb", bool(geo), 0, False, W.data_ptr(), b.data_ptr())
        e = self.entries.get(key)
        if e is not None and (e["W"]() is not W or e["b"]() is not b):
            self._drop(key)
            e = None
        if e is None:
            self.drop_stale(W, b)
            pk = torch.empty((4, C0), dtype=torch.float32, device=W.device)
            d = _lib.PackDesc()
            d.W, d.b, d.wgb = W.data_ptr(), b.data_ptr(), pk.data_ptr()
            d.C, d.cin_w, d.rot, d.cin, d.ndx, d.geo = C0, cin_w, 0, cin_w, 0, int(bool(geo))
            _lib.check(lib.gridgcn_pack_desc_fill(ctypes.byref(d)), "gridgcn_pack_desc_fill")
            e = dict(W=weakref.ref(W, lambda _r, k=key: self._drop(k)), b=weakref.ref(b), pk=pk,
                     bufs=(pk,), fresh=False, ver=None, desc=d)
            self.entries[key] = e
            self.tables.clear()
        if not (e["fresh"] and e["ver"] == (W._version, b._version)):
            build(out=e["pk"])
            torch.autograd.graph.increment_version(e["pk"])
        e["fresh"] = False
        return e["pk"]

    def drop_stale(self, W, b):
        """forget every entry of (W, b) whose recorded storage is no longer the live one"""
        live = (W.data_ptr(), b.data_ptr())
        for k in [k for k in self.entries if k[0] == id(W) and k[1] == id(b) and k[8:10] != live]:
            self._drop(k)

    def _live(self, k):
        e = self.entries[k]
        W, b = e["W"](), e["b"]()
        return W is not None and b is not None and (W.data_ptr(), b.data_ptr()) == k[8:10]

    @staticmethod
    def _alloc(dev, direct, ndx, sizes):
        nwp, ldw, nwb, nwq, nwdx = sizes
        pk = torch.empty(nwp + ldw + 2 * nwb + nwq + nwdx, dtype=torch.float32, device=dev)
        o = nwp + ldw
        return pk, (None if direct else pk[:nwp], pk[nwp:nwp + ldw], pk[o:o + nwb],
                    pk[o + nwb:o + 2 * nwb], pk[o + 2 * nwb:o + 2 * nwb + nwq] if direct else None,
                    pk[o + 2 * nwb + nwq:] if ndx else None)

    @staticmethod
    def _pack_one(lib, W, b, cout, cin_w, rot, cin, ndx, bufs, stream):
        p = lambda t: _ptr(t) if t is not None else None   # noqa: E731
        rc = lib.gridgcn_pack_linear(_ptr(W.detach()), _ptr(b.detach()), cout, cin_w, rot, cin, ndx,
                                     *[p(t) for t in bufs], stream)
        _lib.check(rc, "gridgcn_pack_linear")

    def prepack(self, module):
        """one launch for the layouts of every entry that belongs to `module`'s parameters"""
        t = self.tables.get(id(module))
        if t is not None and t[0]() is not module:
            t = None
        if t is None:
            ids = {id(p) for p in module.parameters()}
            for k in [k for k in self.entries if k[0] in ids and k[1] in ids and not self._live(k)]:
                self._drop(k)      # (storage moved since the entry was made: dead pointers)
            keys = [k for k in self.entries if k[0] in ids and k[1] in ids]
            if len(keys) < 2:
                return
            if torch.cuda.is_current_stream_capturing():
                return             # (a host-to-device copy; the lookups pack per layer instead)
            arr = (_lib.PackDesc * len(keys))(*[self.entries[k]["desc"] for k in keys])
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            dev = self.entries[keys[0]]["pk"].device
            assert all(self.entries[k]["pk"].device == dev for k in keys)
            t = (weakref.ref(module), keys, raw.to(dev),
                 max(self.entries[k]["desc"].n for k in keys))
            self.tables[id(module)] = t
        _, keys, table, max_n = t
        if not all(k in self.entries and self._live(k) for k in keys):
            del self.tables[id(module)]     # a parameter moved under a cached table: rebuild it
            return self.prepack(module)
        with torch.cuda.device(table.device):
            rc = _lib.load().gridgcn_pack_linear_batch(
                table.data_ptr(), len(keys), max_n, torch.cuda.current_stream(table.device).cuda_stream)
        _lib.check(rc, "gridgcn_pack_linear_batch")
        for k in keys:
            e = self.entries[k]
            e["fresh"], e["ver"] = True, (e["W"]()._version, e["b"]()._version)
            torch.autograd.graph.increment_version(e["pk"])


    def release(self, module):
        """end of the module's forward: layouts that no layer looked up do not stay marked fresh"""
        t = self.tables.get(id(module))
        if t is not None:
            for k in t[1]:
                e = self.entries.get(k)
                if e is not None:
                    e["fresh"] = False


PACKS = _PackCache()


class LaunchTimers:
    """Device time of selected library calls INSIDE a running training step (bench.py: `ms_in_step`).  A
    micro-benchmark launches a kernel back to back on random tensors with warm caches; the step pays for it
    behind other kernels' traffic (0.717 ms in the micro-benchmark, 0.823 ms in the traced step).
    With `train_ops.TIMERS = LaunchTimers({key, ...})` set, the chain code brackets every matching call --
    key = ("linear_fwd" | "linear_bwd", rows, cin, cout) -- by a pair of events on the launch stream (eager
    steps only: events cannot be read back from a graph replay).  median(key) -> ms."""

    def __init__(self, keys):
        self.ev = {k: [] for k in keys}

    def bracket(self, key):
        lst = self.ev.get(key)
        if lst is None:
            return None
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        lst.append(pair)
        pair[0].record()
        return pair[1]

    def median(self, key, skip=0):
        ts = sorted(a.elapsed_time(b) for a, b in self.ev[key][skip:])
        return ts[len(ts) // 2] if ts else None


TIMERS = None


def release_packs_hook(module, _inputs, _output):
    PACKS.release(module)


class RawLink:
    """Hand-off between the PRODUCER of a raw (pre-BatchNorm) layer output written into the left `n`
    columns of a wider [E, total] buffer and the chain that CONSUMES the buffer (update_func's
    concat(centre features, aggregate), gcn_module_g_att.py:279-283): the consumer applies the
    producer's BatchNorm+ReLU while it loads (columns >= n: scale 1, shift 0 -- the aggregate is a max
    of products of ReLU outputs, so the ReLU is the identity there) and accumulates the producer's
    BatchNorm-backward sums in the epilogue of its input-gradient kernel.  Neither the activated copy
    of the producer's output nor a separate reduce pass over its gradient exists.
    vec [4, total]: scale, shift, mean, rstd per column (left part written by the producer's
    BatchNorm bookkeeping); sums: fp64 [2, total], left by the consumer's backward."""

    def __init__(self, n, total, device):
        self.n, self.total = n, total
        # (the identity entries of the columns >= n are written by the producer's BatchNorm finalisation
        #  launch: gridgcn_bn_finalize_tail)
        self.vec = torch.empty((4, total), dtype=torch.float32, device=device)
        self.sums = None

    def prev_bn(self):
        return (self.vec[0], self.vec[1], self.vec[2], self.vec[3])

    def take_sums(self, psums, cin):
        """the consumer's psums buffer as the producer's [2, n] table: with nbn() the dX epilogue wrote it
        with row stride n (contiguous, no copy); otherwise [2, total], of which the producer slices its part"""
        n = self.nbn()
        return psums[:2 * n].view(2, n) if n else psums.view(2, cin)

    def nbn(self):
        """input columns of the consumer that carry the producer's BatchNorm, as the dX kernel wants
        them (whole 32-column tiles; 0 = all)"""
        return self.n if self.n % 32 == 0 else 0


def _chain_forward(lib, x, params, bns, eps, rot=0, ndx0=0, prev_bn=None, out_raw=None, last_vec=None,
                   z16_last=False):
    """x [E,cin] contiguous; params = (W, b, gamma, beta) per layer.  Per layer three launches:
    pack the operand layouts of W, the MFMA kernel, the BatchNorm bookkeeping.
    x may be wider than the first layer's weight (zero padded columns) and hold the layer's first
    `rot` input channels behind the others (ops.edge_inputs_rows).  prev_bn = (scale, shift): x is
    the raw (pre-BatchNorm) output of an earlier layer whose BatchNorm+ReLU is applied on the fly.
    out_raw: [E, cout_last] destination of the LAST layer's raw output (row stride >= cout_last: the
    left columns of a wider buffer), last_vec: [4, >= cout_last] destination of its BatchNorm vectors
    (RawLink).  z16_last: the last layer's raw output is stored as bf16 (callers check that every
    reader of it takes that: Z16_STORAGE)."""
    L = len(params) // 4
    E, dev = x.shape[0], x.device
    st = _Chain()
    prev, pscale, pshift = x, None, None
    if prev_bn is not None:
        pscale, pshift = prev_bn
    couts = [params[4 * l].shape[0] for l in range(L)]
    # (+ one 8-byte slot per layer: the arrival ticket of the folded BatchNorm finalisation)
    allsums = _zeros(2 * sum(couts) + L, torch.float64, dev)
    tickets = allsums[2 * sum(couts):]
    so = 0
    stream = _stream(x)
    for l in range(L):
        W, b, gamma, beta = params[4 * l:4 * l + 4]
        cout, cin_w = W.shape
        cin = prev.shape[1]
        assert cin >= cin_w and (l == 0 or cin == cin_w)
        K, ldw, nwp, nwb = packed_sizes(cout, cin)
        direct = DIRECT_FWD and cin % 8 == 0
        nwq = cin * ldw if direct else 0
        # input gradient: all columns of a hidden layer, the first ndx0 of the chain input
        ndx = cin if l > 0 else ndx0
        if not (DIRECT_DX and cout % 8 == 0 and 0 < ndx <= 256):
            ndx = 0
        nt = (ndx + 31) // 32
        nwdx = cout * 32 * (1 if nt <= 1 else 2 if nt <= 2 else 4 if nt <= 4 else 8) if ndx else 0
        Wp, Bp, Wb, Wg, Wq, Wdx = PACKS.get(lib, W, b, cout, cin_w, rot if l == 0 else 0, cin, ndx,
                                            direct, (nwp, ldw, nwb, nwq, nwdx), stream)
        st.Wdx.append(Wdx if ndx else Wb)
        st.ndx.append(ndx)
        last = l == L - 1
        if last and out_raw is not None:
            Z = out_raw
            assert Z.shape == (E, cout) and Z.stride(1) == 1
        else:
            Z = torch.empty((E, cout), dtype=torch.bfloat16 if (last and z16_last) else torch.float32,
                            device=dev)
        zfmt = 1 if Z.dtype == torch.bfloat16 else 0
        assert not zfmt or direct
        ldz = Z.stride(0) if Z.stride(0) != cout else 0
        sums = allsums[so:so + 2 * cout]
        so += 2 * cout
        ps = _ptr(pscale) if pscale is not None else None
        ph = _ptr(pshift) if pshift is not None else None
        t_end = TIMERS.bracket(("linear_fwd", E, cin, cout)) if TIMERS is not None else None
        if last and last_vec is not None:
            vec = last_vec[:, :cout]            # rows of the link's [4, total] table
        else:
            vec = torch.empty((4, cout), dtype=torch.float32, device=dev)
        bn = bns[l]
        track = bn is not None and bn.track_running_stats
        tail = last_vec.shape[1] - cout if (last and last_vec is not None) else 0
        folded = direct and FOLD_FINALIZE
        if folded:
            # the BatchNorm bookkeeping by the kernel's last workgroup (no launch of its own)
            fin = _lib.BnFin()
            fin.gamma, fin.beta = gamma.data_ptr(), beta.data_ptr()
            fin.scale, fin.shift, fin.mean, fin.rstd = (vec[0].data_ptr(), vec[1].data_ptr(),
                                                        vec[2].data_ptr(), vec[3].data_ptr())
            fin.running_mean = bn.running_mean.data_ptr() if track else None
            fin.running_var = bn.running_var.data_ptr() if track else None
            fin.num_batches_tracked = bn.num_batches_tracked.data_ptr() if track else None
            fin.ticket = tickets[l].data_ptr()
            fin.eps, fin.momentum, fin.tail = eps, (_momentum(bn) if track else 0.0), tail
            rc = lib.gridgcn_linear_fwd_direct_fin(_ptr(prev), E, cin, cin, _ptr(Wq), _ptr(Bp), ldw,
                                                   cout, ps, ph, _ptr(Z), _ptr(sums), ldz, zfmt,
                                                   ctypes.byref(fin), stream)
        elif direct:
            rc = lib.gridgcn_linear_fwd_direct_ld(_ptr(prev), E, cin, cin, _ptr(Wq), _ptr(Bp), ldw,
                                                  cout, ps, ph, _ptr(Z), _ptr(sums), ldz, zfmt,
                                                  stream)
        else:
            rc = lib.gridgcn_linear_fwd_ld(_ptr(prev), E, cin, _ptr(Wp), _ptr(Bp), K, ldw, cout,
                                           ps, ph, _ptr(Z), _ptr(sums), ldz, stream)
        if t_end is not None:
            t_end.record()
        _lib.check(rc, "gridgcn_linear_fwd")
        if track:
            _stats_written(bn)
        if not folded:
            # (round 2 folded this with returning fp64 atomics, a fenced ticket and a device-scope read-back:
            #  0.1 ms SLOWER over 31 layers; the fold above drains relaxed atomics instead)
            rc = lib.gridgcn_bn_finalize_tail(
                _ptr(sums), _ptr(gamma.detach()), _ptr(beta.detach()), E, eps,
                _momentum(bn) if track else 0.0, cout, tail,
                _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]),
                _ptr(vec[3]), _ptr(bn.running_mean) if track else None,
                _ptr(bn.running_var) if track else None,
                _ptr(bn.num_batches_tracked) if track else None, stream)
            _lib.check(rc, "gridgcn_bn_finalize")
            if track:
                _stats_written(bn)
        st.Z.append(Z); st.scale.append(vec[0]); st.shift.append(vec[1])
        st.mean.append(vec[2]); st.rstd.append(vec[3])
        st.Wb.append(Wb); st.Wg.append(Wg)
        prev, pscale, pshift = Z, vec[0], vec[1]
    return st


def _chain_backward(lib, x, Zs, scales, shifts, means, rstds, Wbs, Wgs, Wdxs, ndxs, sums, dY, sparse,
                    need_dx, cin_w0=None, rot=0, prev_bn=None, nbn=0):
    """backward through a chain.  `sums` [2*C_L] fp64 = BatchNorm-backward sums of the LAST layer;
    upstream gradient either dense dY [E,C_L] or sparse = (amax, gval, P).  Returns (dX, grads)
    with grads = [dW, db, dgamma, dbeta] * L.  cin_w0 / rot: width of the first layer's weight and
    its column rotation when x is in the padded row layout (see _chain_forward).
    prev_bn = (scale, shift, mean, rstd) of an earlier layer whose raw output is x: dX is then the
    gradient w.r.t. relu(bn(x)) and a third value is returned, the BatchNorm-backward sums of x."""
    L = len(Zs)
    if cin_w0 is None:
        cin_w0 = x.shape[1]
    E, dev = x.shape[0], x.device
    grads = [None] * (4 * L)
    Cs = [Zs[l].shape[1] for l in range(L)]
    cins = [x.shape[1]] + Cs[:-1]
    # one zero fill for the chain: BatchNorm-backward sums of layers 0..L-2 (fp64) + bias gradients
    nps = 2 * sum(Cs[:-1]) + (2 * x.shape[1] if prev_bn is not None else 0)
    zbuf = _zeros(nps * 8 + 4 * sum(Cs), torch.uint8, dev)
    zps = zbuf[:nps * 8].view(torch.float64)
    zdb = zbuf[nps * 8:].view(torch.float32)
    po, bo = 0, 0
    for l in range(L - 1, -1, -1):
        Z, C, cin = Zs[l], Cs[l], cins[l]
        # m1, m2, dgamma, dbeta: written by the layer's own backward kernels from `sums`
        # (gridgcn_linear_bwd_fin: no finalisation launch)
        v = torch.empty((4, C), dtype=torch.float32, device=dev)
        m1, m2 = v[0], v[1]
        grads[4 * l + 2] = v[2]                             # d gamma
        grads[4 * l + 3] = v[3]                             # d beta
        # the conv bias feeds a BatchNorm: its gradient is sum(dZ) == 0 analytically
        grads[4 * l + 1] = zdb[bo:bo + C]
        bo += C
        want_dx = l > 0 or need_dx
        dX = torch.empty((E, cin), dtype=torch.float32, device=dev) if want_dx else None
        psums = None
        if l > 0 or prev_bn is not None:
            psums = zps[po:po + 2 * cin]
            po += 2 * cin
        # written in the framework layout (padding dropped, rotated columns moved back)
        cw, rt = (cin_w0, rot) if l == 0 else (cin, 0)
        dW = torch.empty((C, cw), dtype=torch.float32, device=dev)
        Wb = Wbs[l]
        Wg = Wgs[l] if want_dx else None
        nbytes = ctypes.c_size_t(0)
        lib.gridgcn_linear_bwd_workspace_bytes(E, cin, C, ctypes.byref(nbytes))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        prev = Zs[l - 1] if l > 0 else x
        if l > 0:
            pbn = (scales[l - 1], shifts[l - 1], means[l - 1], rstds[l - 1])
        else:
            pbn = prev_bn
        pbn = [_ptr(t) for t in pbn] if pbn is not None else [None] * 4
        if sparse is not None:
            amax, gval, P = sparse
            sp = (_ptr(amax), _ptr(gval), int(P))
            dyp = None
        else:
            sp = (None, None, 0)
            dyp = _ptr(dY)
        t_end = TIMERS.bracket(("linear_bwd", E, cin, C)) if TIMERS is not None else None
        rc = lib.gridgcn_linear_bwd_fin(
            dyp, _ptr(Z), _ptr(scales[l]), _ptr(shifts[l]), _ptr(means[l]), _ptr(rstds[l]),
            _ptr(sums), _ptr(m1), _ptr(m2), _ptr(v[2]), _ptr(v[3]),
            _ptr(prev), pbn[0], pbn[1], pbn[2], pbn[3],
            _ptr(Wb), _ptr(Wg) if Wg is not None else None,
            _ptr(Wdxs[l]) if (want_dx and ndxs[l]) else None, ndxs[l], E, C, cin, cw, rt,
            dY.stride(0) if (sparse is None and dY is not None) else 0,
            Z.stride(0) if Z.stride(0) != C else 0, nbn if l == 0 else 0,
            1 if Z.dtype == torch.bfloat16 else 0,
            _ptr(dX) if want_dx else None, _ptr(dW),
            _ptr(psums) if psums is not None else None, sp[0], sp[1], sp[2],
            _ptr(ws), nbytes.value, _stream(x))
        if t_end is not None:
            t_end.record()
        _lib.check(rc, "gridgcn_linear_bwd")
        grads[4 * l] = dW
        dY, sums, sparse = dX, psums, None
    if prev_bn is not None:
        return dY, grads, sums
    return dY, grads


SMALL_GEMM = True      # the source-point products on csrc/gridgcn_gemm.hip instead of rocBLAS
_GEMM_WS = {}


def _gemm_small(mode, A, B, C, M, N, K, zero_left=0):
    """gridgcn_gemm_small on 2-D views with unit inner stride: mode 0 A[M,K] B[N,K]^T, 1 A[M,K] B[K,N],
    2 A[K,M]^T B[K,N]; C is written in place (any row stride)."""
    assert A.stride(1) == 1 and B.stride(1) == 1 and C.stride(1) == 1
    lib = _lib.load()
    ws, nb = None, 0
    if mode == 2:
        n = ctypes.c_size_t(0)
        lib.gridgcn_gemm_small_workspace_bytes(M, N, K, ctypes.byref(n))
        nb = n.value
        # (tickets at the end of the buffer: zero at first use, left zero by the kernel -- one buffer per
        #  shape and stream, dropped around a graph capture like the zero arena)
        key = (str(A.device), nb, torch.cuda.current_stream(A.device).cuda_stream)
        ws = _GEMM_WS.get(key)
        if ws is None:
            # (the library bounds a workspace at ~16 MB whatever the row count; at most 32 of them are kept)
            if len(_GEMM_WS) >= 32:
                _GEMM_WS.clear()
            ws = _GEMM_WS[key] = torch.zeros(nb, dtype=torch.uint8, device=A.device)
    rc = lib.gridgcn_gemm_small(mode, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C), C.stride(0),
                                M, N, K, zero_left, _ptr(ws) if ws is not None else None, nb, _stream(A))
    _lib.check(rc, "gridgcn_gemm_small")
    return C


def _small_ok(R, *dims):
    return SMALL_GEMM and R <= 65536 and all(0 < d <= 512 for d in dims)


def _mm_nt(a, b, bias=None, out=None):
    """a [M,K] x b [N,K]^T (+ bias [N]) -> [M,N] on csrc/gridgcn_gemm.hip: the products beside the edge pipeline
    whatever their shape (any K, any row strides) -- nothing of a training or evaluation step goes to rocBLAS.
    Not a throughput kernel (one wave per 32 x 32 tile, operands straight from memory): the large layers never
    come here (the register-direct kernels take them, _WideLayerTrain included)."""
    M, K = a.shape
    N = b.shape[0]
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.stride(1) == 1 and b.shape[1] == K
    rc = _lib.load().gridgcn_gemm_bias(0, _ptr(a), a.stride(0), _ptr(b), b.stride(0),
                                       _ptr(bias.detach().contiguous()) if bias is not None else None, _ptr(out),
                                       out.stride(0), M, N, K, _stream(a))
    _lib.check(rc, "gridgcn_gemm_bias")
    return out


def _mm_nn(a, b, out=None):
    """a [M,K] x b [K,N] -> [M,N] (see _mm_nt)"""
    M, K = a.shape
    N = b.shape[1]
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.stride(1) == 1 and b.shape[0] == K
    rc = _lib.load().gridgcn_gemm_bias(1, _ptr(a), a.stride(0), _ptr(b), b.stride(0), None, _ptr(out),
                                       out.stride(0), M, N, K, _stream(a))
    _lib.check(rc, "gridgcn_gemm_bias")
    return out


def _tn_matmul(a, b, out=None):
    """a^T b for tall operands a [R,m], b [R,n] with small m, n: the contraction is cut into
    128-row slabs (one batched GEMM + a sum) so that the work spreads over the chip -- a plain
    [m,R]x[R,n] GEMM runs on m*n/tile workgroups only.  out: optional (strided) destination."""
    R = a.shape[0]
    if a.is_cuda and a.dtype == torch.float32 and SMALL_GEMM and (out is None or out.stride(1) == 1):
        if a.stride(1) != 1:
            a = a.contiguous()
        if b.stride(1) != 1:
            b = b.contiguous()
        if out is None:
            out = torch.empty((a.shape[1], b.shape[1]), dtype=torch.float32, device=a.device)
        return _gemm_small(2, a, b, out, a.shape[1], b.shape[1], R)
    if R >= 1024 and R % 128 == 0:
        S = R // 128
        prod = torch.bmm(a.view(S, 128, a.shape[1]).transpose(1, 2), b.view(S, 128, b.shape[1]))
        return torch.sum(prod, dim=0, out=out) if out is not None else prod.sum(0)
    if out is not None:
        return out.copy_(torch.matmul(a.t(), b))
    return torch.matmul(a.t(), b)


class _ZeroArena:
    """Small zero-filled accumulators (BatchNorm sums, bias gradients) carved from 4 MB zero chunks:
    one fill per chunk instead of one ~3 us launch per buffer (about 70 per training step).  Slices
    are handed out once and never reused; a chunk is freed when its last slice dies.  Each slice is a
    fresh tensor over the chunk's storage (no view relation, own autograd version counter)."""
    CHUNK = 1 << 22
    LIMIT = 1 << 16

    def __init__(self):
        self.chunk, self.off = {}, {}

    def reset(self):
        _GEMM_WS.clear()          # (same reason: a buffer born inside a capture belongs to that graph)
        """Forget the current chunks (live slices keep theirs alive).  A chunk allocated while a
        hipGraph is being captured lives in THAT graph's memory pool and is only re-zeroed by
        that graph's replay: nothing captured or run later may carve slices out of it
        (graph.GraphedTrainStep calls this around every capture)."""
        self.chunk.clear()
        self.off.clear()

    def zeros(self, shape, dtype, dev):
        if isinstance(shape, int):
            shape = (shape,)
        n = 1
        for d in shape:
            n *= d
        item = torch.empty(0, dtype=dtype).element_size()
        nbytes = n * item
        if nbytes > self.LIMIT or not ZERO_ARENA:
            return torch.zeros(shape, dtype=dtype, device=dev)
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
        need = (nbytes + 255) & ~255
        if key not in self.chunk or self.off[key] + need > self.CHUNK:
            self.chunk[key] = torch.zeros(self.CHUNK, dtype=torch.uint8, device=dev)
            self.off[key] = 0
        o = self.off[key]
        self.off[key] = o + need
        strides, acc = [], 1
        for d in reversed(shape):
            strides.append(acc)
            acc *= d
        return torch.empty(0, dtype=dtype, device=dev).set_(
            self.chunk[key].untyped_storage(), o // item, tuple(shape), tuple(reversed(strides)))


ZERO_ARENA = True
# the optimizer of bench.py / the tests' training loops: grid_gcn_amd.optim.Adam (one launch)
OWN_ADAM = True
# BatchNorm finalisation of a conv layer by the forward kernel's last workgroup instead of a launch of its own
FOLD_FINALIZE = True
# BatchNorm statistics of a single-layer point MLP from per-source counts and geo_vec sums
SRC_STATS = True
FUSE_DROPOUT = True     # head: Dropout evaluated inside fc2's forward / dW kernels (no dropped tensor)
SRC_STATS_MIN_EDGES = 1 << 19     # (below: the edge pass is a 10-20 us launch, these are two)
# geo_vec weight + bias table of the source-side first conv built by the prepack launch
WGB_PREPACK = True
# concat + centre mask + zero padding of a layer boundary in one launch (model.GGCNSeg.forward)
GLUE_KERNELS = True
# evaluation of single-layer-pt edge blocks (the up layers) through the source-side kernels
SRC_EVAL = True
ATT_MAX_EVAL = True
_ARENA = _ZeroArena()
_zeros = _ARENA.zeros
reset_zero_arena = _ARENA.reset
_ZEROS = {}


def _cached_zeros(n, dev):
    """a read-only zero vector"""
    key = (n, str(dev))
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=dev)
    return _ZEROS[key]


def alias_columns(buf, col0, ncol):
    """A fresh tensor (no autograd / view relation) over columns [col0, col0+ncol) of the contiguous
    2-D buffer `buf`: lets two producers write the halves of a concatenation in place."""
    E, ld = buf.shape
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(
        buf.untyped_storage(), buf.storage_offset() + col0, (E, ncol), (ld, 1))


class _Cat2(torch.autograd.Function):
    """concat([a, b], -1) where a and b already ARE the two halves of `full` (alias_columns)."""

    @staticmethod
    def forward(ctx, a, b, full):
        ctx.ca = a.shape[-1]
        return alias_columns(full, 0, full.shape[1])

    @staticmethod
    def backward(ctx, g):
        return g[..., :ctx.ca], g[..., ctx.ca:], None


def _rows2d(t):
    """(tensor, row stride in floats) of a [..., W] float32 tensor seen as rows of W floats -- without a copy
    when the rows are regularly strided (a gradient that is a column slice of a wider buffer)"""
    W = t.shape[-1]
    if t.is_contiguous():
        return t, W
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= W:
        return t, t.stride(0)
    if t.dim() == 3 and t.stride(2) == 1 and t.stride(1) >= W and \
            (t.shape[0] == 1 or t.stride(0) == t.shape[1] * t.stride(1)):
        return t, t.stride(1)
    return t.contiguous(), W


class _CatMask(torch.autograd.Function):
    """(concat([a, b * mask[..., None]], -1), the same rows zero-padded to a multiple of 8 floats) in ONE
    launch: data_layer = concat(cent, features * centmsk) of the layer boundary
    (segmentation/models/ggcn_models_g.py:186, gcn_module_g_att.py:284-285) and the copy of it the centre
    MLP of the up path reads (its first layer's register-direct kernels want rows of whole 32-byte
    pieces).  b None: a column of ones (ggcn_models_g.py:137, data = concat(xyz, 1)).  The backward adds the
    gradients of the two outputs and applies the mask in one launch as well.  a carries no gradient (the
    index operators' centres)."""

    @staticmethod
    def forward(ctx, a, b, mask, pad):
        lib = _lib.load()
        a = a.contiguous()
        lead, ca = a.shape[:-1], a.shape[-1]
        E = a.numel() // ca
        dev = a.device
        if b is not None:
            b = b.contiguous()
            cb = b.shape[-1]
            assert b.shape[:-1] == lead
        else:
            cb = 1
        if mask is not None:
            mask = mask.contiguous()
            assert mask.numel() == E and mask.dtype == torch.float32
        W = ca + cb
        W8 = (W + 7) & ~7
        out = torch.empty(lead + (W,), dtype=torch.float32, device=dev)
        out2 = torch.empty(lead + (W8,), dtype=torch.float32, device=dev) if (pad and W8 != W) else None
        with torch.cuda.device(dev):
            rc = lib.gridgcn_cat_mask(_ptr(a), ca, ca, _ptr(b) if b is not None else None, cb, cb,
                                      _ptr(mask) if mask is not None else None, _ptr(out), W,
                                      _ptr(out2) if out2 is not None else None, W8, E, _stream(a))
        _lib.check(rc, "gridgcn_cat_mask")
        ctx.dims = (ca, cb, E, b is not None)
        ctx.save_for_backward(mask)
        ctx.set_materialize_grads(False)
        if out2 is None:
            return out, None
        return out, out2

    @staticmethod
    def backward(ctx, g1, g2):
        ca, cb, E, has_b = ctx.dims
        if not has_b or (g1 is None and g2 is None) or not ctx.needs_input_grad[1]:
            return None, None, None, None
        lib = _lib.load()
        (mask,) = ctx.saved_tensors
        g = g1 if g1 is not None else g2
        l1 = l2 = 0
        if g1 is not None:
            g1, l1 = _rows2d(g1)
        if g2 is not None:
            g2, l2 = _rows2d(g2)
        db = torch.empty(g.shape[:-1] + (cb,), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.gridgcn_mask_sum(_ptr(g1) if g1 is not None else None, l1,
                                      _ptr(g2) if g2 is not None else None, l2, ca, cb,
                                      _ptr(mask) if mask is not None else None, _ptr(db), E, _stream(g))
        _lib.check(rc, "gridgcn_mask_sum")
        return None, db, None, None


def cat_mask(a, b, mask=None, pad=False):
    """-> (concat([a, b * mask], -1), zero-padded copy or the same tensor); float32 GPU tensors."""
    out, out2 = _CatMask.apply(a, b, mask, pad)
    return out, (out2 if out2 is not None else out)


def _dw_direct_ok(C, cin):
    """mirror of gg_dw_direct_cfg (csrc/gridgcn_direct.hip): shapes the register-direct dW kernel
    takes (the dX kernel additionally needs C % 8 == 0, i.e. a packed Wdx)."""
    if cin > 320 or cin % 4 or C > 256 or C % 8:
        return False
    nq, rem = cin // 128, cin % 128
    np_ = 1 if rem >= 64 else 0
    rem -= 64 * np_
    if rem > 32:
        return False
    nj = 4 * nq + 2 * np_ + (1 if rem else 0)
    mt = 2 if (C >= 64 and nj <= 5) else 1
    return mt * nj <= 10 and nq <= 2 and (C + 32 * mt - 1) // (32 * mt) <= 8


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
        if DIRECT_FWD and cin0 % 8 and prev is None:
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
                and DIRECT_DX
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
    return (DIRECT_FWD and DIRECT_DX and E >= 4096 and cin % 8 == 0 and cin <= 320 and C % 256 == 0
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


# Evaluation constants of a layer -- BatchNorm folded to (scale, shift) with the running statistics, the packed
# forward operand of the weight -- cached per module and rebuilt when a Tensor._version of what they are made of
# moves.  (They were recomputed at every call: five element-wise framework launches per BatchNorm and one pack
# launch per layer, ~200 launches = 1 ms of a 2.7-ms evaluation forward of the segmentation net.)  Whoever
# writes these tensors through raw pointers moves the counter by hand: the training kernels for the running
# statistics (_stats_written), optim.Adam for the parameters, graph.GraphedTrainStep after every replay.
# (`p.data.op_()` does not move a version counter -- as with gridconv.SubGUpdate.packed_layers, call
# clear_eval_cache() after editing parameters that way.)
EVAL_CACHE = True
_EVAL_BN, _EVAL_W = {}, {}
# Tensor._version alone is not enough: torch's fused / foreach optimizers update the parameters without moving it
# (torch.optim.Adam(fused=True) in an eager loop, then eval() -> the first evaluation's packed weights
# were reused).  Every cache key therefore also carries a process-wide PARAMETER GENERATION, advanced by a global
# optimizer-step hook (any torch.optim.Optimizer, this package's Adam included) and by graph.GraphedTrainStep
# after a replay: whatever may have rewritten a weight since the entry was built makes it stale.  An
# evaluation-only loop never advances it, so it keeps its cache.
_PARAM_GEN = [0]


def params_changed(*_args, **_kw):
    """Declare that parameters / BatchNorm buffers may have been rewritten behind autograd's back."""
    _PARAM_GEN[0] += 1


_torch_optimizer_mod.register_optimizer_step_post_hook(params_changed)


def clear_eval_cache():
    _EVAL_BN.clear()
    _EVAL_W.clear()


def _stats_written(bn):
    """the kernels update the running statistics through raw pointers"""
    torch.autograd.graph.increment_version((bn.running_mean, bn.running_var))


def _bn_eval_vectors(bn):
    """(scale, shift) of a BatchNorm in evaluation mode: y = x * scale + shift"""
    key = (_PARAM_GEN[0], bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.eps)
    e = _EVAL_BN.get(id(bn)) if EVAL_CACHE else None
    if e is not None and e[0]() is bn and e[1] == key:
        return e[2], e[3]
    with torch.no_grad():
        sc = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
        sh = (bn.bias - bn.running_mean * sc).contiguous()
    if EVAL_CACHE:
        _EVAL_BN[id(bn)] = (weakref.ref(bn, lambda _r, k=id(bn): _EVAL_BN.pop(k, None)), key, sc, sh)
    return sc, sh


def _eval_packed(lib, lin, cout_p, cin, st):
    """(Bp, Wq, ldw): zero-padded bias and forward operand of lin.weight for a kernel that sees `cin` input
    columns and cout_p >= out_features output columns (gridgcn_pack_linear)"""
    W, b = lin.weight, lin.bias
    cout, cin_w = W.shape
    key = (_PARAM_GEN[0], cout_p, cin, W._version, b._version, W.data_ptr(), b.data_ptr())
    e = _EVAL_W.get(id(lin)) if EVAL_CACHE else None
    if e is not None and e[0]() is lin and e[1] == key:
        return e[2]
    K, ldw, nwp, nwb = packed_sizes(cout_p, cin)
    pk = torch.empty(ldw + cin * ldw, dtype=torch.float32, device=W.device)
    Bp, Wq = pk[:ldw], pk[ldw:]
    _lib.check(lib.gridgcn_pack_linear(_ptr(W.detach()), _ptr(b.detach()), cout, cin_w, 0, cin, 0, None,
                                       _ptr(Bp), None, None, _ptr(Wq), None, st), "pack")
    if EVAL_CACHE:
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
    if len(pt_layers) != 1 or not SRC_EVAL:
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
        e = _EVAL_W.get(("wgb", id(pt_layer))) if EVAL_CACHE else None
        if e is not None and e[0]() is pt_layer and e[1] == wkey:
            wgb = e[2]
        else:
            wgb = torch.cat([W0[:, :3].t() if geo else _cached_zeros(3 * C0, dev).view(3, C0), b0[None]])
            if EVAL_CACHE:
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
        if ATT_MAX_EVAL and a2.lin.in_features == 32 and C in (64, 128) and P <= 128:
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


def _att_bwd_noz(lib, att16, Z1, aS, aH, aM, aR, aWb, aWg, aWx, ndxs, W2, b2, sums_a, amax, ga, P, cwa, st):
    """backward of the attention chain (10 -> 32 -> 128) of an up layer without the second conv's [E, 128]
    pre-activation: gridgcn_att_bwd_noz for the second conv (dA1, dW2, its BatchNorm vectors, the BatchNorm-
    backward sums of the first layer), then the ordinary chain backward for the first conv.  Returns the
    chain's gradient list [dW, db, dgamma, dbeta] * 2."""
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
    t_end = TIMERS.bracket(("linear_bwd", E, cin, C)) if TIMERS is not None else None
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
            noz = Lp == 1 and NO_Z0 and C0 % 4 == 0
            Z0 = None if noz else torch.empty((E, C0), dtype=torch.float32, device=dev)
            att16 = torch.empty((E, 16), dtype=torch.float32, device=dev)
            sums0 = _zeros(2 * C0, torch.float64, dev)
            gsum = gg = None
            if noz and SRC_STATS and (Nsrc + 1) * 28 <= 150 * 1024 and C0 <= 1024 and E >= SRC_STATS_MIN_EDGES:
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
            z16 = (Z16_STORAGE and noz and La == 2 and A0 in (16, 32) and C in (64, 128) and E >= 32
                   and lib.gridgcn_get_mlp_precision() == 1
                   and lib.gridgcn_get_option(_lib.OPT_ATT_BWD_FUSED) == 1)
            # the backward of the second attention conv needs no Z2 (gridgcn_att_bwd_noz): decided HERE, because
            # the tensor is then not saved ...
            nz = (NOZ_ATT_BWD and noz and La == 2 and not z16 and lib.gridgcn_get_mlp_precision() == 0
                  and A0 == 32 and C == 128 and E >= 32 and att16.shape[1] == 16)
            sa = _chain_forward(lib, att16, pa, bns_a, eps, z16_last=z16)
            if noz:
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
        ctx.geo = gsum is not None      # (per-source geo sums of the forward: the backward's geo pass is skipped)
        saZ = list(sa.Z)
        if nz:
            saZ[-1] = torch.empty(0, dtype=torch.float32, device=dev)     # Z2: read by nobody any more
        ctx.save_for_backward(
            src, nebidx, att16, amax, Ysrc if noz else Z0, vec0, W0, zsel, wgb,
            *sp.Z, *sp.scale, *sp.shift, *sp.mean, *sp.rstd, *sp.Wb, *sp.Wg, *sp.Wdx,
            *saZ, *sa.scale, *sa.shift, *sa.mean, *sa.rstd, *sa.Wb, *sa.Wg, *sa.Wdx,
            *((pa[4], pa[5]) if nz else ()), *((gsum, gg) if gsum is not None else ()))
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
                                       sums_a, amax, ga, P, cwa, st)
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
            if noz and SPARSE_L0 and (Nsrc + 1) * 144 <= 150 * 1024:
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
                    rc = lib.gridgcn_edge_lin0_dwg(
                        _ptr(wgs), _ptr(gg), _ptr(_tn_matmul(Ysrc, Gsum)), _ptr(wgb),
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
    if not (has_feats and src.is_cuda and src.dtype == torch.float32 and SRC_FIRST_CONV):
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
    ndx = (ndx or min(cin, 256)) if (DIRECT_DX and C % 8 == 0) else 0
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
            _lib.check(lib.gridgcn_pack_linear(_ptr(W2d[:, :K12].contiguous()), None, N0, K12, 0,
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
        _lib.check(lib.gridgcn_pack_linear(_ptr(W2[:, :K12].contiguous()), None, N0, K12, 0, K12, 0,
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


def edge_block_cls_train(src, nebidx, cent, pt_layers, att1_layers, att2_layers):
    """[B,O,C] = max_p att2(concat(att1(att_vec), pt_mlp(nf0), ctx)) * pt_mlp(nf0) from
    (src [B,Nsrc,4+Cf], nebidx [B,O,P], cent [B,O,>=3]); training mode."""
    params = []
    for l in list(pt_layers) + list(att1_layers) + list(att2_layers):
        params += [l.lin.weight, l.lin.bias, l.bn.weight, l.bn.bias]
    meta = (pt_layers[0].bn.eps, [l.bn for l in pt_layers], att1_layers[0].bn,
            [l.bn for l in att2_layers])
    return _EdgeBlockClsTrain.apply(src.contiguous(), nebidx, cent.contiguous(), meta, *params)


# ------------------------------------------------------------------------------------------------
# segmentation head: the last linear layer (no BatchNorm / ReLU) and the softmax cross-entropy.
# The class dimension is zero padded to a multiple of 8 inside (logits live in an [E, Cp] buffer and
# the op returns its [:, :C] view), so that the same register-direct MFMA kernels run the layer with
# "identity BatchNorm" constants: scale 1, shift +inf (ReLU mask always open), mean 0, m1 = m2 = 0.
_IDENT = {}


def _identity_consts(Cp, dev):
    key = (Cp, str(dev))
    if key not in _IDENT:
        v = torch.zeros((6, Cp), dtype=torch.float32, device=dev)
        v[0] = 1.0             # scale
        v[1] = float("inf")    # shift
        v[3] = 1.0             # rstd      (v[2] mean, v[4] m1, v[5] m2 stay 0)
        _IDENT[key] = v
    return _IDENT[key]


def linear_plain_supported(x, lin):
    return (x.is_cuda and x.dtype == torch.float32 and DIRECT_FWD and DIRECT_DX
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
            fuse = (FUSE_DROPOUT and 0.0 < float(p) < 1.0 and C == 128 and E * C < 2 ** 32
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
    return (supported(layers, x) and DIRECT_FWD and DIRECT_DX and lin.bias is not None
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
