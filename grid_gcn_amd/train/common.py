"""Shared state and helpers of the training path: operand packing cache, chain forward / backward
(one conv + BatchNorm + ReLU stack), the small GEMMs, the zero arena, concat glue.  See grid_gcn_amd/train/__init__.py
for the overview; everything goes through the C ABI of include/gridgcn.h."""
import ctypes
import weakref

import torch
from torch.optim import optimizer as _torch_optimizer_mod

from .. import _lib
from ..ops import _ptr, _stream
from .options import OPT

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

        if not (OPT.WGB_PREPACK and isinstance(W, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter)):
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
    With `OPT.TIMERS = LaunchTimers({key, ...})` set, the chain code brackets every matching call --
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
        direct = OPT.DIRECT_FWD and cin % 8 == 0
        nwq = cin * ldw if direct else 0
        # input gradient: all columns of a hidden layer, the first ndx0 of the chain input
        ndx = cin if l > 0 else ndx0
        if not (OPT.DIRECT_DX and cout % 8 == 0 and 0 < ndx <= 256):
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
        t_end = OPT.TIMERS.bracket(("linear_fwd", E, cin, cout)) if OPT.TIMERS is not None else None
        if last and last_vec is not None:
            vec = last_vec[:, :cout]            # rows of the link's [4, total] table
        else:
            vec = torch.empty((4, cout), dtype=torch.float32, device=dev)
        bn = bns[l]
        track = bn is not None and bn.track_running_stats
        tail = last_vec.shape[1] - cout if (last and last_vec is not None) else 0
        folded = direct and OPT.FOLD_FINALIZE
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
        t_end = OPT.TIMERS.bracket(("linear_bwd", E, cin, C)) if OPT.TIMERS is not None else None
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
    return OPT.SMALL_GEMM and R <= 65536 and all(0 < d <= 512 for d in dims)


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
    if a.is_cuda and a.dtype == torch.float32 and OPT.SMALL_GEMM and (out is None or out.stride(1) == 1):
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
        if nbytes > self.LIMIT or not OPT.ZERO_ARENA:
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


def _stats_written(bn):
    """the kernels update the running statistics through raw pointers"""
    torch.autograd.graph.increment_version((bn.running_mean, bn.running_var))


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
