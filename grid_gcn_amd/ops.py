"""Host-side mirror of the reference operator interface, on top of the C ABI (include/gridgcn.h).

Names, argument order, attribute names and output order are those of the MXNet symbols the
reference registers from gridifyop/additional.so:

    mx.sym.Gridify(data, actual_numpoints, max_o_grid, max_p_grid, kernel_size, stride,
                   coord_shift, voxel_size, grid_size, loc)
        -> nebidx, nebidxmsk, cent, centmsk, actual_centnum       gridify-inl.h:146-152
    mx.sym.GridifyKNN(...)      same signature                    gridifyknn-inl.h
    mx.sym.GridifyUp(downdata, updata, down_actual_numpoints, up_actual_numpoints,
                     max_p_grid, max_o_grid, kernel_size, coord_shift, voxel_size, grid_size)
        -> nebidx, nebidxmsk                                      gridify_up-inl.h:137-143
    mx.sym.contrib.BallKNN(unknown, known, downnum, upnum, k=3, radius=0.1) -> idx
                                                                  ball_k_nn.cc:14-66
    mx.sym.contrib.KNN(unknown, known, downnum, upnum, k=3) -> idx           k_nn.cc:14-66
    batch_take_g(data, index, shape)                              utils/ops.py:78-93

The only addition is `seed` (the reference seeds cuRAND from gettimeofday().tv_usec,
gridify.cu:377-379; seed=0 is the fast_apprxmt fixed-seed mode).

Tensors are torch CUDA(HIP) tensors; PyTorch is used for device memory and the current stream
only.  Errors (ndim, dtype, contiguity, device, attribute range) raise RuntimeError, the analogue
of the MXNetError raised by the reference's CHECK_EQ (gridify-inl.h:174-182).  There is no CPU
path: the reference has none either for the grid ops (gridify.cc:28-39 LOG(FATAL)).
All index ops are non-differentiable (gridify-inl.h:227-231, ball_k_nn.cc:60).
"""
import ctypes

import torch

from . import _lib

# backward of the neighbour gather as a sorted segmented sum (csrc/gridgcn_scatter.hip)
SORTED_TAKE_BWD = True
# BallKNN through a cell grid over the known points instead of the all-pairs scan
BALL_GRID = True

def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _chk(t, name, ndim, dtype, last=None):
    _require(isinstance(t, torch.Tensor), "%s must be a torch.Tensor" % name)
    _require(t.is_cuda, "%s must live on the GPU (no CPU path: gridify.cc:28-39)" % name)
    _require(t.dim() == ndim, "%s should be a %dD tensor" % (name, ndim))
    _require(t.dtype == dtype, "%s must be %s" % (name, dtype))
    _require(t.is_contiguous(), "%s must be contiguous" % name)
    if last is not None:
        _require(t.shape[-1] == last, "last dim of %s should be %d" % (name, last))


def _num(t, name, B):
    """actual_numpoints-style input: int32 [B,1] (or [B])."""
    _require(isinstance(t, torch.Tensor) and t.is_cuda, "%s must be a GPU tensor" % name)
    _require(t.dtype == torch.int32, "%s must be int32" % name)
    _require(t.numel() == B and t.is_contiguous(), "%s should be a [B,1] tensor" % name)
    return t


def _params(max_p_grid, max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size, grid_size,
            seed, seed_dev=None):
    p = _lib.GridParams()
    p.max_p_grid, p.max_o_grid, p.kernel_size = int(max_p_grid), int(max_o_grid), int(kernel_size)
    p.stride, p.loc = int(stride), int(loc)
    _require(len(coord_shift) == 3 and len(voxel_size) == 3 and len(grid_size) == 3,
             "coord_shift / voxel_size / grid_size must have 3 entries")
    for j in range(3):
        p.coord_shift[j] = float(coord_shift[j])
        p.voxel_size[j] = float(voxel_size[j])
        p.grid_size[j] = int(grid_size[j])
    p.seed = int(seed) & (2 ** 64 - 1)
    # optional device scalar (int64 tensor) added to the seed when the kernels run: a captured
    # hipGraph then draws a fresh sample at every replay (graph.GraphedTrainStep bumps it)
    if seed_dev is not None:
        _require(isinstance(seed_dev, torch.Tensor) and seed_dev.is_cuda and
                 seed_dev.dtype == torch.int64 and seed_dev.numel() >= 1,
                 "seed_dev must be an int64 GPU tensor")
        p.seed_dev = seed_dev.data_ptr()
    else:
        p.seed_dev = None
    return p


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _workspace(nbytes, device):
    # torch caching allocator: no hipMalloc in steady state, stream-ordered reuse
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def _gridify_like(fn_name, data, actual_numpoints, max_p_grid, max_o_grid, kernel_size, stride,
                  loc, coord_shift, voxel_size, grid_size, seed, seed_dev=None, extra=()):
    lib = _lib.load()
    _chk(data, "data", 3, torch.float32, 4)
    B, N, _ = data.shape
    _num(actual_numpoints, "actual_numpoints", B)
    _require(actual_numpoints.device == data.device, "inputs must be on the same device")
    p = _params(max_p_grid, max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size,
                grid_size, seed, seed_dev)
    nbytes = ctypes.c_size_t(0)
    _lib.check(getattr(lib, fn_name + "_workspace_bytes")(B, N, ctypes.byref(p),
                                                          ctypes.byref(nbytes)), fn_name)
    dev = data.device
    O, P = int(max_o_grid), int(max_p_grid)
    with torch.cuda.device(dev):
        ws = _workspace(nbytes.value, dev)
        nebidx = torch.empty((B, O, P), dtype=torch.int32, device=dev)
        nebidxmsk = torch.empty((B, O, P), dtype=torch.float32, device=dev)
        cent = torch.empty((B, O, 4), dtype=torch.float32, device=dev)
        centmsk = torch.empty((B, O), dtype=torch.float32, device=dev)
        actual_centnum = torch.empty((B, 1), dtype=torch.int32, device=dev)
        rc = getattr(lib, fn_name)(_ptr(data), _ptr(actual_numpoints), B, N, ctypes.byref(p), *extra,
                                   _ptr(nebidx), _ptr(nebidxmsk), _ptr(cent), _ptr(centmsk),
                                   _ptr(actual_centnum), _ptr(ws), nbytes.value, _stream(data))
    _lib.check(rc, fn_name)
    return nebidx, nebidxmsk, cent, centmsk, actual_centnum


@torch.no_grad()
def Gridify(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1, loc=0,
            coord_shift, voxel_size, grid_size, seed=0, seed_dev=None):
    """Coverage-aware grid query with random voxel sampling (RVS).

    data [B,N,4] f32 (x,y,z,w), actual_numpoints [B,1] i32 ->
    nebidx [B,O,P] i32, nebidxmsk [B,O,P] f32, cent [B,O,4] f32, centmsk [B,O] f32,
    actual_centnum [B,1] i32.  Reference: GridifyForward<gpu>, gridify.cu:294-413.
    """
    return _gridify_like("gridgcn_gridify", data, actual_numpoints, max_p_grid, max_o_grid,
                         kernel_size, stride, loc, coord_shift, voxel_size, grid_size, seed,
                         seed_dev)


@torch.no_grad()
def Gridify_occaware(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1, loc=0,
                     coord_shift, voxel_size, grid_size, seed=0, seed_dev=None, beta=1.0):
    """Gridify with Coverage-Aware Sampling of the centre voxels (the paper's CAS; the reference
    registers it as `Gridify_occaware` but ships no source for it: gridifyop/additional.so).
    PARITY UNPINNED -- our own sequential restatement of the paper's section 3.2 (stated in
    include/gridgcn.h and in the tests' C checker, tests/test_cas.py) is what this is checked against.
    Same signature and outputs as Gridify; beta >= 0 weighs the over-coverage penalty of eq. 3."""
    _require(float(beta) >= 0.0, "beta must be >= 0")
    return _gridify_like("gridgcn_gridify_occaware", data, actual_numpoints, max_p_grid, max_o_grid,
                         kernel_size, stride, loc, coord_shift, voxel_size, grid_size, seed,
                         seed_dev, extra=(ctypes.c_float(float(beta)),))


@torch.no_grad()
def Gridify_fast_rand(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1,
                      loc=0, coord_shift, voxel_size, grid_size, seed=0):
    """The `fast_rand` build of Gridify (gridifyop/fast_rand/gridify.cu): scatter-to-k^3 buckets
    with a thread-index-seeded reservoir, the first max_o_grid occupied voxels as centres, own-voxel
    query.  Same signature and outputs as Gridify (`seed` is not used by this variant)."""
    return _gridify_like("gridgcn_gridify_fast_rand", data, actual_numpoints, max_p_grid,
                         max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size, grid_size,
                         seed)


@torch.no_grad()
def gridify_timed(data, actual_numpoints, iters=50, *, max_p_grid, max_o_grid, kernel_size,
                  stride=1, loc=0, coord_shift, voxel_size, grid_size, seed=0):
    """Average device time (ms) of one Gridify call: `iters` back-to-back calls inside the
    library between two HIP events on the launch stream (gridgcn_gridify_timed) -- no Python,
    no allocation between the calls.  Returns (ms_per_call, outputs of the last call)."""
    lib = _lib.load()
    _chk(data, "data", 3, torch.float32, 4)
    B, N, _ = data.shape
    _num(actual_numpoints, "actual_numpoints", B)
    p = _params(max_p_grid, max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size,
                grid_size, seed)
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.gridgcn_gridify_workspace_bytes(B, N, ctypes.byref(p), ctypes.byref(nbytes)),
               "gridgcn_gridify_workspace_bytes")
    dev = data.device
    O, P = int(max_o_grid), int(max_p_grid)
    ms = ctypes.c_float(0.0)
    with torch.cuda.device(dev):
        ws = _workspace(nbytes.value, dev)
        nebidx = torch.empty((B, O, P), dtype=torch.int32, device=dev)
        nebidxmsk = torch.empty((B, O, P), dtype=torch.float32, device=dev)
        cent = torch.empty((B, O, 4), dtype=torch.float32, device=dev)
        centmsk = torch.empty((B, O), dtype=torch.float32, device=dev)
        actual_centnum = torch.empty((B, 1), dtype=torch.int32, device=dev)
        rc = lib.gridgcn_gridify_timed(_ptr(data), _ptr(actual_numpoints), B, N, ctypes.byref(p),
                                       _ptr(nebidx), _ptr(nebidxmsk), _ptr(cent), _ptr(centmsk),
                                       _ptr(actual_centnum), _ptr(ws), nbytes.value,
                                       _stream(data), int(iters), ctypes.byref(ms))
    _lib.check(rc, "gridgcn_gridify_timed")
    return float(ms.value), (nebidx, nebidxmsk, cent, centmsk, actual_centnum)


@torch.no_grad()
def GridifyKNN(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1, loc=0,
               coord_shift, voxel_size, grid_size, seed=0, seed_dev=None):
    """As Gridify, neighbours = exact top-P by distance to the voxel centre
    (GridifyKNNForward<gpu>, gridifyknn.cu:337-455)."""
    return _gridify_like("gridgcn_gridify_knn", data, actual_numpoints, max_p_grid, max_o_grid,
                         kernel_size, stride, loc, coord_shift, voxel_size, grid_size, seed,
                         seed_dev)


@torch.no_grad()
def GridifyUp(downdata, updata, down_actual_numpoints, up_actual_numpoints, *, max_p_grid,
              max_o_grid, kernel_size, coord_shift, voxel_size, grid_size, seed=0, seed_dev=None):
    """downdata [B,Nd,4], updata [B,max_o_grid,4] -> nebidx [B,O,P] i32, nebidxmsk [B,O,P] f32.
    Reference: GridifyUpForward<gpu>, gridify_up.cu:229-323."""
    lib = _lib.load()
    _chk(downdata, "downdata", 3, torch.float32, 4)
    _chk(updata, "updata", 3, torch.float32, 4)
    B, Nd, _ = downdata.shape
    O, P = int(max_o_grid), int(max_p_grid)
    _require(updata.shape[0] == B and updata.shape[1] == O,
             "updata must be [B, max_o_grid, 4] (it is indexed with stride max_o_grid, "
             "gridify_up.cu:196)")
    _num(down_actual_numpoints, "down_actual_numpoints", B)
    _num(up_actual_numpoints, "up_actual_numpoints", B)
    p = _params(max_p_grid, max_o_grid, kernel_size, 1, 0, coord_shift, voxel_size, grid_size, seed,
                seed_dev)
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.gridgcn_gridify_up_workspace_bytes(B, Nd, ctypes.byref(p), ctypes.byref(nbytes)),
               "gridgcn_gridify_up")
    dev = downdata.device
    with torch.cuda.device(dev):
        ws = _workspace(nbytes.value, dev)
        nebidx = torch.empty((B, O, P), dtype=torch.int32, device=dev)
        nebidxmsk = torch.empty((B, O, P), dtype=torch.float32, device=dev)
        rc = lib.gridgcn_gridify_up(_ptr(downdata), _ptr(updata), _ptr(down_actual_numpoints),
                                    _ptr(up_actual_numpoints), B, Nd, ctypes.byref(p),
                                    _ptr(nebidx), _ptr(nebidxmsk), _ptr(ws), nbytes.value,
                                    _stream(downdata))
    _lib.check(rc, "gridgcn_gridify_up")
    return nebidx, nebidxmsk


def _point_rows(t, name):
    """[B, n, 3] coordinates, contiguous OR a column slice of wider contiguous rows (xyz of [B, n, 4+C]
    point rows: the up path hands those over without a copy).  Returns (tensor, floats per row)."""
    _require(isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 3 and t.shape[-1] == 3
             and t.dtype == torch.float32, "%s should be a float32 [B,n,3] GPU tensor" % name)
    if t.is_contiguous():
        return t, 3
    n, ld = t.shape[1], t.stride(1)
    if t.stride(2) == 1 and ld >= 3 and (t.shape[0] == 1 or t.stride(0) == n * ld):
        return t, ld
    return t.contiguous(), 3


def _knn_common(ball, unknown, known, downnum, upnum, k, radius, out):
    lib = _lib.load()
    n = unknown.shape[1] if (isinstance(unknown, torch.Tensor) and unknown.dim() == 3) else 0
    grid_ok = (ball and BALL_GRID and int(k) <= 6 and isinstance(known, torch.Tensor) and known.dim() == 3
               and known.shape[1] >= 64
               and n * known.shape[1] >= (1 << 16) and radius >= 0)
    rows_ok = (ball and int(k) <= 6 and all(isinstance(t, torch.Tensor) and t.dim() == 3 and t.is_cuda
                                            and t.dtype == torch.float32 and t.shape[-1] == 3
                                            for t in (unknown, known)))
    grid_ok = grid_ok and rows_ok
    if rows_ok:
        unknown, ldu = _point_rows(unknown, "unknown")
        known, ldk = _point_rows(known, "known")
    else:
        _chk(unknown, "unknown", 3, torch.float32, 3)   # ball_k_nn.cc:36-42
        _chk(known, "known", 3, torch.float32, 3)
    B, n, _ = unknown.shape
    _require(known.shape[0] == B, "unknown and known must have the same batch size")
    m = known.shape[1]
    _num(downnum, "downnum", B)
    _num(upnum, "upnum", B)
    dev = unknown.device
    ztail = 0
    if out is None:
        # the reference leaves rows >= upnum[b] untouched (undefined memory); zero them here -- inside
        # the query kernel where it can (one fill less per call)
        if rows_ok:
            out, ztail = torch.empty((B, n, int(k)), dtype=torch.int32, device=dev), 1
        else:
            out = torch.zeros((B, n, int(k)), dtype=torch.int32, device=dev)
    else:
        _chk(out, "out", 3, torch.int32, int(k))
    with torch.cuda.device(dev):
        if grid_ok:
            # same indices through a cell grid over the known points (csrc/gridgcn_ballgrid.hip)
            nb = ctypes.c_size_t(0)
            lib.gridgcn_ball_knn_grid_workspace_bytes(B, m, ctypes.byref(nb))
            ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
            rc = lib.gridgcn_ball_knn_grid_ld(_ptr(unknown), ldu, _ptr(known), ldk, _ptr(downnum),
                                              _ptr(upnum), B, n, m, int(k), ctypes.c_float(radius), ztail,
                                              _ptr(out), _ptr(ws), nb.value, _stream(unknown))
        elif rows_ok:
            rc = lib.gridgcn_ball_knn_ld(_ptr(unknown), ldu, _ptr(known), ldk, _ptr(downnum), _ptr(upnum),
                                         B, n, m, int(k), ctypes.c_float(radius), ztail, _ptr(out),
                                         _stream(unknown))
        elif ball:
            rc = lib.gridgcn_ball_knn(_ptr(unknown), _ptr(known), _ptr(downnum), _ptr(upnum), B, n,
                                      m, int(k), ctypes.c_float(radius), _ptr(out),
                                      _stream(unknown))
        else:
            rc = lib.gridgcn_knn(_ptr(unknown), _ptr(known), _ptr(downnum), _ptr(upnum), B, n, m,
                                 int(k), _ptr(out), _stream(unknown))
    _lib.check(rc, "gridgcn_ball_knn" if ball else "gridgcn_knn")
    return out


@torch.no_grad()
def BallKNN(unknown, known, downnum, upnum, *, k=3, radius=0.1, out=None):
    """unknown [B,n,3], known [B,m,3], downnum/upnum [B,1] i32 -> idx [B,n,k] i32 (-1 = none).
    Reference: _contrib_BallKNN, ball_k_nn-inl.h:43-116 (k <= 6)."""
    return _knn_common(True, unknown, known, downnum, upnum, k, radius, out)


@torch.no_grad()
def KNN(unknown, known, downnum, upnum, *, k=3, out=None):
    """Reference: _contrib_KNN, k_nn-inl.h:40-113."""
    return _knn_common(False, unknown, known, downnum, upnum, k, 0.0, out)


class _BatchTake(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, index, neighbour_index=False):
        lib = _lib.load()
        B, N, C = data.shape
        M = index.numel() // B
        ctx.neighbour_index = neighbour_index
        out = torch.empty(tuple(index.shape) + (C,), dtype=torch.float32, device=data.device)
        with torch.cuda.device(data.device):
            rc = lib.gridgcn_batch_take(_ptr(data), _ptr(index), B, N, C, M, _ptr(out),
                                        _stream(data))
        _lib.check(rc, "gridgcn_batch_take")
        ctx.save_for_backward(index)
        ctx.dims = (B, N, C, M)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (index,) = ctx.saved_tensors
        return batch_take_g_backward(grad_out, index, ctx.dims[1], ctx.neighbour_index), None, None


@torch.no_grad()
def batch_take_g_backward(grad_out, index, N, neighbour_index=False):
    """Scatter-add backward of batch_take_g: grad_out [B,...,C] f32, index [B,...] i32 -> gradient
    w.r.t. data [B,N,C] (what MXNet's take backward does, utils/ops.py:87-92).
    neighbour_index: every index lies in [-1, N-1] (what Gridify / GridifyUp / BallKNN produce):
    the gradient is then formed as a sorted segmented sum, 3-6x faster at the layer sizes."""
    lib = _lib.load()
    B, C = index.shape[0], grad_out.shape[-1]
    M = index.numel() // B
    grad_out = grad_out.contiguous()
    gdata = torch.zeros((B, N, C), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        if SORTED_TAKE_BWD and neighbour_index:
            # sorted segmented sum (csrc/gridgcn_scatter.hip); scatter-add inside for odd widths
            nb = ctypes.c_size_t(0)
            lib.gridgcn_take_backward_workspace_bytes(B, N, M, ctypes.byref(nb))
            ws = torch.empty(nb.value, dtype=torch.uint8, device=grad_out.device)
            rc = lib.gridgcn_batch_take_backward_sorted(_ptr(grad_out), _ptr(index), B, N, C, M,
                                                        _ptr(gdata), _ptr(ws), nb.value,
                                                        _stream(grad_out))
        else:
            rc = lib.gridgcn_batch_take_backward(_ptr(grad_out), _ptr(index), B, N, C, M,
                                                 _ptr(gdata), _stream(grad_out))
    _lib.check(rc, "gridgcn_batch_take_backward")
    return gdata


def batch_take_g(data, index, shape=None, scope="", neighbour_index=False):
    """Per-cloud gather: data [B,N,C] f32, index [B,...] i32 -> [B,...,C]
    (utils/ops.py:78-93; flat take with mode='clip').  `shape`/`scope` are accepted for
    signature compatibility and ignored (they only size the MXNet symbol).
    neighbour_index=True promises indices in [-1, N-1] (see batch_take_g_backward)."""
    _chk(data, "data", 3, torch.float32)
    _require(isinstance(index, torch.Tensor) and index.is_cuda and index.dtype == torch.int32
             and index.is_contiguous() and index.shape[0] == data.shape[0],
             "index must be a contiguous int32 GPU tensor [B,...]")
    return _BatchTake.apply(data, index, bool(neighbour_index))


# ---- fused GridConv edge pipeline (inference-mode BatchNorm) -----------------------------------
def pack_conv_layer(w, b, rows=None, K=None):
    """(W^T [cin,cout], bias [cout]) with BatchNorm folded -> device tensors in the layout
    gridgcn_gridconv_forward reads (include/gridgcn.h):

      * K    = contraction length, a multiple of 4.  `rows[i]` = LDS column fed by input channel
               i (default: identity); unused rows are zero.
      * ldw  = cout rounded up to 32/64/128/256, zero padded.
      * W is packed per group of up to 128 columns as [group][K][32 lanes][NT] (NT = columns of the
        group / 32) so that a lane fetches the NT weights of one k with a single vector load.
    """
    cin, cout = w.shape
    if rows is None:
        rows = list(range(cin))
    if K is None:
        K = (max(rows) + 1 + 3) & ~3
    ldw = next((x for x in (32, 64, 128, 256) if x >= cout), None)
    _require(ldw is not None, "GridConv layer wider than 256 channels is not supported")
    W = torch.zeros((K, ldw), dtype=torch.float32, device=w.device)
    W[torch.as_tensor(rows, device=w.device), :cout] = w
    gw = min(ldw, 128)
    nt = gw // 32
    # [K, ldw] -> [groups, K, 32, nt]:  packed[g, k, j, t] = W[k, g*gw + t*32 + j]
    Wp = W.reshape(K, ldw // gw, nt, 32).permute(1, 0, 3, 2).contiguous()
    B = torch.zeros((ldw,), dtype=torch.float32, device=w.device)
    B[:cout] = b
    return (Wp, B.contiguous(), K, ldw, cout)


@torch.no_grad()
def gridconv_forward(src, nebidx, cent, pt_layers, att_layers, *, has_feats, localfdim):
    """Fused gather -> geo features -> pt-MLP (x) att-MLP -> max over P.

    src [B,Nsrc,4+C] f32, nebidx [B,O,P] i32, cent [B,O,>=3] f32 (xyz first),
    pt_layers / att_layers: lists of pack_conv_layer() tuples (att: exactly two).
    Returns [B,O,C_out] f32 = max_p relu(att2) * relu(pt_last)
    (segmentation/models/gcn_module_g_att.py:135-167, 57-59)."""
    lib = _lib.load()
    _chk(src, "src", 3, torch.float32)
    _chk(nebidx, "nebidx", 3, torch.int32)
    _chk(cent, "cent", 3, torch.float32)
    B, Nsrc, Cs = src.shape
    _, O, P = nebidx.shape
    _require(cent.shape[0] == B and cent.shape[1] == O and cent.shape[2] >= 3, "cent must be [B,O,>=3]")
    _require(len(att_layers) == 2 and 1 <= len(pt_layers) <= 4, "need 1..4 pt layers, 2 att layers")

    def arr(layers):
        a = (_lib.ConvLayer * len(layers))()
        for j, (W, b, K, ldw, cout) in enumerate(layers):
            a[j].W, a[j].b, a[j].K, a[j].ldw, a[j].cout = W.data_ptr(), b.data_ptr(), K, ldw, cout
        return a
    pt, att = arr(pt_layers), arr(att_layers)
    out = torch.empty((B, O, pt_layers[-1][4]), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = lib.gridgcn_gridconv_forward(_ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B,
                                          Nsrc, Cs, O, P, int(bool(has_feats)), int(localfdim),
                                          len(pt_layers), pt, att, _ptr(out), _stream(src))
    _lib.check(rc, "gridgcn_gridconv_forward")
    return out


class _EdgeInputs(torch.autograd.Function):
    """nf, att_vec of sub_g_update straight from (src, nebidx, cent) -- gridgcn_edge_inputs."""

    @staticmethod
    def forward(ctx, src, nebidx, cent, has_feats, localfdim):
        lib = _lib.load()
        B, Nsrc, Cs = src.shape
        _, O, P = nebidx.shape
        geo = (not has_feats) or localfdim != 0
        cin = (3 if geo else 0) + (Cs - 4 if has_feats else 0)
        nf = torch.empty((B, O, P, cin), dtype=torch.float32, device=src.device)
        att = torch.empty((B, O, P, 10), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            rc = lib.gridgcn_edge_inputs(_ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B, Nsrc,
                                         Cs, O, P, int(has_feats), int(localfdim), _ptr(nf),
                                         _ptr(att), _stream(src))
        _lib.check(rc, "gridgcn_edge_inputs")
        ctx.save_for_backward(nebidx)
        ctx.meta = (B, Nsrc, Cs, O, P, has_feats, localfdim)
        ctx.mark_non_differentiable(att)
        return nf, att

    @staticmethod
    def backward(ctx, gnf, gatt):
        B, Nsrc, Cs, O, P, has_feats, localfdim = ctx.meta
        if not has_feats or not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        lib = _lib.load()
        (nebidx,) = ctx.saved_tensors
        gnf = gnf.contiguous()
        gsrc = torch.zeros((B, Nsrc, Cs), dtype=torch.float32, device=gnf.device)
        with torch.cuda.device(gnf.device):
            rc = lib.gridgcn_edge_inputs_backward(_ptr(gnf), _ptr(nebidx), B, Nsrc, Cs, O, P,
                                                  int(has_feats), int(localfdim), _ptr(gsrc),
                                                  _stream(gnf))
        _lib.check(rc, "gridgcn_edge_inputs_backward")
        return gsrc, None, None, None, None


class _EdgeInputsRows(torch.autograd.Function):
    """gridgcn_edge_inputs_rows: nf = features | geo_vec | zero padding (row length a multiple of
    8), att16 = att_vec | 6 zeros -- the row layout the MFMA training kernels read directly."""

    @staticmethod
    def forward(ctx, src, nebidx, cent, has_feats, localfdim):
        lib = _lib.load()
        B, Nsrc, Cs = src.shape
        _, O, P = nebidx.shape
        geo = (not has_feats) or localfdim != 0
        nfeat = Cs - 4 if has_feats else 0
        nfs = (nfeat + (3 if geo else 0) + 7) & ~7
        nf = torch.empty((B, O, P, nfs), dtype=torch.float32, device=src.device)
        att = torch.empty((B, O, P, 16), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            rc = lib.gridgcn_edge_inputs_rows(_ptr(src), _ptr(nebidx), _ptr(cent), cent.shape[2], B,
                                              Nsrc, Cs, O, P, int(has_feats), int(localfdim), nfs,
                                              _ptr(nf), _ptr(att), _stream(src))
        _lib.check(rc, "gridgcn_edge_inputs_rows")
        ctx.save_for_backward(nebidx)
        ctx.meta = (B, Nsrc, Cs, O, P, has_feats, nfs)
        ctx.mark_non_differentiable(att)
        return nf, att

    @staticmethod
    def backward(ctx, gnf, gatt):
        B, Nsrc, Cs, O, P, has_feats, nfs = ctx.meta
        if not has_feats or not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        lib = _lib.load()
        (nebidx,) = ctx.saved_tensors
        gnf = gnf.contiguous()
        gsrc = torch.zeros((B, Nsrc, Cs), dtype=torch.float32, device=gnf.device)
        with torch.cuda.device(gnf.device):
            ws, nbytes = None, ctypes.c_size_t(0)
            if SORTED_TAKE_BWD:
                lib.gridgcn_take_backward_workspace_bytes(B, Nsrc, O * P, ctypes.byref(nbytes))
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=gnf.device)
            rc = lib.gridgcn_edge_inputs_rows_backward(_ptr(gnf), nfs, _ptr(nebidx), B, Nsrc, Cs, O,
                                                       P, _ptr(gsrc),
                                                       _ptr(ws) if ws is not None else None,
                                                       nbytes.value, _stream(gnf))
        _lib.check(rc, "gridgcn_edge_inputs_rows_backward")
        return gsrc, None, None, None, None


def edge_inputs_rows_supported(src, has_feats):
    return (not has_feats) or (src.shape[2] - 4) % 4 == 0


def edge_inputs_rows(src, nebidx, cent, *, has_feats, localfdim):
    """(nf [B,O,P,nfs], att16 [B,O,P,16], rot): the edge inputs of sub_g_update in the padded
    "features first" row layout; rot = number of leading reference channels (geo_vec) that were
    moved behind the features (the first conv's weight columns must be rotated by it)."""
    _chk(src, "src", 3, torch.float32)
    _chk(nebidx, "nebidx", 3, torch.int32)
    _chk(cent, "cent", 3, torch.float32)
    nf, att = _EdgeInputsRows.apply(src, nebidx, cent, bool(has_feats), int(localfdim))
    rot = 3 if (has_feats and localfdim != 0) else 0
    return nf, att, rot


def edge_inputs(src, nebidx, cent, *, has_feats, localfdim):
    """(nf [B,O,P,cin], att_vec [B,O,P,10]) of sub_g_update (gcn_module_g_att.py:190-250) in one
    kernel; differentiable w.r.t. the feature columns of src."""
    _chk(src, "src", 3, torch.float32)
    _chk(nebidx, "nebidx", 3, torch.int32)
    _chk(cent, "cent", 3, torch.float32)
    return _EdgeInputs.apply(src, nebidx, cent, bool(has_feats), int(localfdim))
