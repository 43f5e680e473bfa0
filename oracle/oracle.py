"""numpy/ctypes binding of the CPU oracle (oracle/gridgcn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package grid_gcn_amd.

Every function mirrors the reference operator of the same name (argument order and
meaning follow gridifyop/gridify-inl.h:58-87,146-152, gridify_up-inl.h:58-81,137-143,
ball_k_nn-inl.h:31-40, k_nn-inl.h) and returns numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgridgcn_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "gridgcn_oracle.c")
    if force or not os.path.exists(_SO) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.gridgcn_oracle_xorwow_uniform.restype = ctypes.c_float
        _lib.gridgcn_oracle_xorwow_uniform.argtypes = [ctypes.c_uint64]
    return _lib


def xorwow_raw(seed, xs0, xs1, m0, m1, n=1):
    """n-th raw 32-bit output of the oracle's XORWOW skeleton under the four seed-scramble constants given."""
    lib = _load()
    lib.gridgcn_oracle_xorwow_raw.restype = ctypes.c_uint32
    lib.gridgcn_oracle_xorwow_raw.argtypes = [ctypes.c_uint64] + [ctypes.c_uint32] * 4 + [ctypes.c_int]
    return int(lib.gridgcn_oracle_xorwow_raw(ctypes.c_uint64(seed & (2**64 - 1)), xs0, xs1, m0, m1, n))


def rocrand_xorwow_raw(seed, n=1):
    """n-th raw output of rocRAND's own XORWOW engine (oracle/xorwow_rocrand.cpp); None when the image has no rocRAND
    headers."""
    so = os.path.join(_HERE, "_build", "libxorwow_rocrand.so")
    if not os.path.exists(so):
        subprocess.call(["make", "-C", _HERE, "-s"])
    if not os.path.exists(so):
        return None
    lib = ctypes.CDLL(so)
    lib.gridgcn_rocrand_xorwow_raw.restype = ctypes.c_uint32
    lib.gridgcn_rocrand_xorwow_raw.argtypes = [ctypes.c_uint64, ctypes.c_int]
    return int(lib.gridgcn_rocrand_xorwow_raw(ctypes.c_uint64(seed & (2**64 - 1)), n))


def threads_for(work):
    """OpenMP team size the oracle uses for a loop of `work` independent clouds / queries."""
    lib = _load()
    lib.gridgcn_oracle_threads.restype = ctypes.c_int
    lib.gridgcn_oracle_threads.argtypes = [ctypes.c_longlong]
    return int(lib.gridgcn_oracle_threads(int(work)))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f3(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(3))
    return a


def _i3(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32).reshape(3))


def xorwow_uniform(seed):
    return float(_load().gridgcn_oracle_xorwow_uniform(ctypes.c_uint64(seed & (2**64 - 1))))


def _gridify_like(fn, data, actual_numpoints, max_p_grid, max_o_grid, kernel_size, stride, loc,
                  coord_shift, voxel_size, grid_size, seed, extra=()):
    data = np.ascontiguousarray(data, dtype=np.float32)
    anp = np.ascontiguousarray(np.asarray(actual_numpoints, dtype=np.int32).reshape(-1))
    B, N, C = data.shape
    assert C == 4 and anp.shape[0] == B
    P, O = int(max_p_grid), int(max_o_grid)
    sh, vs, gs = _f3(coord_shift), _f3(voxel_size), _i3(grid_size)
    nebidx = np.empty((B, O, P), np.int32)
    nebmsk = np.empty((B, O, P), np.float32)
    cent = np.empty((B, O, 4), np.float32)
    centmsk = np.empty((B, O), np.float32)
    centnum = np.empty((B, 1), np.int32)
    rc = fn(_p(data), _p(anp), B, N, P, O, int(kernel_size), int(stride), int(loc),
            _p(sh), _p(vs), _p(gs), ctypes.c_uint64(int(seed)), *extra,
            _p(nebidx), _p(nebmsk), _p(cent), _p(centmsk), _p(centnum))
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return nebidx, nebmsk, cent, centmsk, centnum


def gridify(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1, loc=0,
            coord_shift, voxel_size, grid_size, seed=0):
    """S0 restatement of mx.sym.Gridify (gridify.cu:126-190, 218-290)."""
    return _gridify_like(_load().gridgcn_oracle_gridify, data, actual_numpoints, max_p_grid,
                         max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size, grid_size,
                         seed)


def gridify_occaware(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1,
                     loc=0, coord_shift, voxel_size, grid_size, seed=0, beta=1.0):
    """Gridify with Coverage-Aware Sampling.  PARITY UNPINNED: our sequential statement of the
    paper's section 3.2 (gridgcn_oracle.c: cas_refine_cloud); the reference has no source for it."""
    return _gridify_like(_load().gridgcn_oracle_gridify_occaware, data, actual_numpoints,
                         max_p_grid, max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size,
                         grid_size, seed, extra=(ctypes.c_float(float(beta)),))


def gridify_fast_rand(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1,
                      loc=0, coord_shift, voxel_size, grid_size, seed=0):
    """S0 restatement of the fast_rand build of Gridify (gridifyop/fast_rand/gridify.cu:126-272);
    its draws are seeded with the thread index only, `seed` is accepted and ignored."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    anp = np.ascontiguousarray(np.asarray(actual_numpoints, dtype=np.int32).reshape(-1))
    B, N, C = data.shape
    assert C == 4 and anp.shape[0] == B
    P, O = int(max_p_grid), int(max_o_grid)
    sh, vs, gs = _f3(coord_shift), _f3(voxel_size), _i3(grid_size)
    nebidx = np.empty((B, O, P), np.int32)
    nebmsk = np.empty((B, O, P), np.float32)
    cent = np.empty((B, O, 4), np.float32)
    centmsk = np.empty((B, O), np.float32)
    centnum = np.empty((B, 1), np.int32)
    rc = _load().gridgcn_oracle_gridify_fast_rand(
        _p(data), _p(anp), B, N, P, O, int(kernel_size), int(stride), int(loc), _p(sh), _p(vs),
        _p(gs), _p(nebidx), _p(nebmsk), _p(cent), _p(centmsk), _p(centnum))
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return nebidx, nebmsk, cent, centmsk, centnum


def gridify_knn(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1, loc=0,
                coord_shift, voxel_size, grid_size, seed=0):
    """S0 restatement of mx.sym.GridifyKNN (gridifyknn.cu:115-204, 231-332)."""
    return _gridify_like(_load().gridgcn_oracle_gridify_knn, data, actual_numpoints, max_p_grid,
                         max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size, grid_size,
                         seed)


def gridify_up(downdata, updata, down_actual_numpoints, up_actual_numpoints, *, max_p_grid,
               max_o_grid, kernel_size, coord_shift, voxel_size, grid_size, seed=0):
    """S0 restatement of mx.sym.GridifyUp (gridify_up.cu:121-169, 190-224)."""
    down = np.ascontiguousarray(downdata, dtype=np.float32)
    up = np.ascontiguousarray(updata, dtype=np.float32)
    dnp = np.ascontiguousarray(np.asarray(down_actual_numpoints, dtype=np.int32).reshape(-1))
    unp = np.ascontiguousarray(np.asarray(up_actual_numpoints, dtype=np.int32).reshape(-1))
    B, Nd, _ = down.shape
    P, O = int(max_p_grid), int(max_o_grid)
    assert up.shape == (B, O, 4), "updata must be [B, max_o_grid, 4] (gridify_up.cu:196)"
    sh, vs, gs = _f3(coord_shift), _f3(voxel_size), _i3(grid_size)
    nebidx = np.empty((B, O, P), np.int32)
    nebmsk = np.empty((B, O, P), np.float32)
    rc = _load().gridgcn_oracle_gridify_up(_p(down), _p(up), _p(dnp), _p(unp), B, Nd, P, O,
                                           int(kernel_size), _p(sh), _p(vs), _p(gs),
                                           ctypes.c_uint64(int(seed)), _p(nebidx), _p(nebmsk))
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return nebidx, nebmsk


def ball_knn(unknown, known, downnum, upnum, *, k=3, radius=0.1, out=None):
    """BallKNNKernel::Map (ball_k_nn-inl.h:45-93).  Rows >= upnum[b] keep `out`'s content
    (zeros when `out` is None)."""
    un = np.ascontiguousarray(unknown, dtype=np.float32)
    kn = np.ascontiguousarray(known, dtype=np.float32)
    dn = np.ascontiguousarray(np.asarray(downnum, dtype=np.int32).reshape(-1))
    upn = np.ascontiguousarray(np.asarray(upnum, dtype=np.int32).reshape(-1))
    B, n, _ = un.shape
    m = kn.shape[1]
    idx = np.zeros((B, n, k), np.int32) if out is None else out
    rc = _load().gridgcn_oracle_ball_knn(_p(un), _p(kn), _p(dn), _p(upn), B, n, m, int(k),
                                         ctypes.c_float(radius), _p(idx))
    if rc != 0:
        raise RuntimeError("oracle returned %d (k must be <= 6)" % rc)
    return idx


def knn(unknown, known, downnum, upnum, *, k=3, out=None):
    """KNNKernel::Map (k_nn-inl.h:42-91)."""
    un = np.ascontiguousarray(unknown, dtype=np.float32)
    kn = np.ascontiguousarray(known, dtype=np.float32)
    dn = np.ascontiguousarray(np.asarray(downnum, dtype=np.int32).reshape(-1))
    upn = np.ascontiguousarray(np.asarray(upnum, dtype=np.int32).reshape(-1))
    B, n, _ = un.shape
    m = kn.shape[1]
    idx = np.zeros((B, n, k), np.int32) if out is None else out
    rc = _load().gridgcn_oracle_knn(_p(un), _p(kn), _p(dn), _p(upn), B, n, m, int(k), _p(idx))
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return idx


def batch_take(data, index):
    """batch_take_g (utils/ops.py:78-93): data[B,N,C], index[B,...] -> [B,...,C], clip mode."""
    d = np.ascontiguousarray(data, dtype=np.float32)
    ix = np.ascontiguousarray(index, dtype=np.int32)
    B, N, C = d.shape
    M = int(np.prod(ix.shape[1:]))
    out = np.empty(ix.shape + (C,), np.float32)
    _load().gridgcn_oracle_batch_take(_p(d), _p(ix), B, N, C, M, _p(out))
    return out
