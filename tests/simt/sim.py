"""numpy front end of the host-side emulation of the library (tests/simt/): the operator signatures of
grid_gcn_amd.ops on numpy arrays, each a call of the SAME C-ABI entry the product binds (include/gridgcn.h:
gridgcn_gridify, gridgcn_gridify_up, gridgcn_ball_knn, ... -- gridgcn_capi.hip compiled for the host with the rest),
plus a few of the training entries.  TEST INFRASTRUCTURE -- the product never imports this."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from grid_gcn_amd import _lib  # noqa: E402
from grid_gcn_amd._lib import GridParams  # noqa: E402  (the ctypes mirror of gridgcn_grid_params)
import emu  # noqa: E402

load = emu.library
counters = emu.counters


def set_order(order):
    """schedule of the launches that follow (tests/simt/simt_hip.h: simt_order, a bit mask): 0 workgroups / waves / lanes
    ascending, 1 / 2 / 4 workgroups / waves / lanes descending, 8 workgroups in a pseudo-random permutation"""
    load().simt_set_order(int(order))


def set_option(which, value):
    """gridgcn_set_option (e.g. _lib.OPT_INDEX_SMALL)"""
    assert load().gridgcn_set_option(int(which), int(value)) == 0


def _params(max_p_grid, max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size, grid_size, seed):
    p = GridParams()
    p.max_p_grid, p.max_o_grid, p.kernel_size = int(max_p_grid), int(max_o_grid), int(kernel_size)
    p.stride, p.loc = int(stride), int(loc)
    for j in range(3):
        p.coord_shift[j] = float(coord_shift[j])
        p.voxel_size[j] = float(voxel_size[j])
        p.grid_size[j] = int(grid_size[j])
    p.seed = int(seed) & (2 ** 64 - 1)
    p.seed_dev = None
    return p


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _ws(nbytes):
    """a caller's workspace: garbage on entry, 64-byte aligned"""
    raw = np.full(int(nbytes) + 128, 0xA5, np.uint8)
    off = (-raw.ctypes.data) % 64
    return raw[off:off + int(nbytes) + 64]


def _gridify(fn, data, npn, *, max_p_grid, max_o_grid, kernel_size, stride=1, loc=0, coord_shift, voxel_size,
             grid_size, seed=0, extra=()):
    lib = load()
    data = np.ascontiguousarray(data, np.float32)
    npn = np.ascontiguousarray(npn, np.int32)
    B, N, _ = data.shape
    p = _params(max_p_grid, max_o_grid, kernel_size, stride, loc, coord_shift, voxel_size, grid_size, seed)
    nb = ctypes.c_size_t(0)
    rc = getattr(lib, fn + "_workspace_bytes")(B, N, ctypes.byref(p), ctypes.byref(nb))
    assert rc == 0, rc
    ws = _ws(nb.value)
    O, P = int(max_o_grid), int(max_p_grid)
    nebidx = np.full((B, O, P), -7, np.int32)
    nebmsk = np.full((B, O, P), np.nan, np.float32)
    cent = np.full((B, O, 4), np.nan, np.float32)
    centmsk = np.full((B, O), np.nan, np.float32)
    centnum = np.full((B, 1), -7, np.int32)
    rc = getattr(lib, fn)(_ptr(data), _ptr(npn), B, N, ctypes.byref(p), *extra, _ptr(nebidx), _ptr(nebmsk), _ptr(cent),
                          _ptr(centmsk), _ptr(centnum), _ptr(ws), nb.value, None)
    assert rc == 0, (fn, rc)
    return nebidx, nebmsk, cent, centmsk, centnum


def Gridify(data, npn, **kw):
    return _gridify("gridgcn_gridify", data, npn, **kw)


def GridifyKNN(data, npn, **kw):
    return _gridify("gridgcn_gridify_knn", data, npn, **kw)


def Gridify_occaware(data, npn, beta=1.0, **kw):
    return _gridify("gridgcn_gridify_occaware", data, npn, extra=(ctypes.c_float(float(beta)),), **kw)


def Gridify_fast_rand(data, npn, **kw):
    return _gridify("gridgcn_gridify_fast_rand", data, npn, **kw)


def GridifyUp(down, up, down_np, up_np, *, max_p_grid, max_o_grid, kernel_size, coord_shift, voxel_size, grid_size,
              seed=0):
    lib = load()
    down = np.ascontiguousarray(down, np.float32)
    up = np.ascontiguousarray(up, np.float32)
    down_np = np.ascontiguousarray(down_np, np.int32)
    up_np = np.ascontiguousarray(up_np, np.int32)
    B, Nd, _ = down.shape
    assert up.shape == (B, max_o_grid, 4)
    p = _params(max_p_grid, max_o_grid, kernel_size, 1, 0, coord_shift, voxel_size, grid_size, seed)
    nb = ctypes.c_size_t(0)
    assert lib.gridgcn_gridify_up_workspace_bytes(B, Nd, ctypes.byref(p), ctypes.byref(nb)) == 0
    ws = _ws(nb.value)
    nebidx = np.full((B, max_o_grid, max_p_grid), -7, np.int32)
    nebmsk = np.full((B, max_o_grid, max_p_grid), np.nan, np.float32)
    rc = lib.gridgcn_gridify_up(_ptr(down), _ptr(up), _ptr(down_np), _ptr(up_np), B, Nd, ctypes.byref(p), _ptr(nebidx),
                                _ptr(nebmsk), _ptr(ws), nb.value, None)
    assert rc == 0, rc
    return nebidx, nebmsk


def BallKNN(unknown, known, downnum, upnum, k=3, radius=0.1, grid=False):
    lib = load()
    unknown = np.ascontiguousarray(unknown, np.float32)
    known = np.ascontiguousarray(known, np.float32)
    downnum = np.ascontiguousarray(downnum, np.int32)
    upnum = np.ascontiguousarray(upnum, np.int32)
    B, n, _ = unknown.shape
    m = known.shape[1]
    idx = np.zeros((B, n, k), np.int32)                   # (rows >= upnum stay untouched: zeros here, as ops.BallKNN)
    if grid:
        nb = ctypes.c_size_t(0)
        assert lib.gridgcn_ball_knn_grid_workspace_bytes(B, m, ctypes.byref(nb)) == 0
        ws = _ws(nb.value)
        rc = lib.gridgcn_ball_knn_grid(_ptr(unknown), _ptr(known), _ptr(downnum), _ptr(upnum), B, n, m, int(k),
                                       float(radius), _ptr(idx), _ptr(ws), nb.value, None)
    else:
        rc = lib.gridgcn_ball_knn(_ptr(unknown), _ptr(known), _ptr(downnum), _ptr(upnum), B, n, m, int(k),
                                  float(radius), _ptr(idx), None)
    assert rc == 0, rc
    return idx


def KNN(unknown, known, downnum, upnum, k=3):
    lib = load()
    unknown = np.ascontiguousarray(unknown, np.float32)
    known = np.ascontiguousarray(known, np.float32)
    downnum = np.ascontiguousarray(downnum, np.int32)
    upnum = np.ascontiguousarray(upnum, np.int32)
    B, n, _ = unknown.shape
    m = known.shape[1]
    idx = np.zeros((B, n, k), np.int32)
    rc = lib.gridgcn_knn(_ptr(unknown), _ptr(known), _ptr(downnum), _ptr(upnum), B, n, m, int(k), _ptr(idx), None)
    assert rc == 0, rc
    return idx


# ---- training kernels ---------------------------------------------------------------------------------------------
def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def att_bn2_moments(Z1, s1, h1, W2, b2, gamma, beta, eps=1e-3):
    """gridgcn_att_bn2_moments: (scale, shift, mean, rstd [128], sums [2][128] fp64, moments [17 * 64] fp64)"""
    lib = load()
    Z1, s1, h1, W2, b2, gamma, beta = map(_f32, (Z1, s1, h1, W2, b2, gamma, beta))
    E = Z1.shape[0]
    nb = ctypes.c_size_t(0)
    assert lib.gridgcn_att_fwd_noz_workspace_bytes(E, 32, 128, ctypes.byref(nb)) == 0
    ws = _ws(nb.value)
    vec = np.full((4, 128), np.nan, np.float32)
    sums = np.full((2, 128), np.nan, np.float64)
    rc = lib.gridgcn_att_bn2_moments(_ptr(Z1), _ptr(s1), _ptr(h1), _ptr(W2), _ptr(b2), _ptr(gamma), _ptr(beta), E, 32,
                                     128, float(eps), 0.0, _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]), _ptr(vec[3]), None,
                                     None, None, _ptr(sums), _ptr(ws), nb.value, None)
    assert rc == 0, rc
    off = ctypes.c_size_t(0)
    assert lib.gridgcn_att_moments_offset(E, 32, 128, ctypes.byref(off)) == 0
    assert off.value + 17 * 64 * 8 == nb.value
    mom = ws[off.value:off.value + 17 * 64 * 8].view(np.float64).copy()
    return vec, sums, mom


def att_bwd_noz(Z1, ps, psh, pm, pr, W2, b2, sc, mu, rs, bsums, amax, gval, P, mom=None, v2=1):
    """gridgcn_att_bwd_noz (mom None) / gridgcn_att_bwd_noz_mom: dict(dX, dW, v [4][128], psums [2][32], s1 [32])"""
    lib = load()
    Z1, ps, psh, pm, pr, W2, b2, sc, mu, rs, gval = map(_f32, (Z1, ps, psh, pm, pr, W2, b2, sc, mu, rs, gval))
    bsums = np.ascontiguousarray(bsums, np.float64)
    amax = np.ascontiguousarray(amax, np.uint8)
    E = Z1.shape[0]
    nb = ctypes.c_size_t(0)
    assert lib.gridgcn_att_bwd_noz_workspace_bytes(E, 32, 128, ctypes.byref(nb)) == 0
    ws = _ws(nb.value)
    dX = np.full((E + 8, 32), 7.0, np.float32)          # (guard rows: nothing may be written past E)
    dW = np.full((128, 32), np.nan, np.float32)
    v = np.full((4, 128), np.nan, np.float32)
    psums = np.zeros((2, 32), np.float64)
    s1 = np.zeros(32, np.float64)
    set_option(_lib.OPT_ATT_NZ_V2, v2)
    try:
        if mom is not None:
            mom = np.ascontiguousarray(mom, np.float64)
            rc = lib.gridgcn_att_bwd_noz_mom(_ptr(Z1), _ptr(ps), _ptr(psh), _ptr(pm), _ptr(pr), _ptr(W2), _ptr(b2),
                                             _ptr(sc), _ptr(mu), _ptr(rs), _ptr(bsums), _ptr(amax), _ptr(gval), int(P),
                                             E, 32, 128, _ptr(mom), _ptr(dX), _ptr(dW), _ptr(v[0]), _ptr(v[1]),
                                             _ptr(v[2]), _ptr(v[3]), _ptr(psums), _ptr(ws), nb.value, None)
        else:
            rc = lib.gridgcn_att_bwd_noz(_ptr(Z1), _ptr(ps), _ptr(psh), _ptr(pm), _ptr(pr), _ptr(W2), _ptr(b2),
                                         _ptr(sc), _ptr(mu), _ptr(rs), _ptr(bsums), _ptr(amax), _ptr(gval), int(P), E,
                                         32, 128, _ptr(dX), _ptr(dW), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(v[3]),
                                         _ptr(psums), _ptr(s1), _ptr(ws), nb.value, None)
    finally:
        set_option(_lib.OPT_ATT_NZ_V2, 1)
    assert rc == 0, rc
    assert np.all(dX[E:] == 7.0), "rows past E were written"
    return dict(dX=dX[:E], dW=dW, v=v, psums=psums, s1=s1)
