"""The C-ABI library builds for gfx950, loads, and exports every symbol include/gridgcn.h declares.
No compute call is made here (no GPU in CI)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from grid_gcn_amd import build, _lib
    build.build()
    return _lib.load()


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gridgcn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gridgcn_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "libgridgcn_hip.so does not export %s" % s
    from grid_gcn_amd import _lib
    assert sorted(_lib.EXPORTS) == syms


def test_abi_version_and_strerror(lib):
    from grid_gcn_amd import _lib
    assert lib.gridgcn_abi_version() == _lib.ABI_VERSION == 9
    for opt in (_lib.OPT_INDEX_SMALL, _lib.OPT_COL_SPLIT):          # round 4: both on by default, 0 / 1 only
        assert lib.gridgcn_get_option(opt) == 1
        assert lib.gridgcn_set_option(opt, 2) == 1 and lib.gridgcn_set_option(opt, 0) == 0
        assert lib.gridgcn_get_option(opt) == 0 and lib.gridgcn_set_option(opt, 1) == 0
    # kernel-selection options: explicit API, no environment variables (tests/test_guards.py)
    assert lib.gridgcn_get_option(_lib.OPT_ATT_BWD_FUSED) == 1
    assert lib.gridgcn_set_option(_lib.OPT_ATT_BWD_FUSED, 0) == 0
    assert lib.gridgcn_get_option(_lib.OPT_ATT_BWD_FUSED) == 0
    assert lib.gridgcn_set_option(_lib.OPT_ATT_BWD_FUSED, 1) == 0
    assert lib.gridgcn_set_option(12345, 1) == 1 and lib.gridgcn_get_option(12345) == -1
    assert lib.gridgcn_strerror(0) == b"ok"
    assert b"workspace" in lib.gridgcn_strerror(2)


def test_argument_validation_without_gpu(lib):
    """workspace-size queries are pure host code: check the attribute domain."""
    from grid_gcn_amd._lib import GridParams
    p = GridParams()
    p.max_p_grid, p.max_o_grid, p.kernel_size, p.stride, p.loc = 64, 1024, 3, 1, 1
    for j in range(3):
        p.coord_shift[j], p.voxel_size[j], p.grid_size[j] = 1.0, 0.05, 40
    n = ctypes.c_size_t(0)
    assert lib.gridgcn_gridify_workspace_bytes(16, 8192, ctypes.byref(p), ctypes.byref(n)) == 0
    assert n.value > 16 * 8192 * 4
    # scratch stays far below the reference's dense B*G*P table (262 MB at this config, F8)
    assert n.value < 40 * 1024 * 1024
    p.max_p_grid = 129
    assert lib.gridgcn_gridify_workspace_bytes(16, 8192, ctypes.byref(p), ctypes.byref(n)) == 1
    p.max_p_grid, p.kernel_size = 64, 4
    assert lib.gridgcn_gridify_workspace_bytes(16, 8192, ctypes.byref(p), ctypes.byref(n)) == 1
    p.kernel_size = 3
    p.grid_size[0] = 1 << 20
    assert lib.gridgcn_gridify_workspace_bytes(16, 8192, ctypes.byref(p), ctypes.byref(n)) == 1


def test_round3_entries_reject_bad_arguments_without_gpu(lib):
    """argument checks that run before any launch: the round-3 entry points return GRIDGCN_EINVAL (1)."""
    null = None
    # gridgcn_linear_bwd_fin: the sums are mandatory
    rc = lib.gridgcn_linear_bwd_fin(*([null] * 19), 0, 1024, 32, 32, 32, 0, 0, 0, 0, 0, null, null, null, null,
                                    null, 0, null, 0, null)
    assert rc == 1
    # the batched weight pack: no table / no layers
    assert lib.gridgcn_pack_linear_batch(null, 0, 0, null) == 1
    # options outside their ranges
    from grid_gcn_amd import _lib
    assert lib.gridgcn_set_option(_lib.OPT_INDEX_SLAB_SHIFT, 99) == 1
    assert lib.gridgcn_set_option(12345, 0) == 1


def test_round4_dropout_entries_reject_bad_arguments_without_gpu(lib):
    """the fused-Dropout entries check their shape domain before any launch (K % 32, <= 32 outputs, 0 < p < 1)"""
    f32 = (ctypes.c_float * 64)()
    a = ctypes.cast(f32, ctypes.c_void_p)
    # null pointers
    assert lib.gridgcn_linear_fwd_direct_drop(None, 64, 128, 128, None, None, 32, 24, None, None, None, 0.5, 1, None,
                                              None) == 1
    # K not a multiple of 32, too many outputs, p outside (0, 1)
    assert lib.gridgcn_linear_fwd_direct_drop(a, 64, 100, 100, a, a, 32, 24, a, a, a, 0.5, 1, None, None) == 1
    assert lib.gridgcn_linear_fwd_direct_drop(a, 64, 128, 128, a, a, 64, 40, a, a, a, 0.5, 1, None, None) == 1
    assert lib.gridgcn_linear_fwd_direct_drop(a, 64, 128, 128, a, a, 32, 24, a, a, a, 0.0, 1, None, None) == 1
    assert lib.gridgcn_linear_dw_drop(*([None] * 11), 64, 24, 128, 0.5, 1, None, None, None, 0, None) == 1
    assert lib.gridgcn_linear_dw_drop(*([a] * 11), 64, 24, 128, 1.0, 1, None, a, a, 1 << 20, None) == 1
    # a workspace that is too small is its own code (GRIDGCN_EWORKSPACE)
    assert lib.gridgcn_linear_dw_drop(*([a] * 11), 1 << 16, 24, 128, 0.5, 1, None, a, a, 16, None) == 2


def test_ops_fail_loudly_without_gpu():
    """No CPU fallback: CPU tensors are rejected, never silently routed elsewhere."""
    import torch
    from grid_gcn_amd import ops
    data = torch.zeros(1, 16, 4)
    npn = torch.full((1, 1), 16, dtype=torch.int32)
    with pytest.raises(RuntimeError):
        ops.Gridify(data, npn, max_p_grid=4, max_o_grid=4, kernel_size=3, coord_shift=[1, 1, 1],
                    voxel_size=[0.5] * 3, grid_size=[4] * 3)
    with pytest.raises(RuntimeError):
        ops.BallKNN(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3), npn, npn, k=3, radius=1.0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "grid_gcn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)


def test_torch_ops_are_registered_with_fake_kernels():
    """`import grid_gcn_amd` makes torch.ops.gridgcn.* appear (the analogue of the reference's
    static registration on loading additional.so) and the ops trace on fake tensors (no GPU)."""
    import torch
    import grid_gcn_amd  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in ("gridify", "gridify_knn", "gridify_up", "ball_knn", "knn", "batch_take",
                 "batch_take_backward"):
        assert hasattr(torch.ops.gridgcn, name), name
    schema = str(torch.ops.gridgcn.gridify.default._schema)
    for arg in ("max_p_grid", "max_o_grid", "kernel_size", "stride", "loc", "coord_shift",
                "voxel_size", "grid_size", "seed"):
        assert arg in schema, (arg, schema)
    with FakeTensorMode():
        data = torch.empty((2, 100, 4), device="cuda")
        num = torch.empty((2, 1), dtype=torch.int32, device="cuda")
        out = torch.ops.gridgcn.gridify(data, num, max_p_grid=8, max_o_grid=16, kernel_size=3,
                                        stride=1, loc=1, coord_shift=[1.0] * 3,
                                        voxel_size=[0.1] * 3, grid_size=[20] * 3)
        assert [tuple(t.shape) for t in out] == [(2, 16, 8), (2, 16, 8), (2, 16, 4), (2, 16), (2, 1)]
        assert out[0].dtype == torch.int32 and out[4].dtype == torch.int32
        idx = torch.ops.gridgcn.ball_knn(data[..., :3], data[..., :3], num, num, k=5, radius=0.2)
        assert tuple(idx.shape) == (2, 100, 5) and idx.dtype == torch.int32
        g = torch.ops.gridgcn.batch_take(data, out[0])
        assert tuple(g.shape) == (2, 16, 8, 4)
