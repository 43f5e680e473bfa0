"""ctypes binding of libgridgcn_hip.so (the C ABI of include/gridgcn.h).

There is NO fallback: if the library is missing or fails to load, every operator raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GG_HIP_LIB: profiling variant of the same library (lib/libgridgcn_hip_prof.so, tools/prof_phases.py)
LIB_PATH = os.environ.get("GG_HIP_LIB") or os.path.join(_HERE, "lib", "libgridgcn_hip.so")

ABI_VERSION = 9                 # include/gridgcn.h: gridgcn_abi_version()
OPT_ATT_BWD_FUSED = 0           # GRIDGCN_OPT_ATT_BWD_FUSED
OPT_INDEX_SLAB_SHIFT = 1        # GRIDGCN_OPT_INDEX_SLAB_SHIFT
OPT_INDEX_CHUNK = 2             # GRIDGCN_OPT_INDEX_CHUNK
OPT_INDEX_SMALL = 3             # GRIDGCN_OPT_INDEX_SMALL
OPT_COL_SPLIT = 4               # GRIDGCN_OPT_COL_SPLIT
OPT_PAIRMAX_SPLIT = 5           # GRIDGCN_OPT_PAIRMAX_SPLIT
OPT_ATT_NZ_V2 = 6               # GRIDGCN_OPT_ATT_NZ_V2
OPT_BWD_FUSED128 = 7            # GRIDGCN_OPT_BWD_FUSED128
OPT_ATT_EVAL_TILE = 8           # GRIDGCN_OPT_ATT_EVAL_TILE

EXPORTS = [
    "gridgcn_strerror", "gridgcn_abi_version", "gridgcn_set_mlp_precision",
    "gridgcn_get_mlp_precision", "gridgcn_set_option", "gridgcn_get_option",
    "gridgcn_gridify_workspace_bytes", "gridgcn_gridify", "gridgcn_gridify_timed",
    "gridgcn_gridify_occaware_workspace_bytes", "gridgcn_gridify_occaware",
    "gridgcn_gridify_fast_rand_workspace_bytes", "gridgcn_gridify_fast_rand",
    "gridgcn_gridify_knn_workspace_bytes", "gridgcn_gridify_knn",
    "gridgcn_gridify_up_workspace_bytes", "gridgcn_gridify_up",
    "gridgcn_ball_knn", "gridgcn_knn",
    "gridgcn_ball_knn_grid_workspace_bytes", "gridgcn_ball_knn_grid",
    "gridgcn_batch_take", "gridgcn_batch_take_backward", "gridgcn_batch_take_backward_sorted",
    "gridgcn_gridconv_forward", "gridgcn_edge_inputs", "gridgcn_edge_inputs_backward",
    "gridgcn_edge_inputs_rows", "gridgcn_edge_inputs_rows_backward",
    "gridgcn_take_backward_workspace_bytes",
    "gridgcn_edge_lin0_forward", "gridgcn_edge_lin0_backward", "gridgcn_pairmax_fwd_src",
    "gridgcn_edge_lin0_backward_sparse_workspace_bytes", "gridgcn_edge_lin0_backward_sparse",
    "gridgcn_edge_lin0_dwg", "gridgcn_att_max_eval",
    "gridgcn_softmax_ce_fwd", "gridgcn_softmax_ce_bwd", "gridgcn_colsum",
    "gridgcn_linear_fwd", "gridgcn_linear_bwd_workspace_bytes", "gridgcn_linear_bwd",
    "gridgcn_linear_fwd_ld", "gridgcn_linear_fwd_direct_ld", "gridgcn_linear_bwd_ld",
    "gridgcn_pairmax_fwd_src_z", "gridgcn_pack_desc_fill", "gridgcn_pack_linear_batch",
    "gridgcn_linear_bwd_fin", "gridgcn_gemm_small", "gridgcn_gemm_small_workspace_bytes",
    "gridgcn_pairmax_fwd", "gridgcn_pairmax_bwd", "gridgcn_pairmax_bwd_masked",
    "gridgcn_att_bwd_noz_workspace_bytes", "gridgcn_att_bwd_noz", "gridgcn_gemm_bias",
    "gridgcn_att_fwd_noz_workspace_bytes", "gridgcn_att_bn2_moments", "gridgcn_att_pairmax_fwd",
    "gridgcn_att_pairmax_fwd_supported", "gridgcn_att_bwd_noz_mom_supported", "gridgcn_att_moments_offset",
    "gridgcn_att_bwd_noz_mom",
    "gridgcn_bn_relu_apply", "gridgcn_bn_relu_bwd_reduce",
    "gridgcn_bn_relu_dropout_apply", "gridgcn_linear_dx",
    "gridgcn_bn_relu_bwd_elemt",
    "gridgcn_pack_linear", "gridgcn_linear_fwd_direct", "gridgcn_bn_finalize", "gridgcn_bn_bwd_finalize",
    "gridgcn_linear_fwd_direct2", "gridgcn_ctx_max", "gridgcn_ctx_max_backward",
    "gridgcn_bn_dz_segsum", "gridgcn_sparse_add", "gridgcn_bn_stats",
    "gridgcn_ball_knn_grid_ld", "gridgcn_ball_knn_ld", "gridgcn_bn_finalize_tail", "gridgcn_softmax_ce_loss", "gridgcn_colsum_f32",
    "gridgcn_cat_mask", "gridgcn_mask_sum", "gridgcn_adam_step",
    "gridgcn_edge_geo_forward_workspace_bytes", "gridgcn_edge_geo_forward",
    "gridgcn_edge_lin0_backward_sparse_geo", "gridgcn_linear_fwd_direct_fin",
    "gridgcn_linear_fwd_direct_drop", "gridgcn_linear_dw_drop",
]


class GridParams(ctypes.Structure):
    """struct gridgcn_grid_params (include/gridgcn.h) == GridifyParam (gridify-inl.h:58-87)."""
    _fields_ = [("max_p_grid", ctypes.c_int32), ("max_o_grid", ctypes.c_int32),
                ("kernel_size", ctypes.c_int32), ("stride", ctypes.c_int32),
                ("loc", ctypes.c_int32), ("coord_shift", ctypes.c_float * 3),
                ("voxel_size", ctypes.c_float * 3), ("grid_size", ctypes.c_int32 * 3),
                ("seed", ctypes.c_uint64), ("seed_dev", ctypes.c_void_p)]


class PackDesc(ctypes.Structure):
    """struct gridgcn_pack_desc (include/gridgcn.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("W", "b", "Wp", "Bp", "Wb", "Wg", "Wq", "Wdx", "wgb")] + \
               [(n, ctypes.c_int32) for n in ("C", "cin_w", "rot", "cin", "ndx", "K", "ldw", "n", "geo",
                                              "reserved")]


class BnFin(ctypes.Structure):
    """struct gridgcn_bn_fin (include/gridgcn.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("gamma", "beta", "scale", "shift", "mean", "rstd", "running_mean",
                                               "running_var", "num_batches_tracked", "ticket")] + \
               [("eps", ctypes.c_float), ("momentum", ctypes.c_float), ("tail", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class ConvLayer(ctypes.Structure):
    """struct gridgcn_conv_layer (include/gridgcn.h)."""
    _fields_ = [("W", ctypes.c_void_p), ("b", ctypes.c_void_p), ("K", ctypes.c_int32),
                ("ldw", ctypes.c_int32), ("cout", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_lib = None


def load():
    """Load the HIP library; raises RuntimeError (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libgridgcn_hip.so is not built (%s). Run `python -m grid_gcn_amd.build` "
            "(needs hipcc). grid_gcn_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError("cannot load %s: %s (grid_gcn_amd has no CPU fallback)" % (LIB_PATH, e))
    _lib = declare(lib, LIB_PATH)
    return _lib


def declare(lib, path="<library>"):
    """Attach the prototypes of include/gridgcn.h to an already opened library object and check its ABI version.
    load() applies it to the HIP library, the one library the package ever opens; the CPU-tier tests apply it to the
    host-side emulation of the same sources (tests/simt/) to call the same entries on numpy buffers."""
    LIB_PATH = path
    vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    pp = ctypes.POINTER(GridParams)
    lib.gridgcn_strerror.restype = ctypes.c_char_p
    lib.gridgcn_strerror.argtypes = [ci]
    lib.gridgcn_abi_version.restype = ci
    if lib.gridgcn_abi_version() != ABI_VERSION:
        raise RuntimeError("%s is ABI v%d, this package needs v%d: rebuild it "
                           "(python -m grid_gcn_amd.build --force)"
                           % (LIB_PATH, lib.gridgcn_abi_version(), ABI_VERSION))
    lib.gridgcn_set_option.restype = ci
    lib.gridgcn_set_option.argtypes = [ci, ci]
    lib.gridgcn_get_option.restype = ci
    lib.gridgcn_get_option.argtypes = [ci]
    lib.gridgcn_set_mlp_precision.restype = ci
    lib.gridgcn_set_mlp_precision.argtypes = [ci]
    lib.gridgcn_get_mlp_precision.restype = ci
    for name in ("gridgcn_gridify_workspace_bytes", "gridgcn_gridify_knn_workspace_bytes",
                 "gridgcn_gridify_up_workspace_bytes", "gridgcn_gridify_occaware_workspace_bytes",
                 "gridgcn_gridify_fast_rand_workspace_bytes"):
        f = getattr(lib, name)
        f.restype = ci
        f.argtypes = [ci, ci, pp, ctypes.POINTER(cs)]
    for name in ("gridgcn_gridify", "gridgcn_gridify_knn", "gridgcn_gridify_fast_rand"):
        f = getattr(lib, name)
        f.restype = ci
        f.argtypes = [vp, vp, ci, ci, pp, vp, vp, vp, vp, vp, vp, cs, vp]
    lib.gridgcn_gridify_occaware.restype = ci
    lib.gridgcn_gridify_occaware.argtypes = [vp, vp, ci, ci, pp, ctypes.c_float, vp, vp, vp, vp, vp,
                                             vp, cs, vp]
    lib.gridgcn_gridify_timed.restype = ci
    lib.gridgcn_gridify_timed.argtypes = [vp, vp, ci, ci, pp, vp, vp, vp, vp, vp, vp, cs, vp, ci,
                                          ctypes.POINTER(ctypes.c_float)]
    lib.gridgcn_gridify_up.restype = ci
    lib.gridgcn_gridify_up.argtypes = [vp, vp, vp, vp, ci, ci, pp, vp, vp, vp, cs, vp]
    lib.gridgcn_ball_knn.restype = ci
    lib.gridgcn_ball_knn.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, vp, vp]
    lib.gridgcn_ball_knn_grid_workspace_bytes.restype = ci
    lib.gridgcn_ball_knn_grid_workspace_bytes.argtypes = [ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_ball_knn_grid.restype = ci
    lib.gridgcn_ball_knn_grid.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, vp, vp,
                                          cs, vp]
    lib.gridgcn_knn.restype = ci
    lib.gridgcn_knn.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]
    lib.gridgcn_batch_take.restype = ci
    lib.gridgcn_batch_take.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
    lib.gridgcn_batch_take_backward.restype = ci
    lib.gridgcn_batch_take_backward.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
    lib.gridgcn_batch_take_backward_sorted.restype = ci
    lib.gridgcn_batch_take_backward_sorted.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, cs, vp]
    ll = ctypes.c_longlong
    lib.gridgcn_linear_fwd.restype = ci
    lib.gridgcn_linear_fwd.argtypes = [vp, ll, ci, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.gridgcn_linear_fwd_ld.restype = ci
    lib.gridgcn_linear_fwd_ld.argtypes = [vp, ll, ci, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, vp]
    lib.gridgcn_linear_fwd_direct_ld.restype = ci
    lib.gridgcn_linear_fwd_direct_ld.argtypes = [vp, ll, ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, ci, vp]
    lib.gridgcn_linear_fwd_direct_drop.restype = ci
    lib.gridgcn_linear_fwd_direct_drop.argtypes = [vp, ll, ci, ci, vp, vp, ci, ci, vp, vp, vp, ctypes.c_float,
                                                   ctypes.c_uint64, vp, vp]
    lib.gridgcn_linear_dw_drop.restype = ci
    lib.gridgcn_linear_dw_drop.argtypes = [vp] * 11 + [ll, ci, ci, ctypes.c_float, ctypes.c_uint64, vp, vp, vp,
                                           cs, vp]
    lib.gridgcn_linear_fwd_direct_fin.restype = ci
    lib.gridgcn_linear_fwd_direct_fin.argtypes = [vp, ll, ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, ci,
                                                  ctypes.POINTER(BnFin), vp]
    lib.gridgcn_gemm_small.restype = ci
    lib.gridgcn_gemm_small.argtypes = [ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp, cs, vp]
    lib.gridgcn_gemm_small_workspace_bytes.restype = ci
    lib.gridgcn_gemm_small_workspace_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_linear_bwd_fin.restype = ci
    lib.gridgcn_linear_bwd_fin.argtypes = [vp] * 19 + [ci, ll, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp,
                                                       ci, vp, cs, vp]
    lib.gridgcn_linear_bwd_ld.restype = ci
    lib.gridgcn_linear_bwd_ld.argtypes = [vp] * 16 + [ci, ll, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp,
                                                      ci, vp, cs, vp]
    lib.gridgcn_linear_bwd_workspace_bytes.restype = ci
    lib.gridgcn_linear_bwd_workspace_bytes.argtypes = [ll, ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_linear_bwd.restype = ci
    lib.gridgcn_linear_bwd.argtypes = [vp] * 16 + [ci, ll, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp,
                                                   ci, vp, cs, vp]
    lib.gridgcn_pairmax_fwd.restype = ci
    lib.gridgcn_pairmax_fwd.argtypes = [vp] * 6 + [ll, ci, ci, vp, ci, vp, vp, vp]
    lib.gridgcn_pairmax_bwd.restype = ci
    lib.gridgcn_pairmax_bwd.argtypes = [vp] * 12 + [ll, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    lib.gridgcn_pairmax_bwd_masked.restype = ci
    lib.gridgcn_pairmax_bwd_masked.argtypes = [vp] * 10 + [ll, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    lib.gridgcn_att_bwd_noz_workspace_bytes.restype = ci
    lib.gridgcn_att_bwd_noz_workspace_bytes.argtypes = [ll, ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_ball_knn_ld.restype = ci
    lib.gridgcn_ball_knn_ld.argtypes = [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ctypes.c_float, ci, vp, vp]
    lib.gridgcn_ball_knn_grid_ld.restype = ci
    lib.gridgcn_ball_knn_grid_ld.argtypes = [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ctypes.c_float, ci, vp,
                                             vp, ctypes.c_size_t, vp]
    lib.gridgcn_bn_finalize_tail.restype = ci
    lib.gridgcn_bn_finalize_tail.argtypes = [vp, vp, vp, ll, ctypes.c_float, ctypes.c_float, ci, ci, vp, vp,
                                             vp, vp, vp, vp, vp, vp]
    lib.gridgcn_softmax_ce_loss.restype = ci
    lib.gridgcn_softmax_ce_loss.argtypes = [vp, ci, ci, vp, ll, ci, vp, vp, vp, vp]
    lib.gridgcn_colsum_f32.restype = ci
    lib.gridgcn_colsum_f32.argtypes = [vp, ll, ci, ci, vp, vp, vp]
    lib.gridgcn_cat_mask.restype = ci
    lib.gridgcn_cat_mask.argtypes = [vp, ci, ci, vp, ci, ci, vp, vp, ci, vp, ci, ll, vp]
    lib.gridgcn_mask_sum.restype = ci
    lib.gridgcn_mask_sum.argtypes = [vp, ci, vp, ci, ci, ci, vp, vp, ll, vp]
    lib.gridgcn_adam_step.restype = ci
    lib.gridgcn_adam_step.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, ctypes.c_float, vp, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ci, vp]
    lib.gridgcn_gemm_bias.restype = ci
    lib.gridgcn_gemm_bias.argtypes = [ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp]
    lib.gridgcn_att_fwd_noz_workspace_bytes.restype = ci
    lib.gridgcn_att_fwd_noz_workspace_bytes.argtypes = [ll, ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_att_bwd_noz_mom_supported.restype = ci
    lib.gridgcn_att_bwd_noz_mom_supported.argtypes = [ll, ci, ci, ci]
    lib.gridgcn_att_moments_offset.restype = ci
    lib.gridgcn_att_moments_offset.argtypes = [ll, ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_att_bwd_noz_mom.restype = ci
    lib.gridgcn_att_bwd_noz_mom.argtypes = [vp] * 13 + [ci, ll, ci, ci] + [vp] * 8 + [vp, cs, vp]
    lib.gridgcn_att_bn2_moments.restype = ci
    lib.gridgcn_att_bn2_moments.argtypes = ([vp] * 7 + [ll, ci, ci, ctypes.c_float, ctypes.c_float] + [vp] * 8
                                            + [vp, cs, vp])
    lib.gridgcn_att_pairmax_fwd_supported.restype = ci
    lib.gridgcn_att_pairmax_fwd_supported.argtypes = [ll, ci, ci, ci, ci, ci, ll]
    lib.gridgcn_att_pairmax_fwd.restype = ci
    lib.gridgcn_att_pairmax_fwd.argtypes = [vp] * 5 + [ci, ci, ci] + [vp] * 9 + [ll, ci, ci, ci, vp, ci, vp, vp, vp]
    lib.gridgcn_att_bwd_noz.restype = ci
    lib.gridgcn_att_bwd_noz.argtypes = [vp] * 13 + [ci, ll, ci, ci] + [vp] * 8 + [vp, cs, vp]
    lib.gridgcn_bn_relu_apply.restype = ci
    lib.gridgcn_bn_relu_apply.argtypes = [vp, vp, vp, vp, ll, ci, ci, vp]
    lib.gridgcn_bn_relu_dropout_apply.restype = ci
    lib.gridgcn_bn_relu_dropout_apply.argtypes = [vp, vp, vp, vp, ll, ci, ci, ctypes.c_float,
                                                  ctypes.c_uint64, vp, vp]
    lib.gridgcn_linear_dx.restype = ci
    lib.gridgcn_linear_dx.argtypes = [vp] * 14 + [ci, ll, ci, ci, ci, ctypes.c_float,
                                                  ctypes.c_uint64, vp, vp, vp, vp]
    lib.gridgcn_bn_relu_bwd_reduce.restype = ci
    lib.gridgcn_bn_relu_bwd_reduce.argtypes = [vp, vp, vp, vp, vp, vp, ll, ci, ci, vp, vp]
    lib.gridgcn_bn_relu_bwd_elemt.restype = ci
    lib.gridgcn_bn_relu_bwd_elemt.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ll, ci, vp, vp]
    lib.gridgcn_pack_linear.restype = ci
    lib.gridgcn_pack_linear.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.gridgcn_edge_inputs_rows.restype = ci
    lib.gridgcn_edge_inputs_rows.argtypes = [vp, vp, vp] + [ci] * 9 + [vp, vp, vp]
    lib.gridgcn_edge_inputs_rows_backward.restype = ci
    lib.gridgcn_edge_inputs_rows_backward.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, cs, vp]
    lib.gridgcn_take_backward_workspace_bytes.restype = ci
    lib.gridgcn_take_backward_workspace_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_linear_fwd_direct.restype = ci
    lib.gridgcn_linear_fwd_direct.argtypes = [vp, ll, ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, vp]
    lib.gridgcn_linear_fwd_direct2.restype = ci
    lib.gridgcn_linear_fwd_direct2.argtypes = [vp, ci, ci, vp, ci, ci, ll, vp, vp, vp, ci, ci, ci,
                                               vp, vp, vp, vp, vp]
    lib.gridgcn_ctx_max.restype = ci
    lib.gridgcn_ctx_max.argtypes = [vp, vp, vp] + [ci] * 6 + [vp, vp, vp]
    lib.gridgcn_ctx_max_backward.restype = ci
    lib.gridgcn_ctx_max_backward.argtypes = [vp, vp, ll, ci, ci, vp, vp]
    lib.gridgcn_bn_dz_segsum.restype = ci
    lib.gridgcn_bn_dz_segsum.argtypes = [vp] * 8 + [ll, ci, ci, vp, vp]
    lib.gridgcn_bn_stats.restype = ci
    lib.gridgcn_bn_stats.argtypes = [vp, ll, ci, ci, vp, vp]
    lib.gridgcn_sparse_add.restype = ci
    lib.gridgcn_sparse_add.argtypes = [vp, vp, ll, ci, ci, vp, vp]
    cf = ctypes.c_float
    lib.gridgcn_bn_finalize.restype = ci
    lib.gridgcn_bn_finalize.argtypes = [vp, vp, vp, ll, cf, cf, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gridgcn_bn_bwd_finalize.restype = ci
    lib.gridgcn_bn_bwd_finalize.argtypes = [vp, ll, ci, vp, vp, vp, vp, vp]
    lib.gridgcn_edge_geo_forward_workspace_bytes.restype = ci
    lib.gridgcn_edge_geo_forward_workspace_bytes.argtypes = [ci, ci, ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_edge_geo_forward.restype = ci
    lib.gridgcn_edge_geo_forward.argtypes = [vp, vp, vp, vp] + [ci] * 7 + [vp] * 7 + [cs, vp]
    lib.gridgcn_edge_lin0_forward.restype = ci
    lib.gridgcn_edge_lin0_forward.argtypes = [vp, vp, vp, vp] + [ci] * 7 + [vp] * 6
    lib.gridgcn_edge_lin0_backward.restype = ci
    lib.gridgcn_edge_lin0_backward.argtypes = [vp] * 15 + [ci] * 5 + [vp, vp, vp, cs, vp]
    lib.gridgcn_pack_desc_fill.restype = ci
    lib.gridgcn_pack_desc_fill.argtypes = [ctypes.POINTER(PackDesc)]
    lib.gridgcn_pack_linear_batch.restype = ci
    lib.gridgcn_pack_linear_batch.argtypes = [vp, ci, ci, vp]
    lib.gridgcn_pairmax_fwd_src_z.restype = ci
    lib.gridgcn_pairmax_fwd_src_z.argtypes = [vp] * 5 + [ci, ci, ci] + [vp, ci] + [vp] * 4 + [ll, ci, ci, vp,
                                                                                       ci, vp, vp, vp]
    lib.gridgcn_pairmax_fwd_src.restype = ci
    lib.gridgcn_pairmax_fwd_src.argtypes = [vp] * 5 + [ci, ci, ci] + [vp] * 5 + [ll, ci, ci, vp, ci,
                                                                              vp, vp, vp]
    lib.gridgcn_edge_lin0_backward_sparse_workspace_bytes.restype = ci
    lib.gridgcn_edge_lin0_backward_sparse_workspace_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(cs)]
    lib.gridgcn_edge_lin0_backward_sparse.restype = ci
    lib.gridgcn_edge_lin0_backward_sparse.argtypes = [vp] * 14 + [ci] * 5 + [vp] * 5 + [cs, vp]
    lib.gridgcn_edge_lin0_backward_sparse_geo.restype = ci
    lib.gridgcn_edge_lin0_backward_sparse_geo.argtypes = [vp] * 14 + [ci] * 5 + [vp] * 4 + [cs, vp]
    lib.gridgcn_att_max_eval.restype = ci
    lib.gridgcn_att_max_eval.argtypes = [vp] * 14 + [ci] * 5 + [vp, ci, vp]
    lib.gridgcn_edge_lin0_dwg.restype = ci
    lib.gridgcn_edge_lin0_dwg.argtypes = [vp] * 9 + [ci, vp, ci, vp]
    lib.gridgcn_softmax_ce_fwd.restype = ci
    lib.gridgcn_softmax_ce_fwd.argtypes = [vp, ci, ci, vp, ll, ci, vp, vp, vp]
    lib.gridgcn_softmax_ce_bwd.restype = ci
    lib.gridgcn_softmax_ce_bwd.argtypes = [vp, ci, ci, vp, ll, ci, vp, vp, vp, vp, vp, vp]
    lib.gridgcn_colsum.restype = ci
    lib.gridgcn_colsum.argtypes = [vp, ll, ci, ci, vp, vp]
    lib.gridgcn_edge_inputs.restype = ci
    lib.gridgcn_edge_inputs.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp]
    lib.gridgcn_edge_inputs_backward.restype = ci
    lib.gridgcn_edge_inputs_backward.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
    lib.gridgcn_gridconv_forward.restype = ci
    lib.gridgcn_gridconv_forward.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci,
                                             ctypes.POINTER(ConvLayer), ctypes.POINTER(ConvLayer),
                                             vp, vp]
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, load().gridgcn_strerror(rc).decode()))
