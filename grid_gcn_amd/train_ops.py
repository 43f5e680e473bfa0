"""Training path on the hand-written gfx950 kernels: the host logic of one training step between
the index ops and the optimizer (kernels: csrc/gridgcn_direct.hip, gridgcn_bwdfused.hip, gridgcn_edgelin.hip,
gridgcn_pairmax.hip, gridgcn_scatter.hip, gridgcn_head.hip; LDS-staged fallbacks for odd shapes in
gridgcn_train.hip).  Everything goes through the C ABI of include/gridgcn.h.

The code lives in grid_gcn_amd/train/ (one module per block: common, mlp, edge, cls, head, evalpath, timers;
the path switches are ONE object, train/options.py: OPT); this module re-exports it under the names the models,
bench.py, the tools and the tests have always used.

A "chain" is a stack of (1x1 conv -> BatchNorm(batch statistics) -> ReLU) layers (mlp2d_c /
mlp1d_c, utils/ops.py:236-260):

  forward   per layer: pack the weight layouts (1 launch), Z_l = act_{l-1} * W_l + b_l on fp32 MFMA
            where act_{l-1} = relu(bn(Z_{l-1})) is applied while the rows are loaded (never
            materialised) and the epilogue accumulates the batch statistics of Z_l, then the
            BatchNorm bookkeeping (folded into the kernel's last workgroup).
  backward  per layer: dZ formed in registers, a dX kernel (with the previous layer's
            BatchNorm-backward sums in its epilogue) and a dW kernel + reduce -- or, for the 128-output
            per-point layers, ONE kernel for all three (csrc/gridgcn_bwdfused.hip).

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
from . import _lib  # noqa: F401
from .train.options import OPT, PathOptions  # noqa: F401
from .train.common import (  # noqa: F401
    LaunchTimers, PACKS, RawLink, _ARENA, _Cat2, _CatMask, _Chain, _GEMM_WS, _IDENT, _PARAM_GEN,
    _PackCache, _ZEROS, _ZeroArena, _cached_zeros, _chain_backward, _chain_forward, _dw_direct_ok,
    _gemm_small, _identity_consts, _mm_nn, _mm_nt, _momentum, _rows2d, _small_ok, _stats_written,
    _tn_matmul, _zeros, alias_columns, cat_mask, get_mlp_precision, pack_groups, pack_tiles, packed_sizes,
    params_changed, release_packs_hook, reset_zero_arena, set_mlp_precision, supported,)
from .train.mlp import (  # noqa: F401
    _MLPTrain, _WideLayerTrain, _pack_tmp, _padded_supported, _wide_direct_ok, mlp_bn_relu_train,
    mlp_wide_train, wide_supported,)
from .train.edge import (  # noqa: F401
    _EdgeBlockSrcTrain, _EdgeBlockTrain, _att_bwd_noz, edge_block_src_supported, edge_block_src_train,
    edge_block_supported, edge_block_train,)
from .train.cls import (  # noqa: F401
    _EdgeBlockClsTrain, _pack_bwd_part, edge_block_cls_supported, edge_block_cls_train,)
from .train.head import (  # noqa: F401
    _HeadTrain, _LinearMM, _LinearPlain, _SoftmaxCE, _pad_is_zero, dropout_mask, head_supported,
    head_train, linear_mm, linear_plain_supported, linear_plain_train, softmax_ce,)
from .train.evalpath import (  # noqa: F401
    _EVAL_BN, _EVAL_W, _bn_eval_vectors, _chain_eval_raw, _eval_packed, clear_eval_cache,
    edge_block_cls_eval, edge_block_src_eval, edge_block_src_eval_supported, head_eval, mlp_bn_relu_eval,)
from .train.timers import (  # noqa: F401
    median_ms, time_att_bwd_noz, time_linear_bwd, time_linear_fwd,)
