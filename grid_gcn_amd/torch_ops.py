"""`torch.ops.gridgcn.*`: the PyTorch analogue of the reference's "load the .so, the ops appear".

The reference registers its operators into MXNet's registry when gridifyop/additional.so is loaded
(ctypes.CDLL before `import mxnet`, train_gpu_ggcn_mdl40.py:7-11; MXNET_REGISTER_OP_PROPERTY /
NNVM_REGISTER_OP in gridify.cc:63-67, gridify_up.cc:62-68, gridifyknn.cc:63-67, ball_k_nn.cc:14,
k_nn.cc:14).  Importing this module (done by `import grid_gcn_amd`) registers the same operators in
the dispatcher namespace `gridgcn`, with fake (meta) kernels so that they trace under
torch.compile / export:

    torch.ops.gridgcn.gridify(data, actual_numpoints, max_p_grid=, max_o_grid=, kernel_size=,
                              stride=, loc=, coord_shift=, voxel_size=, grid_size=, seed=0)
        -> (nebidx, nebidxmsk, cent, centmsk, actual_centnum)
    torch.ops.gridgcn.gridify_knn(...)            same
    torch.ops.gridgcn.gridify_fast_rand(...)      same (the fast_rand build variant)
    torch.ops.gridgcn.gridify_occaware(..., beta=1.0)   same + coverage-aware sampling (parity
                                                  unpinned: binary-only in the reference)
    torch.ops.gridgcn.gridify_up(downdata, updata, down_actual_numpoints, up_actual_numpoints,
                                 max_p_grid=, max_o_grid=, kernel_size=, coord_shift=, voxel_size=,
                                 grid_size=, seed=0) -> (nebidx, nebidxmsk)
    torch.ops.gridgcn.ball_knn(unknown, known, downnum, upnum, k=3, radius=0.1) -> idx
    torch.ops.gridgcn.knn(unknown, known, downnum, upnum, k=3) -> idx
    torch.ops.gridgcn.batch_take(data, index) -> out          (differentiable w.r.t. data)

Index ops have no autograd formula, as in the reference (gridify-inl.h:227-231, ball_k_nn.cc:60).
GPU only: there is no CPU kernel (gridify.cc:28-39 is LOG(FATAL) in the reference as well).
The reference's BallKNN/KNN leave rows >= upnum untouched in a caller-provided buffer; the
functional ops return a fresh tensor whose untouched rows hold 0 (as grid_gcn_amd.ops does).
"""
from typing import List, Tuple

import torch
from torch import Tensor

from . import ops

_lib_def = torch.library.custom_op


@_lib_def("gridgcn::gridify", mutates_args=(), device_types="cuda")
def gridify(data: Tensor, actual_numpoints: Tensor, max_p_grid: int, max_o_grid: int,
            kernel_size: int, stride: int, loc: int, coord_shift: List[float],
            voxel_size: List[float], grid_size: List[int],
            seed: int = 0) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    return ops.Gridify(data, actual_numpoints, max_p_grid=max_p_grid, max_o_grid=max_o_grid,
                       kernel_size=kernel_size, stride=stride, loc=loc, coord_shift=coord_shift,
                       voxel_size=voxel_size, grid_size=grid_size, seed=seed)


@_lib_def("gridgcn::gridify_knn", mutates_args=(), device_types="cuda")
def gridify_knn(data: Tensor, actual_numpoints: Tensor, max_p_grid: int, max_o_grid: int,
                kernel_size: int, stride: int, loc: int, coord_shift: List[float],
                voxel_size: List[float], grid_size: List[int],
                seed: int = 0) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    return ops.GridifyKNN(data, actual_numpoints, max_p_grid=max_p_grid, max_o_grid=max_o_grid,
                          kernel_size=kernel_size, stride=stride, loc=loc,
                          coord_shift=coord_shift, voxel_size=voxel_size, grid_size=grid_size,
                          seed=seed)


def _gridify_fake(data, actual_numpoints, max_p_grid, max_o_grid, kernel_size, stride, loc,
                  coord_shift, voxel_size, grid_size, seed=0):
    B = data.shape[0]
    O, P = max_o_grid, max_p_grid
    return (data.new_empty((B, O, P), dtype=torch.int32), data.new_empty((B, O, P)),
            data.new_empty((B, O, 4)), data.new_empty((B, O)),
            data.new_empty((B, 1), dtype=torch.int32))


@_lib_def("gridgcn::gridify_occaware", mutates_args=(), device_types="cuda")
def gridify_occaware(data: Tensor, actual_numpoints: Tensor, max_p_grid: int, max_o_grid: int,
                     kernel_size: int, stride: int, loc: int, coord_shift: List[float],
                     voxel_size: List[float], grid_size: List[int], seed: int = 0,
                     beta: float = 1.0) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """`Gridify_occaware` (coverage-aware sampling; the reference ships it as a binary only --
    parity unpinned, see ops.Gridify_occaware)."""
    return ops.Gridify_occaware(data, actual_numpoints, max_p_grid=max_p_grid,
                                max_o_grid=max_o_grid, kernel_size=kernel_size, stride=stride,
                                loc=loc, coord_shift=coord_shift, voxel_size=voxel_size,
                                grid_size=grid_size, seed=seed, beta=beta)


@_lib_def("gridgcn::gridify_fast_rand", mutates_args=(), device_types="cuda")
def gridify_fast_rand(data: Tensor, actual_numpoints: Tensor, max_p_grid: int, max_o_grid: int,
                      kernel_size: int, stride: int, loc: int, coord_shift: List[float],
                      voxel_size: List[float], grid_size: List[int],
                      seed: int = 0) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """the `fast_rand` build of Gridify (gridifyop/fast_rand/): same operator, other sampling"""
    return ops.Gridify_fast_rand(data, actual_numpoints, max_p_grid=max_p_grid,
                                 max_o_grid=max_o_grid, kernel_size=kernel_size, stride=stride,
                                 loc=loc, coord_shift=coord_shift, voxel_size=voxel_size,
                                 grid_size=grid_size, seed=seed)


gridify.register_fake(_gridify_fake)
gridify_knn.register_fake(_gridify_fake)
gridify_fast_rand.register_fake(_gridify_fake)


@gridify_occaware.register_fake
def _(data, actual_numpoints, max_p_grid, max_o_grid, kernel_size, stride, loc, coord_shift,
      voxel_size, grid_size, seed=0, beta=1.0):
    return _gridify_fake(data, actual_numpoints, max_p_grid, max_o_grid, kernel_size, stride, loc,
                         coord_shift, voxel_size, grid_size, seed)


@_lib_def("gridgcn::gridify_up", mutates_args=(), device_types="cuda")
def gridify_up(downdata: Tensor, updata: Tensor, down_actual_numpoints: Tensor,
               up_actual_numpoints: Tensor, max_p_grid: int, max_o_grid: int, kernel_size: int,
               coord_shift: List[float], voxel_size: List[float], grid_size: List[int],
               seed: int = 0) -> Tuple[Tensor, Tensor]:
    return ops.GridifyUp(downdata, updata, down_actual_numpoints, up_actual_numpoints,
                         max_p_grid=max_p_grid, max_o_grid=max_o_grid, kernel_size=kernel_size,
                         coord_shift=coord_shift, voxel_size=voxel_size, grid_size=grid_size,
                         seed=seed)


@gridify_up.register_fake
def _(downdata, updata, down_actual_numpoints, up_actual_numpoints, max_p_grid, max_o_grid,
      kernel_size, coord_shift, voxel_size, grid_size, seed=0):
    B = downdata.shape[0]
    return (downdata.new_empty((B, max_o_grid, max_p_grid), dtype=torch.int32),
            downdata.new_empty((B, max_o_grid, max_p_grid)))


@_lib_def("gridgcn::ball_knn", mutates_args=(), device_types="cuda")
def ball_knn(unknown: Tensor, known: Tensor, downnum: Tensor, upnum: Tensor, k: int = 3,
             radius: float = 0.1) -> Tensor:
    return ops.BallKNN(unknown, known, downnum, upnum, k=k, radius=radius)


@_lib_def("gridgcn::knn", mutates_args=(), device_types="cuda")
def knn(unknown: Tensor, known: Tensor, downnum: Tensor, upnum: Tensor, k: int = 3) -> Tensor:
    return ops.KNN(unknown, known, downnum, upnum, k=k)


@ball_knn.register_fake
def _(unknown, known, downnum, upnum, k=3, radius=0.1):
    return unknown.new_empty((unknown.shape[0], unknown.shape[1], k), dtype=torch.int32)


@knn.register_fake
def _(unknown, known, downnum, upnum, k=3):
    return unknown.new_empty((unknown.shape[0], unknown.shape[1], k), dtype=torch.int32)


@_lib_def("gridgcn::batch_take", mutates_args=(), device_types="cuda")
def batch_take(data: Tensor, index: Tensor) -> Tensor:
    with torch.no_grad():
        return ops.batch_take_g(data.contiguous(), index)


@batch_take.register_fake
def _(data, index):
    return data.new_empty(tuple(index.shape) + (data.shape[2],))


@_lib_def("gridgcn::batch_take_backward", mutates_args=(), device_types="cuda")
def batch_take_backward(grad_out: Tensor, index: Tensor, N: int) -> Tensor:
    return ops.batch_take_g_backward(grad_out, index, N)


@batch_take_backward.register_fake
def _(grad_out, index, N):
    return grad_out.new_empty((index.shape[0], N, grad_out.shape[-1]))


def _take_setup(ctx, inputs, output):
    data, index = inputs
    ctx.save_for_backward(index)
    ctx.N = data.shape[1]


def _take_bwd(ctx, grad_out):
    (index,) = ctx.saved_tensors
    return torch.ops.gridgcn.batch_take_backward(grad_out, index, ctx.N), None


batch_take.register_autograd(_take_bwd, setup_context=_take_setup)
