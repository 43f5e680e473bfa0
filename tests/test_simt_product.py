"""CPU tier: the product's HOST code driving the product's KERNELS with no GPU in the machine.

`emulated_gpu()` (tests/simt/emu.py) points grid_gcn_amd._lib at the host-side emulation of the library and lets CPU
tensors pass for device tensors; the functions below are then the GPU tier's own test bodies (imported from the
tests/test_gpu_*.py modules, DEV switched to "cpu") at sizes the emulator finishes in seconds: operand packing,
forward / backward chains with their BatchNorm hand-offs, the segmentation and classification edge blocks against the
stock PyTorch modules and the float64 fixtures, the round-6 opt-in paths.  Every allocation is poisoned with NaN.

What this proves and what it does not: tests/simt/simt_hip.h.  GG_SIMT_FULL=1 adds the slow cases (all four float64
fixtures with their gradients, a whole training step of the segmentation net against the CPU model on the oracle's
index operators: ~10 minutes)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from simt import emu  # noqa: E402

FULL = os.environ.get("GG_SIMT_FULL") == "1"
slow = pytest.mark.skipif(not FULL, reason="slow under emulation: set GG_SIMT_FULL=1")


@pytest.fixture(autouse=True)
def _gpu():
    c0 = emu.counters()
    with emu.emulated_gpu():
        yield
    c1 = emu.counters()
    assert c1[2] == c0[2], "a cross-lane read took a lane outside the set executing the operation"
    assert c1[3] == c0[3], "a cross-lane operation was reached in divergent control flow"


def _gpu_module(name):
    """a tests/test_gpu_*.py module with its device constant pointed at the emulator's tensors"""
    import importlib
    m = importlib.import_module(name)
    if hasattr(m, "DEV"):
        m.DEV = "cpu"
    return m


@pytest.mark.parametrize("E,cin,dims", [(500, 8, [32, 32, 64]), (300, 3, [32, 64]), (77, 132, [128]),
                                        (1000, 264, [128, 256]), (31, 24, [8, 128]), (2100, 64, [256, 32])])
def test_mlp_chain_forward_backward(E, cin, dims):
    """gridgcn_pack_linear, gridgcn_linear_fwd_direct_fin (BatchNorm finalisation by the last workgroup), the LDS-staged
    forward for odd widths, gridgcn_linear_bwd_fin (dX / dW / reduce, the column-split forms): against nn.Linear ->
    BatchNorm1d -> ReLU stacks, outputs, input and parameter gradients, running statistics"""
    _gpu_module("test_gpu_train_ops").test_mlp_train_matches_torch(E, cin, dims)


@pytest.mark.parametrize("E,cin,dims", [(1500, 72, [64, 256, 32]), (95, 64, [256])])
def test_col_split_equals_whole_rows(E, cin, dims):
    _gpu_module("test_gpu_train_ops").test_col_split_equals_whole_rows(E, cin, dims)


def _to_cpu(fn):
    """a GPU test body that spells its device "cuda:0": .to("cuda:0") lands on the emulator's (CPU) tensors"""
    def run(*a, **k):
        t_to, m_to = torch.Tensor.to, torch.nn.Module.to
        fix = lambda args: tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in args)  # noqa: E731
        torch.Tensor.to = lambda self, *aa, **kk: t_to(self, *fix(aa), **kk)
        torch.nn.Module.to = lambda self, *aa, **kk: m_to(self, *fix(aa), **kk)
        try:
            return fn(*a, **k)
        finally:
            torch.Tensor.to, torch.nn.Module.to = t_to, m_to
    return run


@pytest.mark.parametrize("cin,pt,att,O,P", [(32, [32, 64], [32, 64, 64], 9, 32),
                                            pytest.param(0, [64, 64, 128], [64, 128, 128], 40, 64, marks=slow)])
def test_cls_edge_block(cin, pt, att, O, P):
    """the classification GridConv edge block (two-source first attention conv, context as a per-centre bias): forward,
    every gradient, running statistics, evaluation -- against the stock modules"""
    m = _gpu_module("test_model_cls")
    _to_cpu(m.test_cls_edge_block_kernels_match_stock_modules)(cin, pt, att, O, P)


@pytest.mark.parametrize("name", ["gridconv_seg_L1", pytest.param("gridconv_seg_L0", marks=slow),
                                  pytest.param("gridconv_up2", marks=slow),
                                  pytest.param("gridconv_cls_L0", marks=slow)])
def test_gridconv_float64_fixtures(name):
    """the GPU tier's fixture tests (north_star's 1e-5 bar, relative to max(1, max|x|)): evaluation kernel and
    training kernels of one GridConv layer against tests/golden/gridconv_*.npz"""
    m = _gpu_module("test_gpu_gridconv_golden")
    m.test_hip_eval_matches_fixture(name)
    m.test_hip_train_not_worse_than_stock_fp32(name)


@slow
@pytest.mark.parametrize("name", ["gridconv_seg_L1", "gridconv_up2", pytest.param(
    "gridconv_cls_L0", marks=pytest.mark.xfail(strict=False, reason=(
        "under emulation ONE hidden ReLU input of the last point conv (row 31957, channel 4 of 8.4 M) is +2^-24 where "
        "float64 has it <= 0 -- the host's 1/sqrtf against the GPU's v_rsq_f32 in the BatchNorm's rstd is enough -- and "
        "the open mask moves the point branch's weight gradients by up to 9e-4 of their scale (the fp32 discontinuity "
        "(ii) of DESIGN section 2); every library call of the path reproduces numpy from its own inputs at 3e-7 - "
        "2.5e-6, the attention branch's gradients are within 2.3e-6; on the GPU the test measured 3.5e-6")))])
def test_gridconv_gradients_against_float64(name):
    _gpu_module("test_gpu_gridconv_golden").test_hip_gradients_bounded_by_stock_fp32(name)


# GPU-tier test bodies of tests/test_gpu_train_ops.py at (their own) parameters the emulator finishes in seconds: the
# whole file was swept once (round 6: every body passes except the ones that need more than ~90 s or a CUDA runtime
# call of their own); these stay in the CPU tier
GPU_BODIES = [
    ("test_att_pairmax_fwd_matches_the_materialised_path", (2, 150, 33, True)),
    ("test_att_pairmax_fwd_matches_the_materialised_path", (3, 150, 700, False)),
    ("test_att_max_eval_tile_kernel_matches_the_general_kernel", (3, 150, 700, False)),
    ("test_edge_block_train_matches_torch", (1, 33, 128, 3, [32, 32, 64])),
    ("test_edge_block_train_matches_torch", (2, 40, 32, 67, [64, 64, 128])),
    ("test_edge_lin0_backward_sparse_fixed_point_matches_float64", (2, 50, 300, 5, 40, True)),
    ("test_gemm_small_matches_torch", (2048, 64, 64, 3)),
    ("test_gemm_small_matches_torch", (192, 256, 128, 3)),
    ("test_head_train_matches_torch", (333, 64, [64], 13, 0.3)),
    ("test_linear_plain_and_loss_match_torch", (4000, 128, 21)),
    ("test_linear_plain_and_loss_match_torch", (333, 64, 13)),
    ("test_mlp_eval_matches_modules", (5000, 4, [128])),
    ("test_mlp_eval_matches_modules", (300, 67, [64, 32])),
    ("test_pack_linear_layouts", (128, 131)),
    ("test_pairmax_fwd_first_argmax_exact", (192, 32, 256)),
    ("test_softmax_ce_matches_torch", (1000, 21)),
    ("test_softmax_ce_matches_torch", (257, 8)),
    ("test_softmax_ce_class_weights_match_weighted_gradient_op", ()),
    ("test_wide_stack_rocblas_plus_bn_kernels_matches_stock", (700, 8, [768])),
    ("test_unsupported_width_falls_to_modules", ()),
]


@pytest.mark.parametrize("body,args", GPU_BODIES, ids=["%s-%s" % (b[5:40], "x".join(str(a)[:8] for a in c)) for b, c in GPU_BODIES])
def test_gpu_tier_bodies_of_the_training_kernels(body, args):
    _to_cpu(getattr(_gpu_module("test_gpu_train_ops"), body))(*args)


# ... and of the other GPU-tier files: the glue kernels (Adam, concat + mask, edge inputs), the statistics of the
# source-side first conv, batch_take with its three backward forms, the inference edge kernels
MORE_BODIES = [
    ("test_gpu_glue", "test_adam_matches_torch", (1e-2,)),
    ("test_gpu_glue", "test_adam_many_tensors_none_grads_and_mxnet_form", ()),
    ("test_gpu_glue", "test_cat_mask_matches_torch", (4, 64, True, True)),
    ("test_gpu_glue", "test_cat_mask_matches_torch", (3, None, False, True)),
    ("test_gpu_glue", "test_cat_mask_matches_torch", (3, 5, True, True)),
    ("test_gpu_glue", "test_cat_mask_matches_torch", (4, 6, False, False)),
    ("test_gpu_glue", "test_ball_knn_reads_wider_rows_in_place", ()),
    ("test_gpu_glue", "test_edge_geo_forward_statistics_match_the_edge_pass", (2, 24, 256, 8, 64, True)),
    ("test_gpu_glue", "test_edge_geo_forward_statistics_match_the_edge_pass", (1, 300, 77, 5, 32, False)),
    ("test_gpu_parity", "test_batch_take_matches_oracle_and_grad", ()),
    ("test_gpu_parity", "test_error_behaviour", ()),
    ("test_gpu_gridconv", "test_edge_inputs_forward_backward", (64, 3)),
    ("test_gpu_gridconv", "test_edge_inputs_forward_backward", (33, 3)),
    ("test_gpu_gridconv", "test_edge_inputs_rows_layout", (36, 3)),
    ("test_gpu_gridconv", "test_edge_inputs_rows_layout", (0, 3)),
    ("test_gpu_gridconv", "test_edge_block_source_side_first_conv", (64, 3, [64, 64, 128], 90, 12)),
    ("test_gpu_gridconv", "test_att_max_eval_kernel_equals_two_kernel_path", (64, 64, 90, 7)),
    ("test_gpu_gridconv", "test_att_max_eval_kernel_equals_two_kernel_path", (32, 128, 41, 33)),
    ("test_gpu_train_ops", "test_wide_layers_without_rocblas_match_torch", (300, 64, [512])),
    ("test_gpu_train_ops", "test_pairmax_fwd_first_argmax_exact", (300, 5, 64)),     # one thread per (centre, quad)
    ("test_gpu_train_ops", "test_pairmax_fwd_first_argmax_exact", (77, 5, 30)),      # ... per (centre, channel)
    ("test_gpu_train_ops", "test_mlp_train_matches_torch", (200, 13, [32])),         # padded input rows
    ("test_gpu_train_ops", "test_mlp_train_matches_torch", (300, 100, [64, 16])),    # LDS-staged dW
    ("test_gpu_train_ops", "test_mlp_train_matches_torch", (150, 7, [4, 2])),        # the first-generation GEMM family
    ("test_gpu_train_ops", "test_pack_cache_batch_launch_equals_single_packs", ()),
    ("test_gpu_glue", "test_adam_state_dict_round_trip_and_tensor_lr", ()),
    ("test_gpu_glue", "test_adam_second_checkpoint_carries_the_advanced_step", ()),     # (round 6: never ran on a GPU)
    ("test_gpu_glue", "test_adam_checkpoints_travel_to_and_from_torch_adam", ()),
    ("test_gpu_fuzz", "test_gridconv_training_block_random_shapes", (0,)),              # random (O, P, widths) blocks
    ("test_gpu_fuzz", "test_gridconv_training_block_random_shapes", (1,)),
    ("test_gpu_fuzz", "test_gridconv_training_block_random_shapes", (2,)),
    ("test_gpu_fuzz", "test_gridconv_training_block_random_shapes", (3,)),
    pytest.param("test_gpu_train_ops", "test_linear_bwd_fused128_matches_separate_kernels", (32768, 128, True, 0),
                 marks=slow),
    pytest.param("test_gpu_gridconv", "test_full_model_eval_fused_vs_torch", (), marks=slow),
]


def _body_id(v):
    return v if isinstance(v, str) and v.startswith("test_") else None


@pytest.mark.parametrize("module,body,args", MORE_BODIES)
def test_gpu_tier_bodies_of_the_other_files(module, body, args):
    _to_cpu(getattr(_gpu_module(module), body))(*args)


def _up_layer_case(seed, B=1, Nsrc=96, O=640):
    """an up layer of the segmentation net at toy size: [B, Nsrc] source points with 128 features, O up points with
    5 neighbours each (P = 5: the Z2-free attention pair), centre MLP + update MLP"""
    from grid_gcn_amd.gridconv import SubGUpdate
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    layer = SubGUpdate(128, [128], 3, False, center_in=4, center_dim=[128], out_dim=[128], bn_decay=0.9).train()
    for mod in layer.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.3)
    src = torch.cat([torch.rand(B, Nsrc, 3, generator=g) * 2 - 1, torch.ones(B, Nsrc, 1),
                     torch.randn(B, Nsrc, 128, generator=g).clamp_(min=0)], 2)
    upl = torch.cat([torch.rand(B, O, 3, generator=g) * 2 - 1, torch.ones(B, O, 1)], 2)
    nebidx = torch.randint(-1, Nsrc, (B, O, 5), generator=g, dtype=torch.int32)
    return layer, src, upl, nebidx


def test_up_layer_z2_free_pair_and_round6_moments():
    """An up layer through the Z2-free attention forward / backward (gridgcn_att_bn2_moments, gridgcn_att_pairmax_fwd,
    gridgcn_att_bwd_noz) against the stock modules; then the same step with OPT.NOZ_BWD_MOMENTS -- the backward
    takes S1 / S2 from the forward's moments (round 6) -- against the shipped form: same output, gradients within
    fp32 round-off of each other."""
    from grid_gcn_amd import ops
    from grid_gcn_amd.train.options import OPT
    layer, src, upl, nebidx = _up_layer_case(5)
    ref = copy.deepcopy(layer)
    ref.mfma_train = False
    state = copy.deepcopy(layer.state_dict())
    cot = torch.randn(src.shape[0], nebidx.shape[1], 128)

    def step(m, mom, stock=False):
        m.load_state_dict(state)
        m.zero_grad(set_to_none=True)
        s = src.clone().requires_grad_(True)
        with OPT.override(NOZ_BWD_MOMENTS=mom):
            if stock:
                y = m(upl[..., 0:3], ops.batch_take_g(s, nebidx), None, center_ori_feats=upl)
            else:
                y = m.forward_src(upl, s, nebidx, None, center_ori_feats=upl)
            (y * cot).sum().backward()
        return y.detach(), s.grad.detach(), [p.grad.detach().clone() for p in m.parameters()]

    y0, gs0, gp0 = step(ref, False, stock=True)
    y1, gs1, gp1 = step(layer, False)
    y2, gs2, gp2 = step(layer, True)
    scale = max(1.0, float(y0.abs().max()))
    assert float((y1 - y0).abs().max()) <= 1e-5 * scale
    assert torch.equal(y2, y1)                                   # the forward is the same code
    # (feature columns: the kernels treat the coordinates as data -- the model feeds them detached)
    gs0, gs1, gs2 = gs0[..., 4:], gs1[..., 4:], gs2[..., 4:]
    gscale = max(1e-6, float(gs0.abs().max()))
    assert float((gs1 - gs0).abs().max()) <= 2e-4 * gscale
    assert float((gs2 - gs1).abs().max()) <= 1e-6 * gscale       # dX path: same arithmetic, same order
    for (n, _), a, b, c in zip(layer.named_parameters(), gp0, gp1, gp2):
        if n.endswith("lin.bias"):
            continue                                             # (a bias in front of a BatchNorm: exact 0 vs noise)
        s_ = max(1e-6, float(a.abs().max()))
        assert float((b - a).abs().max()) <= 5e-4 * s_, n
        assert float((c - b).abs().max()) <= 5e-6 * s_, n


@slow
def test_segmentation_training_step_against_the_cpu_model():
    """one forward + backward of GGCNSeg (1 x 1024 points) through every kernel of the training path against the same
    net on the oracle's index operators and the stock modules: the loss to fp32 round-off; the gradient as a vector
    (a 24-row BatchNorm in the coarsest layer makes single entries discretely sensitive at this size)"""
    from grid_gcn_amd import model, synth
    from oracle.torch_index_ops import OracleIndexOps
    torch.manual_seed(0)
    data, npn = synth.make_batch(1, 1024, "planes")
    cfg = dict(model.SEG_8192, dropout=0.0)
    with emu.emulated_gpu(poison=False):      # (plain CPU reference: no emulated kernel in it)
        pass
    net_cpu = model.GGCNSeg(cfg, index_ops=OracleIndexOps).train()
    net_emu = model.GGCNSeg(cfg)
    net_emu.load_state_dict(copy.deepcopy(net_cpu.state_dict()))
    net_emu.train()
    x, n = torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn)
    lab = torch.randint(1, 21, (1, 1024))
    loss = model.seg_loss(net_emu(x, n), lab)
    loss.backward()
    torch.Tensor.is_cuda_saved = None
    # the reference outside the emulation: plain CPU tensors on the stock path
    del torch.Tensor.is_cuda
    try:
        loss_cpu = model.seg_loss(net_cpu(x, n), lab)
        loss_cpu.backward()
    finally:
        torch.Tensor.is_cuda = property(lambda self: True)
    assert abs(float(loss) - float(loss_cpu)) <= 5e-6 * max(1.0, abs(float(loss_cpu)))
    ga = torch.cat([p.grad.reshape(-1) for p in net_cpu.parameters()]).double()
    gb = torch.cat([p.grad.reshape(-1) for p in net_emu.parameters()]).double()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert 1.0 - cos < 2e-3, cos


def _tiny_cls_cfg():
    """the classifier of BASELINE configs[1] with its layer WIDTHS (what the kernels are specialised on) on a grid small
    enough for the emulator: 3 Gridify layers 8^3 / 4^3 / 1 voxel, k = 3 / 3 / 1, P = 32 (the edge kernels want whole
    32-row tiles per centre), O = 16 / 4 / 1"""
    from grid_gcn_amd import model_cls
    grid = dict(num_points=160, coord_shift=[1.0, 1.0, 1.0], loc=1, down=[
        dict(voxel_size=[0.25] * 3, grid_size=[8] * 3, kernel_size=3, max_p_grid=32, max_o_grid=16),
        dict(voxel_size=[0.5] * 3, grid_size=[4] * 3, kernel_size=3, max_p_grid=32, max_o_grid=4),
        dict(voxel_size=[2.0] * 3, grid_size=[1] * 3, kernel_size=1, max_p_grid=32, max_o_grid=1)])
    return dict(model_cls.CLS_MN40, grid=grid, dropout=0.0)


def test_classifier_training_step_against_the_cpu_model():
    """GGCNCls, forward + backward, through model_cls.py's own host code and every kernel of the classification path
    (Gridify x 3, the context max / scatter, the two-source attention GEMMs with a bias per centre, the dZ segment sums,
    the FC head) against the same net on the oracle's index operators and the stock modules.  The full-size GPU tests of
    this net (tests/test_model_cls.py) cannot run here (144 GMAC per step); this is the whole graph at 3 x 160 points."""
    from grid_gcn_amd import model_cls, synth
    from oracle.torch_index_ops import OracleIndexOps
    torch.manual_seed(0)
    cfg = _tiny_cls_cfg()
    data, npn = synth.make_batch(3, 160, "ball")
    npn = npn.copy()
    npn[1, 0] = 131
    net_cpu = model_cls.GGCNCls(cfg, index_ops=OracleIndexOps, fixed_seed=True).train()
    net_emu = model_cls.GGCNCls(cfg, fixed_seed=True)
    net_emu.load_state_dict(copy.deepcopy(net_cpu.state_dict()))
    net_emu.train()
    x, n = torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn)
    lab = torch.tensor([3, 17, 39])
    cov0 = emu.kernel_coverage()
    loss = model_cls.cls_loss(net_emu(x, n), lab)
    loss.backward()
    cov1 = emu.kernel_coverage()
    ran = {k.split("<")[0].strip() for k, v in cov1.items() if v > cov0.get(k, 0)}
    assert {"gg_k_ctx_max", "gg_k_ctx_scatter", "gg_k_dz_segsum", "gg_k_small_build"} <= ran, sorted(ran)
    del torch.Tensor.is_cuda                  # the reference outside the emulation: plain CPU tensors, stock path
    try:
        loss_cpu = model_cls.cls_loss(net_cpu(x, n), lab)
        loss_cpu.backward()
    finally:
        torch.Tensor.is_cuda = property(lambda self: True)
    assert abs(float(loss.detach()) - float(loss_cpu.detach())) <= 2e-5 * max(1.0, abs(float(loss_cpu.detach())))
    ga = torch.cat([p.grad.reshape(-1) for p in net_cpu.parameters()]).double()
    gb = torch.cat([p.grad.reshape(-1) for p in net_emu.parameters()]).double()
    assert torch.isfinite(gb).all()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert 1.0 - cos < 2e-3, cos


def test_synth200k_net_training_step_against_the_cpu_model():
    """GGCNSynth (BASELINE configs[4]: four GridConv down layers, K = P = 64, widths 64 / 128 / 256 / 512) on a grid the
    emulator can afford -- 2 x 600 points, 8^3 / 4^3 / 2^3 / 1 voxels, O = 12 / 6 / 3 / 1 -- through model_synth.py's host code
    and the down layers' attention kernels at P = 64 (the materialised-pre-activation backward `gg_k_att_bwd_fused`, which
    the segmentation step of the test above only meets at P = 32) against the CPU model on the oracle's index operators"""
    from grid_gcn_amd import model_synth, synth
    from oracle.torch_index_ops import OracleIndexOps
    torch.manual_seed(0)
    grid = dict(num_points=600, coord_shift=[1.0, 1.0, 1.0], loc=1, down=[
        dict(voxel_size=[0.25] * 3, grid_size=[8] * 3, kernel_size=3, max_p_grid=64, max_o_grid=12),
        dict(voxel_size=[0.5] * 3, grid_size=[4] * 3, kernel_size=3, max_p_grid=64, max_o_grid=6),
        dict(voxel_size=[1.0] * 3, grid_size=[2] * 3, kernel_size=3, max_p_grid=64, max_o_grid=3),
        dict(voxel_size=[2.0] * 3, grid_size=[1] * 3, kernel_size=1, max_p_grid=64, max_o_grid=1)])
    cfg = dict(model_synth.SYNTH_200K, grid=grid)
    data, npn = synth.make_batch(2, 600, "planes")
    net_cpu = model_synth.GGCNSynth(cfg, index_ops=OracleIndexOps, fixed_seed=True).train()
    net_emu = model_synth.GGCNSynth(cfg, fixed_seed=True)
    net_emu.load_state_dict(copy.deepcopy(net_cpu.state_dict()))
    net_emu.train()
    x, n = torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn)
    lab = torch.tensor([5, 33])
    cov0 = emu.kernel_coverage()
    loss = model_synth.synth_loss(net_emu(x, n), lab)
    loss.backward()
    cov1 = emu.kernel_coverage()
    ran = {k.split("<")[0].strip() for k, v in cov1.items() if v > cov0.get(k, 0)}
    assert {"gg_k_att_bwd_fused", "gg_k_small_build", "gg_k_query_gridify"} <= ran, sorted(ran)
    del torch.Tensor.is_cuda
    try:
        loss_cpu = model_synth.synth_loss(net_cpu(x, n), lab)
        loss_cpu.backward()
    finally:
        torch.Tensor.is_cuda = property(lambda self: True)
    assert abs(float(loss.detach()) - float(loss_cpu.detach())) <= 2e-5 * max(1.0, abs(float(loss_cpu.detach())))
    ga = torch.cat([p.grad.reshape(-1) for p in net_cpu.parameters()]).double()
    gb = torch.cat([p.grad.reshape(-1) for p in net_emu.parameters()]).double()
    assert torch.isfinite(gb).all()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert 1.0 - cos < 2e-3, cos


def _tiny_seg_cfg(base, **over):
    """a segmentation config on a grid of 1024 points: down O = 128 / 32 / 8 (P = 32), up M = 32 / 128 / 1024 (P = 5) --
    the layer widths and branch options of `base` (model.SEG_8192 / SEG_81920) untouched"""
    grid = dict(num_points=1024, coord_shift=[1.0, 1.0, 1.0], loc=1,
                down=[dict(voxel_size=[0.125] * 3, grid_size=[16] * 3, kernel_size=3, max_p_grid=32, max_o_grid=128),
                      dict(voxel_size=[0.25] * 3, grid_size=[8] * 3, kernel_size=3, max_p_grid=32, max_o_grid=32),
                      dict(voxel_size=[0.5] * 3, grid_size=[4] * 3, kernel_size=3, max_p_grid=32, max_o_grid=8)],
                up=[dict(voxel_size=[0.5] * 3, grid_size=[4] * 3, kernel_size=3, max_p_grid=5, max_o_grid=32),
                    dict(voxel_size=[0.25] * 3, grid_size=[8] * 3, kernel_size=3, max_p_grid=5, max_o_grid=128),
                    dict(voxel_size=[0.125] * 3, grid_size=[16] * 3, kernel_size=3, max_p_grid=5, max_o_grid=1024)])
    return dict(base, grid=grid, dropout=0.0, **over)


SEG_VARIANTS = {
    # name: (base config, config overrides, index operators (emulated, oracle), mode, loss bar, 1 - cos bar)
    "seg81920_layers_train": ("SEG_81920", {}, ("HipIndexOps", "OracleIndexOps"), "train", 5e-6, 2e-3),
    "gridify_up_variant_train": ("SEG_8192", dict(up_neigh_fetch=False), ("HipIndexOps", "OracleIndexOps"), "train", 5e-6, 2e-3),
    "gridify_knn_variant_train": ("SEG_8192", {}, ("HipIndexOpsKNN", "OracleIndexOpsKNN"), "train", 5e-6, 2e-3),
    "seg81920_layers_eval": ("SEG_81920", {}, ("HipIndexOps", "OracleIndexOps"), "eval", 2e-4, None),
    "seg8192_layers_eval": ("SEG_8192", {}, ("HipIndexOps", "OracleIndexOps"), "eval", 2e-4, None),
    # (bf16 operands: the gradient keeps its DIRECTION, cos > 0.9 -- the GPU tier's own bar for this mode,
    #  test_bf16_mode_whole_model_loss_and_gradient_direction: a 0.4 % perturbation flips arg-maxima of the neighbour max)
    "seg81920_layers_bf16_train": ("SEG_81920", {}, ("HipIndexOps", "OracleIndexOps"), "bf16", 3e-2, 1e-1),
}


@slow
@pytest.mark.parametrize("name", list(SEG_VARIANTS))
def test_segmentation_variants_against_the_cpu_model(name):
    """the paths of GGCNSeg the fp32 training step above does not take, each against the CPU model on the oracle's index
    operators: the 81 920-point net's layer family (localfdim = 3, no ReLU behind the max: the source-side first conv,
    the Z2-free attention pair), the GridifyUp and GridifyKNN variants, the EVALUATION forward (BatchNorm folded, the
    fused inference kernels, train/evalpath.py), bf16 operands (stated tolerance: the emulator adds a bf16 MFMA's sixteen
    products in ascending k, the hardware's order is undocumented)"""
    from grid_gcn_amd import model, synth
    from grid_gcn_amd.train import common as tcommon
    import oracle.torch_index_ops as oix
    base, over, (hip_ix, orc_ix), mode, loss_bar, cos_bar = SEG_VARIANTS[name]
    torch.manual_seed(0)
    cfg = _tiny_seg_cfg(getattr(model, base), **over)
    data, npn = synth.make_batch(2, 1024, "planes")
    npn = npn.copy()
    npn[1, 0] = 900
    net_cpu = model.GGCNSeg(cfg, index_ops=getattr(oix, orc_ix), fixed_seed=True)
    net_emu = model.GGCNSeg(cfg, index_ops=getattr(model, hip_ix), fixed_seed=True)
    g = torch.Generator().manual_seed(3)
    for m in net_cpu.modules():
        if isinstance(m, torch.nn.BatchNorm1d):          # (evaluation: running statistics that are not the identity)
            m.weight.data.uniform_(0.5, 1.5, generator=g); m.bias.data.normal_(0, 0.3, generator=g)
            m.running_mean.normal_(0, 0.2, generator=g); m.running_var.uniform_(0.5, 1.5, generator=g)
    net_emu.load_state_dict(copy.deepcopy(net_cpu.state_dict()))
    x, n = torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn)
    lab = torch.randint(1, 21, (2, 1024))

    def reference(fn):
        del torch.Tensor.is_cuda
        try:
            return fn()
        finally:
            torch.Tensor.is_cuda = property(lambda self: True)

    if mode == "eval":
        net_cpu.eval(); net_emu.eval()
        with torch.no_grad():
            got = net_emu(x, n)
            want = reference(lambda: net_cpu(x, n))
        assert net_emu.last_tail_done == 2                  # the fused inference kernels ran, head included
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) <= loss_bar * scale
        return
    net_cpu.train(); net_emu.train()
    if mode == "bf16":
        tcommon.set_mlp_precision("bf16")
    try:
        loss = model.seg_loss(net_emu(x, n), lab)
        loss.backward()
    finally:
        tcommon.set_mlp_precision("fp32")
    assert net_emu.last_tail_done == 2                      # the training kernels ran, head included

    def ref_step():
        l = model.seg_loss(net_cpu(x, n), lab)
        l.backward()
        return l
    loss_cpu = reference(ref_step)
    assert abs(float(loss.detach()) - float(loss_cpu.detach())) <= loss_bar * max(1.0, abs(float(loss_cpu.detach())))
    ga = torch.cat([p.grad.reshape(-1) for p in net_cpu.parameters()]).double()
    gb = torch.cat([p.grad.reshape(-1) for p in net_emu.parameters()]).double()
    assert torch.isfinite(gb).all()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert 1.0 - cos < cos_bar, cos


def test_classifier_evaluation_forward_against_the_cpu_model():
    """GGCNCls in evaluation mode (running statistics folded, train/evalpath.py: edge_block_cls_eval, the FC head through
    the plain GEMM) at the reduced grid of the training-step test, against the CPU model on the oracle's index operators"""
    from grid_gcn_amd import model_cls, synth
    from oracle.torch_index_ops import OracleIndexOps
    torch.manual_seed(1)
    cfg = _tiny_cls_cfg()
    data, npn = synth.make_batch(3, 160, "ball")
    net_cpu = model_cls.GGCNCls(cfg, index_ops=OracleIndexOps).eval()
    g = torch.Generator().manual_seed(5)
    for m in net_cpu.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5, generator=g); m.bias.data.normal_(0, 0.3, generator=g)
            m.running_mean.normal_(0, 0.2, generator=g); m.running_var.uniform_(0.5, 1.5, generator=g)
    net_emu = model_cls.GGCNCls(cfg).eval()
    net_emu.load_state_dict(copy.deepcopy(net_cpu.state_dict()))
    x, n = torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn)
    cov0 = emu.kernel_coverage()
    with torch.no_grad():
        got = net_emu(x, n)
        cov1 = emu.kernel_coverage()
        del torch.Tensor.is_cuda
        try:
            want = net_cpu(x, n)
        finally:
            torch.Tensor.is_cuda = property(lambda self: True)
    ran = {k.split("<")[0].strip() for k, v in cov1.items() if v > cov0.get(k, 0)}
    assert "gg_k_ctx_max" in ran, sorted(ran)              # (the classification edge block's own kernels ran)
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= 2e-4 * scale, float((got - want).abs().max()) / scale


@slow
def test_smoke_entry_point_body(monkeypatch):
    """__graft_entry__.smoke() as the driver runs it at round end -- Gridify and BallKNN against the oracle, one training
    step of the segmentation net against the CPU model -- with the emulator in place of cuda:0 (270 s)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("graft_entry_emu", os.path.join(root, "__graft_entry__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    _to_cpu(mod.smoke)()


@slow
def test_bench_main_on_the_emulator():
    """bench.py's main() -- the program the driver runs at round end -- executed with the emulator in place of cuda:0 at
    1 x 1024 points (tools/simt_bench.py): the line it prints carries the contract's fields"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import simt_bench
    line = simt_bench.run(["--config", "cfg3", "--batch", "1", "--points", "1024", "--no-micro"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_step", "ms_per_cagq_layer"):
        assert k in line, k
    assert line["step_mode"] == "eager" and line["value"] > 0 and line["config"]["points_per_cloud"] == 1024
