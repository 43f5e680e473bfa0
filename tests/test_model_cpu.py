"""Host logic on CPU: the model wiring (with the oracle as index provider) and the
data-parallel gradient exchange over gloo with world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from grid_gcn_amd import dp, model, synth
from oracle.torch_index_ops import OracleIndexOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(B, N, first_id=0):
    data, npn = synth.make_batch(B, N, "planes", first_id=first_id)
    return torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn)


@pytest.mark.parametrize("cfg", [model.SEG_8192, model.SEG_81920], ids=["seg8192", "seg81920"])
def test_seg_model_shapes_and_grads(cfg):
    torch.manual_seed(0)
    net = model.GGCNSeg(cfg, index_ops=OracleIndexOps)
    nparam = sum(p.numel() for p in net.parameters())
    assert 300_000 < nparam < 400_000          # ~0.33 M parameters
    x, n = _inputs(2, 1024)
    out = net(x, n)
    assert out.shape == (2, 1024, 21)
    lab = torch.randint(0, 21, (2, 1024))
    model.seg_loss(out, lab).backward()
    for name, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name


def test_seg_model_gridify_up_variant():
    """up_neigh_fetch: False -> GridifyUp instead of BallKNN (ggcn_models_g.py:206-210)."""
    cfg = dict(model.SEG_8192, up_neigh_fetch=False)
    torch.manual_seed(0)
    net = model.GGCNSeg(cfg, index_ops=OracleIndexOps)
    x, n = _inputs(1, 8192)                    # last up layer: max_o_grid == N == 8192
    out = net(x, n)
    assert out.shape == (1, 8192, 21) and torch.isfinite(out).all()


def test_loss_ignores_label_zero():
    logits = torch.randn(2, 5, 21)
    lab = torch.tensor([[0, 1, 2, 0, 3], [0, 0, 0, 0, 4]])
    want = torch.nn.functional.cross_entropy(logits.reshape(-1, 21)[lab.reshape(-1) != 0],
                                             lab.reshape(-1)[lab.reshape(-1) != 0])
    assert torch.allclose(model.seg_loss(logits, lab), want)


def _dp_case(rank):
    g = torch.Generator().manual_seed(1000 + rank)
    cxyz = torch.rand(1, 40, 3, generator=g)
    nbr = torch.rand(1, 40, 8, 4 + 16, generator=g)
    cmask = (torch.rand(1, 40, generator=g) > 0.2).float()
    cori = torch.rand(1, 40, 12, generator=g)
    return cxyz, nbr, cmask, cori


def _dp_layer(seed):
    from grid_gcn_amd.gridconv import SubGUpdate
    torch.manual_seed(seed)
    net = SubGUpdate(16, [32, 32], localfdim=3, relu=True, center_in=12, center_dim=[16],
                     out_dim=[24])
    net.eval()                                 # BN on running stats: grads are batch-additive
    return net


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _dp_layer(100 + rank)                # deliberately different init per rank
    sync = dp.FlatGradAllReduce(net)
    sync.broadcast_parameters(0)
    cxyz, nbr, cmask, cori = _dp_case(rank)    # each rank owns one cloud (batch shard)
    net(cxyz, nbr, cmask, cori).square().sum().backward()
    sync()
    g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    if rank == 0:
        q.put(g.numpy())
    dist.destroy_process_group()


def test_dp_gloo_world2_matches_full_batch():
    """average of the per-rank gradients == (full-batch gradient) / world, on a GridConv layer.
    (The index ops are not part of this identity in the reference either: cuRAND seeds depend on
    the cloud's position b in the batch, gridify.cu:126,149, and BallKNN's -1 gathers row b*Nd-1,
    utils/ops.py:89-92, so a cloud's result depends on which shard it sits in.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g_dp = q.get(timeout=300)
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    net = _dp_layer(100)
    cs = [_dp_case(r) for r in range(2)]
    args = [torch.cat([c[j] for c in cs]) for j in range(4)]
    net(*args).square().sum().backward()
    g_full = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy() / 2
    np.testing.assert_allclose(g_dp, g_full, rtol=1e-4, atol=1e-6)


def test_weighted_gradient_scales_rows_by_label_weight():
    """custom_op/weighted_gradient.py: identity forward, gradient row * weight[label]."""
    from grid_gcn_amd import model
    torch.manual_seed(0)
    lg1 = torch.randn(50, 21, requires_grad=True)
    lg2 = lg1.detach().clone().requires_grad_(True)
    lab = torch.randint(0, 21, (50,))
    w = torch.rand(21) + 0.5
    l1 = model.seg_loss(lg1, lab)
    l2 = model.seg_loss(lg2, lab, weights=w.tolist())
    assert float(l1) == float(l2)
    l1.backward()
    l2.backward()
    exp = lg1.grad * w[lab][:, None]
    assert torch.allclose(lg2.grad, exp, rtol=1e-6, atol=1e-9)
    assert float(lg2.grad[lab == 0].abs().max()) == 0.0


def test_bn_decay_schedule_and_setter():
    from grid_gcn_amd import model
    # base_solver.py:74 with the shipped configs.yaml values (bn_decay .9, factor .5, clip .99)
    assert abs(model.bn_decay_at(1) - 0.55) < 1e-12
    assert abs(model.bn_decay_at(2) - 0.775) < 1e-12
    assert model.bn_decay_at(20) == 0.99
    net = model.GGCNSeg(model.SEG_8192, index_ops=OracleIndexOps)
    model.set_bn_decay(net, 0.775)
    moms = {m.momentum for m in net.modules() if isinstance(m, torch.nn.BatchNorm1d)}
    assert len(moms) == 1 and abs(moms.pop() - 0.225) < 1e-12


# ---- bench.py's step() wiring under data parallelism (2 processes, gloo) ------------------------
TINY_GRID = dict(
    num_points=512, coord_shift=[1.0, 1.0, 1.0], loc=1,
    down=[dict(voxel_size=[0.2] * 3, grid_size=[10] * 3, kernel_size=3, max_p_grid=8, max_o_grid=64),
          dict(voxel_size=[0.4] * 3, grid_size=[5] * 3, kernel_size=3, max_p_grid=8, max_o_grid=16),
          dict(voxel_size=[1.0] * 3, grid_size=[2] * 3, kernel_size=3, max_p_grid=8, max_o_grid=4)],
    up=[dict(voxel_size=[1.0] * 3, grid_size=[2] * 3, kernel_size=3, max_p_grid=3, max_o_grid=16),
        dict(voxel_size=[0.4] * 3, grid_size=[5] * 3, kernel_size=3, max_p_grid=3, max_o_grid=64),
        dict(voxel_size=[0.2] * 3, grid_size=[10] * 3, kernel_size=3, max_p_grid=3, max_o_grid=512)])
TINY_SEG = dict(grid=TINY_GRID, inputDim=[0, 16, 32], pt_ele_dim=[[8, 16], [16, 32], [32, 64]],
                localfdim=3, relu=False, up_inputDim=[64, 32, 32], up_center_dim=[[16]] * 3,
                up_pt_ele_dim=[[32]] * 3, up_gcn_outDim=[[32]] * 3, up_neigh_fetch=True,
                num_classes=5, bn_decay=0.9, dropout=0.0)


def _tiny_shard(rank, S=2, N=512):
    from grid_gcn_amd import synth
    data, npn = synth.make_batch(S, N, "planes", first_id=300 + rank * S)
    g = torch.Generator().manual_seed(900 + rank)
    lab = torch.randint(0, 5, (S, N), generator=g)
    return torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn), lab


def _tiny_net(seed):
    from grid_gcn_amd import model
    torch.manual_seed(seed)
    return model.GGCNSeg(TINY_SEG, index_ops=OracleIndexOps, seed=0, fixed_seed=True).train()


def _step_worker(rank, world, port, q, nsteps):
    from grid_gcn_amd import model
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _tiny_net(200 + rank)                      # different init per rank: broadcast must fix it
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    sync = dp.FlatGradAllReduce(net)
    sync.broadcast_parameters()
    x, n, lab = _tiny_shard(rank)

    def step():                                       # == bench.py's step()
        opt.zero_grad(set_to_none=True)
        loss = model.seg_loss(net(x, n), lab)
        loss.backward()
        sync()
        opt.step()
        return loss

    g1 = None
    for s in range(nsteps):
        step()
        if s == 0:
            g1 = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
        # gradients have become views of the flat bucket (no scatter-back copies)
        base = sync.flat.untyped_storage().data_ptr()
        assert all(p.grad.untyped_storage().data_ptr() == base for p in net.parameters())
    q.put((rank, g1.numpy(), torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()))
    dist.destroy_process_group()


def test_dp_step_wiring_gloo_world2():
    """bench.py's step() -- model + FlatGradAllReduce (gradients as views of the flat bucket) +
    Adam + zero_grad(set_to_none=True) -- on two gloo ranks, each owning a 2-cloud shard:
    (1) the first step's synchronised gradient == mean of the shard gradients computed in ONE process,
    (2) after 3 steps the parameters equal those of the single process applying Adam to the averaged
        gradients, and (3) they are bitwise identical on both ranks."""
    from grid_gcn_amd import model
    nsteps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 1000
    procs = [ctx.Process(target=_step_worker, args=(r, 2, port, q, nsteps)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(2):
        r, g1, par = q.get(timeout=600)
        res[r] = (g1, par)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert res[0][1].tobytes() == res[1][1].tobytes(), "parameters diverged across ranks"
    assert res[0][0].tobytes() == res[1][0].tobytes(), "synchronised gradients differ across ranks"
    # single-process emulation: BatchNorm statistics per shard (per device in the reference, too)
    net = _tiny_net(200)                              # rank 0's weights == what the broadcast gives
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    shards = [_tiny_shard(r) for r in range(2)]
    params = list(net.parameters())
    g_first = None
    for s in range(nsteps):
        acc = [torch.zeros_like(p) for p in params]
        for x, n, lab in shards:
            opt.zero_grad(set_to_none=True)
            model.seg_loss(net(x, n), lab).backward()
            for a, p in zip(acc, params):
                if p.grad is not None:
                    a += p.grad
        for a, p in zip(acc, params):
            p.grad = a / 2
        if s == 0:
            g_first = torch.cat([p.grad.reshape(-1) for p in params]).numpy()
        opt.step()
    gmax = float(np.abs(g_first).max())
    np.testing.assert_allclose(res[0][0], g_first, rtol=1e-4, atol=2e-6 * gmax)
    want = torch.cat([p.detach().reshape(-1) for p in params]).numpy()
    # Adam divides by sqrt(v): where the gradient is far from zero the update is well conditioned and
    # the parameters must agree closely; near-zero gradients (fp32 summation-order noise decides
    # their sign) may differ by up to one full update lr per step
    stable = np.abs(g_first) > 1e-3 * gmax
    np.testing.assert_allclose(res[0][1][stable], want[stable], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(res[0][1], want, rtol=0, atol=2.1e-3 * nsteps)


def _agree_worker(rank, world, port, q, values):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = []
    for what, vals in values:
        try:
            bench.ranks_agree(what, vals[rank], world, torch.device("cpu"))
            out.append("ok")
        except RuntimeError as e:
            out.append("raised:" + str(e)[:40])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_refuses_a_job_whose_ranks_disagree_on_the_step_mode():
    """bench.ranks_agree (round 5): with N > 1 every rank must time the same kind of step; a rank
    that fell back to the eager step (or whose RCCL capture probe said something else) makes ALL ranks raise instead
    of producing a line that would read as a scaling loss.  Two gloo ranks: equal values pass, unequal ones raise on
    both ranks, None == None passes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 1000
    values = [("step_mode", ("hipgraph", "hipgraph")), ("step_mode", ("hipgraph", "eager")),
              ("rccl_capture_probe", (None, None)), ("rccl_capture_probe", ("ok", "failed"))]
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q, values)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r][0] == "ok" and res[r][2] == "ok", res
        assert res[r][1].startswith("raised:") and res[r][3].startswith("raised:"), res
