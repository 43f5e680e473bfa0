"""Classification network (BASELINE configs[1]/[2]): CPU wiring test with the oracle as index
provider, GPU test that the HIP index path gives the same logits as the oracle path."""
import numpy as np
import pytest
import torch

from grid_gcn_amd import model_cls, synth
from oracle.torch_index_ops import OracleIndexOps


def _inputs(B, N):
    data, npn = synth.make_batch(B, N, "ball")
    return torch.from_numpy(data[..., :3].copy()), torch.from_numpy(npn)


def test_cls_model_shapes_params_grads():
    torch.manual_seed(0)
    net = model_cls.GGCNCls(index_ops=OracleIndexOps)
    nparam = sum(p.numel() for p in net.parameters())
    assert 1_600_000 < nparam < 2_000_000         # ~1.77 M parameters
    # per-edge MACs of layer 0 (54 080)
    l0 = net.layers[0]
    macs = sum(m.lin.in_features * m.lin.out_features
               for seq in (l0.pt_mlp, l0.att1, l0.att2) for m in seq)
    assert macs == 54080
    x, n = _inputs(2, 1024)
    out = net(x, n)
    assert out.shape == (2, 40)
    model_cls.cls_loss(out, torch.tensor([3, 7])).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.gpu
def test_cls_model_hip_index_ops_match_oracle_path():
    torch.manual_seed(0)
    net = model_cls.GGCNCls(index_ops=OracleIndexOps).eval()
    x, n = _inputs(4, 1024)
    with torch.no_grad():
        want = net(x, n)
        net_gpu = model_cls.GGCNCls().eval()
        net_gpu.load_state_dict(net.state_dict())
        net_gpu = net_gpu.to("cuda:0")
        got = net_gpu(x.to("cuda:0"), n.to("cuda:0")).cpu()
    # identical neighbour sets (bit-exact index ops) -> only fp32 GEMM order differs: the distance is MEASURED
    # (GG_PARITY_REPORT -> profiles/r6_float_parity.txt) and the bar is 3 x that, relative to max(1, max|logit|)
    scale = max(1.0, float(want.abs().max()))
    dist = float((got - want).abs().max()) / scale
    from conftest import parity_report
    parity_report("model cls eval (4 x 1024) HIP path vs CPU model on oracle index ops: max|dlogit| = %.3e * "
                  "max(1, max|logit| = %.3g)" % (dist, float(want.abs().max())))
    assert dist <= CLS_EVAL_BAR, dist


# Measured on the EMULATOR (no GPU in round 6; profiles/r6_float_parity.txt): 2.8e-9 relative to max(1, max|logit|) at cfg2's
# batch of 32 -- the evaluation forward has no atomics and the emulated fp32 MFMA is the hardware's k-ordered chain, so the
# GPU's figure is expected to be the same to within a few ulp.  The bar is that x 350 until a GPU session prints its own
# figure into GG_PARITY_REPORT (then: 3 x it).  It was 2e-3 with no measurement behind it.
CLS_EVAL_BAR = 1e-6


@pytest.mark.gpu
def test_cls_cfg2_batch32_gridify_bit_exact_and_eval_logits():
    """BASELINE configs[1] at its REAL batch: 32 clouds x 1024 points (VERDICT r5 10a -- the index kernels are
    parameterised by B: one workgroup per cloud in gg_k_small_build, XCD placement b mod 8).  The three Gridify
    layers of the classifier, chained as the model chains them, byte for byte against the C oracle; then the
    evaluation logits of the whole net against the CPU model on the oracle's index ops."""
    from grid_gcn_amd import ops
    from oracle import oracle as orc
    cfg = synth.CLS_MODELNET40
    data, npn = synth.make_batch(32, 1024, "ball")
    npn = npn.copy()
    npn[5, 0], npn[17, 0], npn[31, 0] = 1000, 513, 1          # ragged clouds inside the batch
    d, n = data, npn
    for l in range(3):
        kw = synth.gridify_kwargs(cfg, l, seed=3 + l)
        want = orc.gridify(d, n, **kw)
        got = [t.cpu().numpy() for t in ops.Gridify(torch.from_numpy(d).to("cuda:0"),
                                                     torch.from_numpy(n).to("cuda:0"), **kw)]
        for j, (g, w) in enumerate(zip(got, want)):
            assert g.tobytes() == w.tobytes(), "layer %d output %d" % (l, j)
        d, n = want[2], want[4]
    torch.manual_seed(1)
    net = model_cls.GGCNCls(index_ops=OracleIndexOps).eval()
    x, nn_ = _inputs(32, 1024)
    with torch.no_grad():
        want = net(x, nn_)
        net_gpu = model_cls.GGCNCls().eval()
        net_gpu.load_state_dict(net.state_dict())
        net_gpu = net_gpu.to("cuda:0")
        got = net_gpu(x.to("cuda:0"), nn_.to("cuda:0")).cpu()
    scale = max(1.0, float(want.abs().max()))
    dist = float((got - want).abs().max()) / scale
    from conftest import parity_report
    parity_report("model cls eval (32 x 1024, cfg2's batch) HIP path vs CPU model on oracle index ops: "
                  "max|dlogit| = %.3e * max(1, max|logit| = %.3g)" % (dist, float(want.abs().max())))
    assert dist <= CLS_EVAL_BAR, dist


@pytest.mark.gpu
def test_cls_cfg2_batch32_training_step_matches_stock_modules():
    """one fwd + bwd of GGCNCls at cfg2's batch of 32 (the size bench.py --config cfg2 times): the hand-written
    training kernels against the stock PyTorch modules on the same HIP index outputs."""
    torch.manual_seed(0)
    cfg = dict(model_cls.CLS_MN40, dropout=0.0)
    net = model_cls.GGCNCls(cfg, fixed_seed=True).to("cuda:0").train()
    x, n = _inputs(32, 1024)
    x, n = x.to("cuda:0"), n.to("cuda:0")
    lab = torch.randint(0, 40, (32,), device="cuda:0")
    state = {k: v.clone() for k, v in net.state_dict().items()}
    res = []
    for mfma in (True, False):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        for l in net.layers:
            l.mfma_train = mfma
        loss = model_cls.cls_loss(net(x, n), lab)
        loss.backward()
        res.append((float(loss), torch.cat([p.grad.reshape(-1) for p in net.parameters()])))
    dl = abs(res[0][0] - res[1][0]) / max(1.0, abs(res[1][0]))
    a, b = res[0][1].double(), res[1][1].double()
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    rel = float((a - b).norm() / b.norm())
    from conftest import parity_report
    parity_report("model cls (32 x 1024, cfg2's batch) HIP kernels vs stock fp32 modules: |dloss|/loss %.3e  "
                  "1-cos(grad) %.3e  rel-L2(grad) %.3e" % (dl, 1.0 - cos, rel))
    assert torch.isfinite(res[0][1]).all()
    # NOT YET MEASURED at this batch (the GPU pool was closed when the test was written): sanity bars an order of
    # magnitude above the 8-cloud test's measured distances (1.3e-7 / 1.2e-5 / 4.9e-3; this net is discretely
    # sensitive at fp32 round-off: see there); the measured values go to GG_PARITY_REPORT
    assert dl < 2e-5, dl
    assert 1.0 - cos < 4e-4, cos
    assert rel < 5e-2, rel


@pytest.mark.gpu
def test_cls_training_step_mfma_kernels_match_stock_modules():
    """classification net, one fwd+bwd on the GPU: conv+BN+ReLU stacks through the hand-written
    training kernels (where their width limits allow) vs the stock PyTorch modules."""
    torch.manual_seed(0)
    cfg = dict(model_cls.CLS_MN40, dropout=0.0)
    net = model_cls.GGCNCls(cfg, fixed_seed=True).to("cuda:0").train()
    x, n = _inputs(8, 1024)
    x, n = x.to("cuda:0"), n.to("cuda:0")
    lab = torch.randint(0, 40, (8,), device="cuda:0")
    state = {k: v.clone() for k, v in net.state_dict().items()}
    res = []
    for mfma in (True, False):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        for l in net.layers:
            l.mfma_train = mfma
        loss = model_cls.cls_loss(net(x, n), lab)
        loss.backward()
        res.append((float(loss), torch.cat([p.grad.reshape(-1) for p in net.parameters()])))
    assert abs(res[0][0] - res[1][0]) < 2e-6 * max(1.0, abs(res[1][0]))    # measured 1.3e-7
    # this net is discretely sensitive at fp32 round-off (max-pool arg-max over 128 padded
    # neighbours, BatchNorm over 8 rows in the head): perturbing the INPUT of the stock path by 1e-7
    # already moves single weight gradients by ~1e-2 of their scale, so the
    # two paths are compared as vectors
    a, b = res[0][1].double(), res[1][1].double()
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    rel = float((a - b).norm() / b.norm())
    from conftest import parity_report
    parity_report("model cls (8 x 1024) HIP kernels vs stock fp32 modules: |dloss|/loss %.3e  1-cos(grad) %.3e  "
                  "rel-L2(grad) %.3e" % (abs(res[0][0] - res[1][0]) / max(1.0, abs(res[1][0])), 1.0 - cos, rel))
    # measured 1 - cos 1.2e-5, rel-L2 4.9e-3 (profiles/r5_float_parity.txt); bars = 3x
    assert 1.0 - cos < 3.6e-5, cos
    assert rel < 1.5e-2, rel


@pytest.mark.gpu
@pytest.mark.parametrize("cin,pt,att,O,P", [(0, [64, 64, 128], [64, 128, 128], 40, 64),
                                            (128, [128, 128, 256], [128, 256, 256], 12, 64),
                                            (32, [32, 64], [32, 64, 64], 9, 32)])
def test_cls_edge_block_kernels_match_stock_modules(cin, pt, att, O, P):
    """classification GridConv edge block (pt-MLP, att1, att2 on concat(att1 | pt-MLP | context),
    product, max) on the hand-written kernels -- two-source first attention conv, context as a
    per-centre bias -- against the stock modules on the gathered / concatenated tensors:
    forward, every parameter gradient, the source gradient, the running statistics."""
    import copy
    from grid_gcn_amd import ops
    from grid_gcn_amd.train import cls as tcls
    DEV = "cuda:0"
    torch.manual_seed(cin + O)
    gen = torch.Generator().manual_seed(cin + P)
    B, Nsrc = 3, 150
    ref = model_cls.SubGUpdateCls(cin, pt, att).to(DEV).train()
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    new = copy.deepcopy(ref)
    ref.mfma_train = False
    src = torch.rand(B, Nsrc, 4 + cin, generator=gen) * 2 - 1
    src[..., 4:].clamp_(min=0)                      # features behind a ReLU: exact zeros, ties
    src1 = src.to(DEV).requires_grad_(cin > 0)
    src2 = src.to(DEV).requires_grad_(cin > 0)
    nebidx = torch.randint(0, Nsrc, (B, O, P), generator=gen, dtype=torch.int32).to(DEV)
    cent = (torch.rand(B, O, 4, generator=gen) * 2 - 1).to(DEV)
    msk = (torch.rand(B, O, generator=gen) > 0.2).float().to(DEV)
    assert tcls.edge_block_cls_supported(list(new.pt_mlp), list(new.att1), list(new.att2),
                                              src2, P)
    y1 = ref(cent[..., 0:3], ops.batch_take_g(src1, nebidx), msk)
    y2 = new.forward_src(cent, src2, nebidx, msk)
    assert y2 is not None and y1.shape == y2.shape
    assert float((y1 - y2).abs().max()) <= 3e-5 * max(1.0, float(y1.abs().max()))
    g = torch.randn(y1.shape, generator=gen).to(DEV)
    y1.backward(g)
    y2.backward(g)

    def close(a, b, tol=5e-4):
        s = max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol * s, (float((a - b).abs().max()), s)
    if cin:
        close(src2.grad[..., 4:], src1.grad[..., 4:])
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), new.named_parameters()):
        if n1.endswith("lin.bias"):
            continue                      # bias in front of a BatchNorm: exact 0 vs round-off noise
        close(p2.grad, p1.grad)
    for (n1, b1), (n2, b2) in zip(ref.named_buffers(), new.named_buffers()):
        if "num_batches" not in n1:
            close(b2, b1, 1e-5)
        else:
            assert int(b1) == int(b2)
    # evaluation (running statistics) through the same forward kernels
    ref.eval(), new.eval()
    with torch.no_grad():
        e1 = ref(cent[..., 0:3], ops.batch_take_g(src1, nebidx), msk)
        e2 = new.forward_src(cent, src2, nebidx, msk)
    assert e2 is not None
    assert float((e1 - e2).abs().max()) <= 3e-5 * max(1.0, float(e1.abs().max()))
