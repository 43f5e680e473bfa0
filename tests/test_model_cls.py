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
    # identical neighbour sets (bit-exact index ops) -> only fp32 GEMM order differs
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-3, atol=2e-3)
