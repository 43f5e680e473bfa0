"""GPU: the hand-written GridConv kernels against the independent float64 fixtures
(tests/golden/gridconv_*.npz, oracle/gridconv_ref.py; see tests/test_gridconv_golden.py for how
they are made and for the bars).

  eval-mode BatchNorm  : |HIP - fixture| <= 1e-5 * max(1, max|x|)            (north_star bar)
  train-mode BatchNorm : |HIP - fixture| <= 1e-5 * max(1, max|x|)            (north_star bar; measured
                         3.5e-7 .. 1.3e-6, profiles/r4_float_parity.txt) and not worse than 2 x the stock fp32 ops
  gradients            : per tensor, |HIP - fp64| <= 1e-5 * max|g| (measured <= 3.5e-6) and
                         <= max(3 * |stock fp32 - fp64|, 1e-5 * max|g|),
                         fp64 = the stock modules in float64 on the CPU (their forward is pinned to
                         the restatement at 1e-9 by the CPU tests)
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import gridconv_cases as gc  # noqa: E402
from grid_gcn_amd import ops  # noqa: E402

DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def report(line):
    """the measured distances, for profiles/ (GG_PARITY_REPORT=<file>; VERDICT r3 item 7: how far from
    north_star's 1e-5 the train-mode path actually is -- a number, not a bound)"""
    path = os.environ.get("GG_PARITY_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def run_hip(case, m, train):
    cent, src, idx = T(case["cent"]), T(case["src"]), T(case["nebidx"])
    cm = None if case["centmsk"] is None else T(case["centmsk"])
    cof = None if case["center_ori_feats"] is None else T(case["center_ori_feats"])
    m.train(train)
    if case["kind"] == "seg":
        if train:
            return m.forward_src(cent, src, idx, cm, center_ori_feats=cof)
        with torch.no_grad():
            return m.forward_fused(cent, src, idx, cm, center_ori_feats=cof)
    # classification block: the kernels of tcls._EdgeBlockClsTrain / edge_block_cls_eval
    with (torch.enable_grad() if train else torch.no_grad()):
        out = m.forward_src(cent, src, idx, cm)
    assert out is not None, "classification edge block fell back to the stock modules"
    return out


def run_stock_gpu(case, m, train):
    """stock PyTorch-ROCm ops on the same device (no hand-written GridConv kernel)."""
    cent, src, idx = T(case["cent"]), T(case["src"]), T(case["nebidx"])
    cm = None if case["centmsk"] is None else T(case["centmsk"])
    cof = None if case["center_ori_feats"] is None else T(case["center_ori_feats"])
    m.train(train)
    old = m.mfma_train
    m.mfma_train = False
    try:
        B, N, C = src.shape
        flat = (idx.long() + (torch.arange(B, device=DEV) * N).view(B, 1, 1)).clamp(0, B * N - 1)
        nb = src.reshape(B * N, C)[flat]
        if case["kind"] == "seg":
            return m(cent[..., 0:3], nb, cm, center_ori_feats=cof)
        return m(cent[..., 0:3], nb, cm)
    finally:
        m.mfma_train = old


@pytest.mark.parametrize("name", list(gc.CASES))
def test_hip_eval_matches_fixture(name):
    case = gc.CASES[name]()
    g = np.load(os.path.join(GOLD, name + ".npz"))
    m = gc.build_module(case).to(DEV)
    out = run_hip(case, m, False).cpu().numpy()[:, case["rows"], :]
    err = np.abs(out.astype(np.float64) - g["eval"]).max()
    report("eval   %-18s |HIP - fp64| = %.3e = %.3e * max(1, max|x| = %.3g)   (bar 1e-5)" % (
        name, err, err / max(1.0, float(g["eval_absmax"])), float(g["eval_absmax"])))
    assert err <= 1e-5 * max(1.0, float(g["eval_absmax"])), (name, err)


@pytest.mark.parametrize("name", list(gc.CASES))
def test_hip_train_not_worse_than_stock_fp32(name):
    case = gc.CASES[name]()
    g = np.load(os.path.join(GOLD, name + ".npz"))
    want, scale = g["train"].astype(np.float64), max(1.0, float(g["train_absmax"]))
    m = gc.build_module(case).to(DEV)
    with torch.no_grad():
        stock = run_stock_gpu(case, m, True).cpu().numpy()[:, case["rows"], :]
    m = gc.build_module(case).to(DEV)           # fresh moving statistics
    hip = run_hip(case, m, True).detach().cpu().numpy()[:, case["rows"], :]
    e_stock = np.abs(stock - want).max()
    e_hip = np.abs(hip - want).max()
    report("train  %-18s |HIP - fp64| = %.3e * scale, |stock fp32 - fp64| = %.3e * scale, ratio %.2f   "
           "(scale = max(1, max|x|) = %.3g; bar: <= max(2 x stock, 1e-5))" % (
               name, e_hip / scale, e_stock / scale, e_hip / max(e_stock, 1e-30), scale))
    assert e_stock <= 1e-4 * scale, (name, e_stock)
    assert e_hip <= max(2.0 * e_stock, 1e-5 * scale), (name, e_hip, e_stock)
    # round 4: the measured distances (profiles/r4_float_parity.txt: 3.5e-7 .. 1.3e-6 of the scale) sit inside
    # north_star's bar itself, so the bar is asserted as it stands -- no reference to the stock ops needed
    assert e_hip <= 1e-5 * scale, (name, e_hip, scale)


@pytest.mark.parametrize("name", ["gridconv_seg_L1", "gridconv_up2", "gridconv_cls_L0"])
def test_hip_gradients_bounded_by_stock_fp32(name):
    case = gc.CASES[name]()
    seg = case["kind"] == "seg"
    kw = (lambda cof: dict(center_ori_feats=cof)) if seg else (lambda cof: {})
    rng = np.random.default_rng(7)
    B, O = case["nebidx"].shape[0], case["nebidx"].shape[1]

    def grads(m, fwd, dtype, dev):
        m.train(True)
        src = torch.from_numpy(case["src"]).to(dev).to(dtype).requires_grad_(True)
        out = fwd(m, src)
        G = torch.from_numpy(rng_cot[:, :, :out.shape[2]]).to(dev).to(dtype)
        (out * G).sum().backward()
        gs = {}
        if src.shape[2] > 4:
            gs["src"] = src.grad[..., 4:].detach().double().cpu().numpy()
        for n_, p in m.named_parameters():
            if p.grad is not None:
                gs[n_] = p.grad.detach().double().cpu().numpy()
        return gs

    m64 = gc.build_module_f64(case)
    with torch.no_grad():
        c_out = m64.out_channels
    rng_cot = rng.normal(0, 1, (B, O, c_out))

    def fwd_cpu(m, src):
        from oracle.torch_index_ops import OracleIndexOps
        nb = OracleIndexOps.batch_take_g(src, torch.from_numpy(case["nebidx"]))
        cent = torch.from_numpy(case["cent"]).to(src.dtype)
        cm = None if case["centmsk"] is None else torch.from_numpy(case["centmsk"]).to(src.dtype)
        cof = None if case["center_ori_feats"] is None else \
            torch.from_numpy(case["center_ori_feats"]).to(src.dtype)
        return m(cent[..., 0:3], nb, cm, **kw(cof))

    def fwd_stock_gpu(m, src):
        cent, idx = T(case["cent"]), T(case["nebidx"])
        cm = None if case["centmsk"] is None else T(case["centmsk"])
        cof = None if case["center_ori_feats"] is None else T(case["center_ori_feats"])
        m.mfma_train = False
        Bn, N, C = src.shape
        flat = (idx.long() + (torch.arange(Bn, device=DEV) * N).view(Bn, 1, 1)).clamp(0, Bn * N - 1)
        return m(cent[..., 0:3], src.reshape(Bn * N, C)[flat], cm, **kw(cof))

    def fwd_hip(m, src):
        cent, idx = T(case["cent"]), T(case["nebidx"])
        cm = None if case["centmsk"] is None else T(case["centmsk"])
        cof = None if case["center_ori_feats"] is None else T(case["center_ori_feats"])
        out = m.forward_src(cent, src, idx, cm, **kw(cof))
        assert out is not None
        return out

    g64 = grads(m64, fwd_cpu, torch.float64, "cpu")
    g_stock = grads(gc.build_module(case).to(DEV), fwd_stock_gpu, torch.float32, DEV)
    g_hip = grads(gc.build_module(case).to(DEV), fwd_hip, torch.float32, DEV)
    assert set(g_hip) == set(g64)
    for k in g64:
        scale = max(np.abs(g64[k]).max(), 1e-30)
        e_stock = np.abs(g_stock[k] - g64[k]).max()
        e_hip = np.abs(g_hip[k] - g64[k]).max()
        report("grad   %-18s %-28s |HIP - fp64| = %.3e * max|g|, |stock - fp64| = %.3e * max|g|, ratio %.2f" % (
            name, k, e_hip / scale, e_stock / scale, e_hip / max(e_stock, 1e-30)))
        assert e_hip <= max(3.0 * e_stock, 1e-5 * scale), (name, k, e_hip, e_stock, scale)
        # round 4: every gradient tensor within 1e-5 of its own scale (measured: <= 3.5e-6).  The conv biases
        # in front of a BatchNorm are analytically zero (fp64 leaves ~1e-17 of round-off, the kernels return 0)
        if not k.endswith("lin.bias"):
            assert e_hip <= 1e-5 * scale, (name, k, e_hip, scale)
        else:
            assert np.abs(g_hip[k]).max() == 0.0 and scale < 1e-9, (name, k, scale)
