"""Float path pinned by an independent restatement.

tests/golden/gridconv_*.npz hold GridConv layer outputs computed in float64 by
oracle/gridconv_ref.py -- numpy, the reference's NCHW layout, written from
segmentation/models/gcn_module_g_att.py:120-287, classification/models/gcn_module_g.py:64-223 and
utils/ops.py:141-158,236-260, not from grid_gcn_amd/gridconv.py.  Checked here on the CPU:
  * the restatement reproduces its own fixtures (drift),
  * the product's stock-op modules (gridconv.SubGUpdate, model_cls.SubGUpdateCls) with the same
    weights agree with them, train and eval BatchNorm.
The GPU tests (tests/test_gpu_gridconv_golden.py) hold the HIP kernels to the same fixtures.

Bars.  (1) The stock modules evaluated in float64 equal the restatement to 1e-9: channel order,
concat order, BatchNorm axis/eps/variance convention, mask and ReLU placement are pinned EXACTLY.
(2) fp32, eval-mode BatchNorm (moving statistics): |got - want| <= 1e-5 * max(1, max|want|), the
north_star bar.  (3) fp32, train-mode BatchNorm: NO fp32 evaluation can meet 1e-5 here -- the batch
mean of the attention inputs (coordinates, |mean| ~ 1, sigma ~ 0.03) is subtracted from O(1) values,
so one ulp of the mean is amplified by |mean|/sigma; the stock PyTorch fp32 ops sit at ~2e-5 * max|x|.
The bar is therefore relative to that: the HIP kernels' error against the float64 truth must stay
within 2x the stock fp32 ops' error on the same inputs (GPU test), and the stock ops themselves within
1e-4 * max|x| (here).
"""
import os

import numpy as np
import pytest
import torch

import gridconv_cases as gc
from oracle.torch_index_ops import OracleIndexOps

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def check(got, want, absmax, what):
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    bound = TOL * max(1.0, float(absmax))
    assert err <= bound, "%s: max |err| %.3g > %.3g (max|x| %.3g)" % (what, err, bound, absmax)
    return err


@pytest.mark.parametrize("name", list(gc.CASES))
def test_restatement_reproduces_fixture(name):
    case = gc.CASES[name]()
    g = golden(name)
    for mode, train in (("train", True), ("eval", False)):
        out = gc.reference_output(case, train)[:, case["rows"], :]
        assert np.abs(out - g[mode]).max() <= 1e-6 * max(1.0, float(g[mode + "_absmax"]))


def run_stock(case, m, train, dtype=torch.float32):
    """the product's stock-op path on the CPU: gather (utils/ops.py:78-93) + module forward."""
    m.train(train)
    src = torch.from_numpy(case["src"]).to(dtype)
    nb = OracleIndexOps.batch_take_g(src, torch.from_numpy(case["nebidx"]))
    cent = torch.from_numpy(case["cent"]).to(dtype)
    cm = None if case["centmsk"] is None else torch.from_numpy(case["centmsk"]).to(dtype)
    with torch.no_grad():
        if case["kind"] == "seg":
            cof = None if case["center_ori_feats"] is None else \
                torch.from_numpy(case["center_ori_feats"]).to(dtype)
            out = m(cent[..., 0:3], nb, cm, center_ori_feats=cof)
        else:
            out = m(cent[..., 0:3], nb, cm)
    return out.numpy()


@pytest.mark.parametrize("name", list(gc.CASES))
def test_stock_modules_fp64_equal_restatement(name):
    """semantics pinned exactly: same graph, float64 on both sides"""
    case = gc.CASES[name]()
    m = gc.build_module_f64(case)
    for train in (False, True):     # eval first: the train forward updates the moving statistics
        want = gc.reference_output(case, train)
        got = run_stock(case, m, train, torch.float64)
        err = np.abs(got - want).max()
        assert err <= 1e-9 * max(1.0, np.abs(want).max()), (name, train, err)


@pytest.mark.parametrize("name", list(gc.CASES))
def test_stock_modules_fp32_match_fixture(name):
    case = gc.CASES[name]()
    g = golden(name)
    m = gc.build_module(case)
    out = run_stock(case, m, False)[:, case["rows"], :]
    check(out, g["eval"], g["eval_absmax"], "%s eval stock fp32" % name)
    out = run_stock(case, m, True)[:, case["rows"], :]
    err = np.abs(out.astype(np.float64) - g["train"]).max()
    assert err <= 1e-4 * max(1.0, float(g["train_absmax"])), (name, err)
