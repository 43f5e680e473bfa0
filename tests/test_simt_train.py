"""CPU tier: training kernels of the PRODUCT under the host-side wave64 emulator (tests/simt/: fp32 MFMA as a wave
rendezvous -- a k-ordered chain of fused multiply-adds, as the hardware's -- raw buffer loads / stores with the
descriptor's per-dword range check), against float64 restatements of what they compute.

Covered here: the Z2-free attention backward of the up layers (csrc/gridgcn_attbwd_nz.hip) in its three forms -- the
round-4 kernel, the round-5 stripped tile loop, and the round-6 form that takes S1 / S2 from the forward's moments
(gridgcn_att_bwd_noz_mom, written while the GPU pool was closed: this is what has executed it) -- and the moments pass
of the forward (csrc/gridgcn_attfwd.hip: gg_k_att_moments + reduce + the BatchNorm from the quadratic form)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from simt import sim  # noqa: E402


@pytest.fixture(autouse=True)
def _no_guesses():
    c0 = sim.counters()
    yield
    c1 = sim.counters()
    assert c1[2] == c0[2], "a cross-lane read took a lane outside the set executing the operation"
    assert c1[3] == c0[3], "a cross-lane operation was reached in divergent control flow"


def _inputs(ncent, P, seed):
    rng = np.random.default_rng(seed)
    cin, C = 32, 128
    E = ncent * P
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    d = dict(E=E, P=P, ncent=ncent)
    d["Z1"] = f(E, cin)
    d["ps"], d["psh"] = np.abs(f(cin)) + 0.5, f(cin) * 0.1
    d["pm"], d["pr"] = f(cin) * 0.1, np.abs(f(cin)) + 0.5
    d["W2"], d["b2"] = f(C, cin) * 0.2, f(C) * 0.1
    d["sc"], d["mu"], d["rs"] = np.abs(f(C)) + 0.5, f(C) * 0.1, np.abs(f(C)) + 0.5
    d["bsums"] = rng.standard_normal((2, C)) * E ** 0.5
    d["amax"] = rng.integers(0, P, (ncent, C)).astype(np.uint8)
    d["gval"] = f(ncent, C)
    return d


def _reference(d):
    """float64: dZ2 = sparse + (z2 - mu) bz + cz;  dA1 = dZ2 W2;  dW2 = dZ2^T a1;  sums of the layer in front"""
    E, P, ncent = d["E"], d["P"], d["ncent"]
    D = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in d.items()}
    a1 = np.maximum(D["Z1"] * D["ps"] + D["psh"], 0.0)
    z2 = a1 @ D["W2"].T + D["b2"]
    m1, m2 = D["bsums"][0] / E, D["bsums"][1] / E
    bz = -(D["sc"] * D["rs"]) * m2
    cz = -(D["sc"] * m1)
    dZ2 = (z2 - D["mu"]) * bz + cz
    rows = (np.arange(ncent)[:, None] * P + d["amax"].astype(np.int64))          # [ncent, C]
    cols = np.broadcast_to(np.arange(128), rows.shape)
    np.add.at(dZ2, (rows, cols), D["gval"])
    dA1 = dZ2 @ D["W2"]
    dW2 = dZ2.T @ a1
    g1 = np.where(a1 > 0, dA1, 0.0)
    zhat = D["Z1"] * D["pr"] - D["pm"] * D["pr"]
    return dict(dX=dA1, dW=dW2, psums=np.stack([g1.sum(0), (g1 * zhat).sum(0)]), s1=a1.sum(0), m1=m1, m2=m2,
                dgamma=D["bsums"][1], dbeta=D["bsums"][0], a1=a1)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


CASES = [(64, 5, 1), (67, 3, 2), (200, 5, 3), (33, 1, 4), (13, 7, 5), (410, 5, 6)]


@pytest.mark.parametrize("ncent,P,seed", CASES)
def test_att_bwd_noz_three_forms(ncent, P, seed):
    d = _inputs(ncent, P, seed)
    ref = _reference(d)
    args = [d[k] for k in ("Z1", "ps", "psh", "pm", "pr", "W2", "b2", "sc", "mu", "rs", "bsums", "amax", "gval")]
    r5 = sim.att_bwd_noz(*args, P, v2=1)
    r4 = sim.att_bwd_noz(*args, P, v2=0)
    # round 5 == round 4, word for word (the same arithmetic in the same order)
    assert np.array_equal(r5["dX"], r4["dX"]) and np.array_equal(r5["dW"], r4["dW"]) and np.array_equal(r5["v"], r4["v"])
    assert np.allclose(r5["psums"], r4["psums"], rtol=1e-12, atol=0) and np.allclose(r5["s1"], r4["s1"], rtol=1e-12)
    # against float64
    assert _rel(r5["dX"], ref["dX"]) < 5e-6
    assert _rel(r5["dW"], ref["dW"]) < 2e-5
    assert _rel(r5["psums"], ref["psums"]) < 2e-5
    assert _rel(r5["s1"], ref["s1"]) < 2e-6
    assert np.allclose(r5["v"][0], ref["m1"], rtol=1e-6) and np.allclose(r5["v"][1], ref["m2"], rtol=1e-6)
    assert np.allclose(r5["v"][2], ref["dgamma"], rtol=1e-6) and np.allclose(r5["v"][3], ref["dbeta"], rtol=1e-6)
    # round 6: S1 / S2 from the forward's moments
    gamma = np.ones(128, np.float32)
    beta = np.zeros(128, np.float32)
    _, _, mom = sim.att_bn2_moments(d["Z1"], d["ps"], d["psh"], d["W2"], d["b2"], gamma, beta)
    r6 = sim.att_bwd_noz(*args, P, mom=mom)
    assert np.array_equal(r6["dX"], r5["dX"])                     # the dX path is untouched
    assert np.array_equal(r6["v"], r5["v"])
    assert np.allclose(r6["psums"], r5["psums"], rtol=1e-12, atol=0)
    assert _rel(r6["dW"], ref["dW"]) < 2e-5
    assert _rel(r6["dW"], r5["dW"]) < 2e-6
    assert np.all(r6["s1"] == 0.0)                                # (not produced by this form)


@pytest.mark.parametrize("E", [32, 33, 511, 512, 2049])
def test_att_bn2_moments_against_the_materialised_conv(E):
    """S1, S2 of a1 = relu(Z1 * s1 + h1) in fp64 from fp32 MFMA products; the BatchNorm of z2 = W2 a1 + b2 from the
    quadratic form against float64 statistics of the materialised z2"""
    rng = np.random.default_rng(E)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    Z1 = f(E, 32)
    s1, h1 = np.abs(f(32)) + 0.5, f(32) * 0.3
    W2, b2 = f(128, 32) * 0.2, f(128) * 0.1
    gamma, beta = np.abs(f(128)) + 0.5, f(128) * 0.3
    eps = 1e-3
    vec, sums, mom = sim.att_bn2_moments(Z1, s1, h1, W2, b2, gamma, beta, eps)
    a1 = np.maximum(Z1.astype(np.float64) * s1 + h1, 0.0)
    S1, S2 = a1.sum(0), a1.T @ a1
    S1k = mom[1024:1056] + mom[1056:1088]
    assert _rel(S1k, S1) < 1e-6
    S2k = np.empty((32, 32))
    for k in range(32):
        for i in range(32):
            S2k[k, i] = mom[((k & 3) + 4 * (k >> 3)) * 64 + i + 32 * ((k >> 2) & 1)]
    assert _rel(S2k, S2) < 2e-6
    z2 = a1 @ W2.astype(np.float64).T + b2
    mean, var = z2.mean(0), z2.var(0)
    rstd = 1.0 / np.sqrt(var + eps)
    assert np.abs(vec[2] - mean).max() < 2e-6 * max(1.0, np.abs(mean).max())
    assert _rel(vec[3], rstd) < 5e-6
    assert _rel(vec[0], gamma * rstd) < 5e-6
    assert np.abs(vec[1] - (beta - mean * gamma * rstd)).max() < 1e-5
    assert _rel(sums[0], z2.sum(0)) < 1e-6 and _rel(sums[1], (z2 * z2).sum(0)) < 2e-6
