"""grid_gcn_amd.augment (batched, device-side) against a numpy restatement of the reference's
per-cloud host augmentation (utils/utils.py:158-178, 274-297, 348-390, 408-420), fed the same draws."""
import numpy as np
import torch

from grid_gcn_amd import augment

RNG = np.random.default_rng(3)
B, N = 4, 257
PC = RNG.normal(0, 0.5, (B, N, 3)).astype(np.float32)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(got, want, tol=2e-6):
    assert np.abs(got.numpy().astype(np.float64) - want).max() <= tol


def test_rotate_y_matches_reference_formula():
    ang = RNG.uniform(0, 2 * np.pi, B)
    want = np.zeros((B, N, 3))
    for k in range(B):                                            # utils.py:168-177
        c, s = np.cos(ang[k]), np.sin(ang[k])
        want[k] = PC[k].astype(np.float64) @ np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    close(augment.rotate_point_cloud(T(PC), angles=T(ang.astype(np.float32))), want)


def test_rotate_perturbation_matches_reference_formula():
    ang = np.clip(0.06 * RNG.standard_normal((B, 3)), -0.18, 0.18)
    want = np.zeros((B, N, 3))
    for k in range(B):                                            # utils.py:283-296
        a = ang[k]
        Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
        want[k] = PC[k].astype(np.float64) @ (Rz @ (Ry @ Rx))
    close(augment.rotate_perturbation_point_cloud(T(PC), angles=T(ang.astype(np.float32))), want)


def test_jitter_shift_scale_dropout_match_reference_formulas():
    noise = RNG.standard_normal((B, N, 3)).astype(np.float32)
    close(augment.jitter_point_cloud(T(PC), noise=T(noise)),
          PC + np.clip(0.01 * noise.astype(np.float64), -0.05, 0.05))      # utils.py:357-359
    sh = RNG.uniform(-0.1, 0.1, (B, 3)).astype(np.float32)
    close(augment.shift_point_cloud(T(PC), shifts=T(sh)), PC + sh[:, None, :].astype(np.float64))
    sc = RNG.uniform(0.8, 1.25, B).astype(np.float32)
    close(augment.random_scale_point_cloud(T(PC), scales=T(sc)), PC * sc[:, None, None].astype(np.float64))
    ratios = (RNG.random(B) * 0.875).astype(np.float32)
    u = RNG.random((B, N)).astype(np.float32)
    want = PC.copy()
    for b in range(B):                                            # utils.py:413-417
        idx = np.where(u[b] <= ratios[b])[0]
        if len(idx):
            want[b, idx, :] = want[b, 0, :]
    got = augment.random_point_dropout(T(PC), ratios=T(ratios), u=T(u))
    assert np.array_equal(got.numpy(), want)
    perm = RNG.permutation(N)
    assert np.array_equal(augment.shuffle_points(T(PC), perm=T(perm)).numpy(), PC[:, perm, :])


def test_levels_keep_the_contract():
    x = torch.cat([T(PC), torch.ones(B, N, 1)], dim=2)             # xyz + weight column
    g = torch.Generator().manual_seed(0)
    for level in range(1, 11):
        y = augment.augment_batch(x, level=level, dropout_ratio=0.5, gen=g)
        assert y.shape == x.shape and torch.isfinite(y).all()
    # rigid levels preserve pairwise distances; level 9 scales them by one factor per cloud
    y = augment.augment_batch(T(PC), level=4, gen=g)
    d0 = (T(PC)[:, :8, None, :] - T(PC)[:, None, :8, :]).norm(dim=-1)
    d1 = (y[:, :8, None, :] - y[:, None, :8, :]).norm(dim=-1)
    assert torch.allclose(d0, d1, atol=1e-5)
    y = augment.augment_batch(T(PC), level=9, gen=g)
    d9 = (y[:, :8, None, :] - y[:, None, :8, :]).norm(dim=-1)
    ratio = (d9 / d0.clamp_min(1e-9))[:, 0, 1:]
    assert (ratio.max(dim=1).values - ratio.min(dim=1).values).max() < 1e-4
    assert ((ratio[:, 0] >= 0.8 - 1e-5) & (ratio[:, 0] <= 1.25 + 1e-5)).all()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_augmentation_runs_on_the_device():
    """the whole pipeline on HBM-resident clouds: no host round trip, deterministic per generator"""
    dev = "cuda:0"
    x = torch.cat([T(PC), torch.ones(B, N, 1)], dim=2).to(dev)
    outs = []
    for _ in range(2):
        g = torch.Generator(device=dev).manual_seed(3)
        y = x
        for level in (1, 3, 7):
            y = augment.augment_batch(y, level=level, dropout_ratio=0.3, gen=g)
        assert y.device.type == "cuda" and y.shape == x.shape and torch.isfinite(y).all()
        assert torch.equal(y[..., 3], x[..., 3])              # the weight column is untouched
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    y = augment.shuffle_points(x, gen=torch.Generator(device=dev).manual_seed(1))
    assert torch.equal(y.sort(dim=1).values, x.sort(dim=1).values)
