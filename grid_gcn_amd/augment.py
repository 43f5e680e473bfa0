"""Device-side batch augmentation of point clouds (the reference does it in numpy on the host, one
cloud at a time: utils/utils.py:158-178 rotate_point_cloud, :274-297 rotate_perturbation_point_cloud,
:348-360 jitter_point_cloud, :363-375 shift_point_cloud, :378-390 random_scale_point_cloud, :408-420
random_point_dropout, :393-405 shuffle_points; composed by the loaders' _augment_batch_data_level1..10,
data_loader/new_ggcn_gpu_modelnet_loader.py:163-260).

Every transform here takes the whole [B, N, 3(+)] batch on whatever device it lives on and runs as a
handful of batched tensor ops, so a training loop that keeps its clouds in HBM never goes back to
the host.  The random draws can be handed in (`angles=`, `noise=`, ...): with the reference's numpy
draws the result equals the reference's formula (tests/test_augment.py); otherwise they come from a
torch.Generator on the data's device.
"""
import math

import torch


def _rand(shape, like, gen):
    return torch.rand(shape, device=like.device, dtype=like.dtype, generator=gen)


def _randn(shape, like, gen):
    return torch.randn(shape, device=like.device, dtype=like.dtype, generator=gen)


def rotate_point_cloud(batch, angles=None, gen=None):
    """rotation about the y (up) axis, one angle in [0, 2 pi) per cloud (utils.py:158-178):
    row-vector convention xyz' = xyz @ [[c,0,s],[0,1,0],[-s,0,c]]."""
    B = batch.shape[0]
    if angles is None:
        angles = _rand((B,), batch, gen) * (2 * math.pi)
    c, s = torch.cos(angles), torch.sin(angles)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    R = torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1),
                     torch.stack([-s, z, c], -1)], -2)                    # [B,3,3]
    out = batch.clone()
    out[..., 0:3] = torch.bmm(batch[..., 0:3], R)
    return out


def rotate_perturbation_point_cloud(batch, angles=None, angle_sigma=0.06, angle_clip=0.18, gen=None):
    """small random rotation R = Rz Ry Rx per cloud, angles ~ clip(sigma * N(0,1)) (utils.py:274-297)."""
    B = batch.shape[0]
    if angles is None:
        angles = torch.clamp(angle_sigma * _randn((B, 3), batch, gen), -angle_clip, angle_clip)
    cx, sx = torch.cos(angles[:, 0]), torch.sin(angles[:, 0])
    cy, sy = torch.cos(angles[:, 1]), torch.sin(angles[:, 1])
    cz, sz = torch.cos(angles[:, 2]), torch.sin(angles[:, 2])
    z, o = torch.zeros_like(cx), torch.ones_like(cx)
    Rx = torch.stack([torch.stack([o, z, z], -1), torch.stack([z, cx, -sx], -1),
                      torch.stack([z, sx, cx], -1)], -2)
    Ry = torch.stack([torch.stack([cy, z, sy], -1), torch.stack([z, o, z], -1),
                      torch.stack([-sy, z, cy], -1)], -2)
    Rz = torch.stack([torch.stack([cz, -sz, z], -1), torch.stack([sz, cz, z], -1),
                      torch.stack([z, z, o], -1)], -2)
    R = torch.bmm(Rz, torch.bmm(Ry, Rx))
    out = batch.clone()
    out[..., 0:3] = torch.bmm(batch[..., 0:3], R)
    return out


def jitter_point_cloud(batch, noise=None, sigma=0.01, clip=0.05, gen=None):
    """per-point jitter clip(sigma * N(0,1), -clip, clip) (utils.py:348-360)."""
    assert clip > 0
    if noise is None:
        noise = _randn(batch.shape, batch, gen)
    return batch + torch.clamp(sigma * noise, -clip, clip)


def shift_point_cloud(batch, shifts=None, shift_range=0.1, gen=None):
    """one uniform shift in [-r, r]^3 per cloud (utils.py:363-375)."""
    B = batch.shape[0]
    if shifts is None:
        shifts = (_rand((B, 3), batch, gen) * 2 - 1) * shift_range
    return batch + shifts[:, None, :]


def random_scale_point_cloud(batch, scales=None, scale_low=0.8, scale_high=1.25, gen=None):
    """one uniform scale per cloud (utils.py:378-390)."""
    B = batch.shape[0]
    if scales is None:
        scales = scale_low + (scale_high - scale_low) * _rand((B,), batch, gen)
    return batch * scales[:, None, None]


def random_point_dropout(batch, ratios=None, u=None, max_dropout_ratio=0.875, gen=None):
    """per cloud: ratio = U * max_ratio, every point with u <= ratio is replaced by the cloud's first
    point (utils.py:408-420)."""
    B, N = batch.shape[0], batch.shape[1]
    if ratios is None:
        ratios = _rand((B,), batch, gen) * max_dropout_ratio
    if u is None:
        u = _rand((B, N), batch, gen)
    drop = u <= ratios[:, None]
    return torch.where(drop[..., None], batch[:, 0:1, :].expand_as(batch), batch)


def shuffle_points(batch, perm=None, gen=None):
    """the same permutation of the points for every cloud of the batch (utils.py:393-405)."""
    if perm is None:
        perm = torch.randperm(batch.shape[1], device=batch.device, generator=gen)
    return batch[:, perm, :]


def augment_batch(batch, level=1, dropout_ratio=0.0, gen=None):
    """_augment_batch_data_level{1..10} of the ModelNet loader (new_ggcn_gpu_modelnet_loader.py:
    163-260) for clouds without normals, on the batch's device."""
    xyz = lambda t: t[..., 0:3]  # noqa: E731

    def put(t, new_xyz):
        out = t.clone()
        out[..., 0:3] = new_xyz
        return out
    d = batch
    if level == 1:
        d = rotate_point_cloud(d, gen=gen)
        d = put(d, jitter_point_cloud(xyz(d), gen=gen))
    elif level in (2, 3):
        d = rotate_perturbation_point_cloud(rotate_point_cloud(d, gen=gen), gen=gen)
        j = random_scale_point_cloud(xyz(d), gen=gen, **(dict(scale_low=1.0, scale_high=1.15)
                                                         if level == 2 else {}))
        if level == 3:
            j = shift_point_cloud(j, gen=gen)
        d = put(d, jitter_point_cloud(j, gen=gen))
    elif level == 4:
        d = rotate_perturbation_point_cloud(rotate_point_cloud(d, gen=gen), gen=gen)
    elif level == 5:
        d = put(d, jitter_point_cloud(xyz(d), gen=gen))
    elif level in (6, 7, 8):
        kw = {6: ({}, {}), 7: (dict(scale_low=0.7, scale_high=1.4), dict(shift_range=0.1)),
              8: (dict(scale_low=0.75, scale_high=1.0), dict(shift_range=0.05))}[level]
        d = put(d, shift_point_cloud(random_scale_point_cloud(xyz(d), gen=gen, **kw[0]), gen=gen, **kw[1]))
    elif level == 9:
        d = put(d, random_scale_point_cloud(xyz(d), gen=gen))
    elif level == 10:
        d = put(d, shift_point_cloud(xyz(d), gen=gen))
    else:
        raise ValueError("augmentation level %r" % (level,))
    if dropout_ratio > 0:
        d = random_point_dropout(d, max_dropout_ratio=dropout_ratio, gen=gen)
    return d
