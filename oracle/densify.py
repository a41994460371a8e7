"""CPU restatement (numpy) of the reference's adaptive density control — TEST INFRASTRUCTURE ONLY (see the header
of gsr_oracle.c): only tests/ may import it.

Follows src/densification.jl:1-297 (`densify_and_prune!`, `densify_clone!`, `densify_split!`,
`_add_split_noise!`, `prune_points!`, `densification_postfix!`, `append_gaussians!`, `_append_optimizer!`,
`_prune_optimizer!`), src/strategy.jl:28-136 (`DefaultStrategy`, `post_train_step!`, `update_stats!`) and
src/gaussians.jl:119-137 (`reset_opacity!`, `inverse_sigmoid`) statement by statement.  Arrays are the C-order
equivalents of the Julia ones (Gaussian index FIRST): points (N,3), features_dc (N,1,3), features_rest (N,K-1,3),
scales (N,3) or (N,1), rotations (N,4), opacities (N,1); an optimizer is a dict(mu=, nu=, step=) of FLAT moment
vectors (`opt.μ[1]`, `opt.ν[1]`, training.jl:396-413).

Parity unpinned against a live reference in one respect: `_add_split_noise!` draws `randn(Float32)` from the
backend's device RNG (densification.jl:128), which nothing can reproduce; here — and in the HIP kernel — the
normals come from a counter-based integer generator keyed by (seed, row) followed by Box-Muller.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

f32 = np.float32
PARAMS = ("points", "features_dc", "features_rest", "scales", "rotations", "opacities")  # append/prune order


def sigmoid(x):
    """NU.sigmoid"""
    x = np.asarray(x, f32)
    return (f32(1) / (f32(1) + np.exp(-x, dtype=f32))).astype(f32)


def inverse_sigmoid(x):
    """gaussians.jl:137"""
    x = np.asarray(x, f32)
    return np.log(x / (f32(1) - x), dtype=f32)


@dataclass
class Model:
    points: np.ndarray
    features_dc: np.ndarray
    features_rest: np.ndarray
    scales: np.ndarray
    rotations: np.ndarray
    opacities: np.ndarray

    def __len__(self):
        return self.points.shape[0]

    def copy(self):
        return Model(*[getattr(self, k).copy() for k in PARAMS])


@dataclass
class Strategy:
    """DefaultStrategy (strategy.jl:28-66)"""
    max_radii: np.ndarray
    accum_grad_means_2d: np.ndarray
    denom: np.ndarray
    dense_percent: float = 1e-2
    densify_from_iter: int = 500
    densify_until_iter: int = 15_000
    densification_interval: int = 100
    densify_grad_threshold: float = 2e-4
    opacity_reset_interval: int = 3_000
    min_opacity: float = 0.005

    @classmethod
    def for_model(cls, n, **kw):
        return cls(np.zeros(n, np.int32), np.zeros(n, f32), np.zeros(n, f32), **kw)


def new_optimizers(gs: Model):
    return {k: dict(mu=np.zeros(getattr(gs, k).size, f32), nu=np.zeros(getattr(gs, k).size, f32), step=0) for k in PARAMS}


def max_exp_scale(scales):
    """reshape(maximum(exp.(gs.scales); dims=1), :)"""
    return np.exp(np.asarray(scales, f32), dtype=f32).max(axis=1)


# ---- the split noise generator (shared, bit for bit in its integer part, with densify.hip) ----
def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def uniform01(seed, rows, draw):
    with np.errstate(over="ignore"):
        h = _mix32(_mix32(np.uint32(seed) ^ (rows.astype(np.uint32) * np.uint32(0x9E3779B9))) + np.uint32(draw) * np.uint32(0x85EBCA6B))
    return ((h >> np.uint32(8)).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)


def randn3(seed, n):
    rows = np.arange(n, dtype=np.uint32)
    u1, u2, u3, u4 = (uniform01(seed, rows, d) for d in range(4))
    r1 = np.sqrt(f32(-2) * np.log(u1, dtype=f32), dtype=f32)
    r2 = np.sqrt(f32(-2) * np.log(u3, dtype=f32), dtype=f32)
    tp = f32(6.2831853071795864)
    return np.stack([r1 * np.cos(tp * u2, dtype=f32), r1 * np.sin(tp * u2, dtype=f32), r2 * np.cos(tp * u4, dtype=f32)], 1).astype(f32)


def unnorm_quat2rot(q):
    """render.jl:322-333, row-major (N,3,3)"""
    q = np.asarray(q, f32)
    inv = f32(1) / np.sqrt((q * q).sum(1, dtype=f32), dtype=f32)
    w, x, y, z = (q[:, k] * inv for k in range(4))
    x2, y2, z2, xy, xz, yz, wx, wy, wz = x * x, y * y, z * z, x * y, x * z, y * z, w * x, w * y, w * z
    R = np.empty((q.shape[0], 3, 3), f32)
    R[:, 0, 0] = 1 - 2 * (y2 + z2); R[:, 0, 1] = 2 * (xy - wz); R[:, 0, 2] = 2 * (xz + wy)
    R[:, 1, 0] = 2 * (xy + wz); R[:, 1, 1] = 1 - 2 * (x2 + z2); R[:, 1, 2] = 2 * (yz - wx)
    R[:, 2, 0] = 2 * (xz - wy); R[:, 2, 1] = 2 * (yz + wx); R[:, 2, 2] = 1 - 2 * (x2 + y2)
    return R


# ---- optimizer state edits (densification.jl:255-297) ----
def _append_optimizer(opt, extension):
    z = np.zeros(extension.size, f32)
    opt["mu"] = np.concatenate([opt["mu"], z]); opt["nu"] = np.concatenate([opt["nu"], z])


def _prune_optimizer(opt, mask, x):
    shape = x.shape
    opt["mu"] = np.ascontiguousarray(opt["mu"].reshape(shape)[mask]).reshape(-1)
    opt["nu"] = np.ascontiguousarray(opt["nu"].reshape(shape)[mask]).reshape(-1)


def _sel(gs: Model, k, mask, reps=1):
    """x[:, mask] (repeated `reps` times as a block: Julia's repeat(x, 1, reps)); an empty features_rest is passed
    through untouched (densification.jl:40-41,85-86)"""
    x = getattr(gs, k)
    if k == "features_rest" and x.size == 0:
        return x[:0]
    return np.tile(x[mask], (reps,) + (1,) * (x.ndim - 1))


def append_gaussians(gs: Model, optimizers, new):
    """densification.jl:214-253"""
    for k in PARAMS:
        if k == "features_rest" and gs.features_rest.size == 0:
            continue
        _append_optimizer(optimizers[k], new[k])
        setattr(gs, k, np.concatenate([getattr(gs, k), new[k]], 0))
    if gs.features_rest.size == 0:
        gs.features_rest = np.zeros((len(gs), 0, 3), f32)


def densification_postfix(strategy: Strategy, gs: Model, optimizers, new):
    """densification.jl:193-210: append, then the statistics restart from zero for the WHOLE model"""
    append_gaussians(gs, optimizers, new)
    n = len(gs)
    strategy.max_radii = np.zeros(n, np.int32)
    strategy.accum_grad_means_2d = np.zeros(n, f32)
    strategy.denom = np.zeros(n, f32)


def prune_points(strategy: Strategy, gs: Model, optimizers, valid_mask):
    """densification.jl:138-191"""
    for k in PARAMS:
        if k == "features_rest" and gs.features_rest.size == 0:
            continue
        _prune_optimizer(optimizers[k], valid_mask, getattr(gs, k))
        setattr(gs, k, np.ascontiguousarray(getattr(gs, k)[valid_mask]))
    if gs.features_rest.size == 0:
        gs.features_rest = np.zeros((len(gs), 0, 3), f32)
    strategy.max_radii = strategy.max_radii[valid_mask]
    strategy.accum_grad_means_2d = strategy.accum_grad_means_2d[valid_mask]
    strategy.denom = strategy.denom[valid_mask]


def densify_clone(strategy, gs, optimizers, grad, grad_threshold, extent, dense_percent):
    """densification.jl:29-62"""
    gamma = f32(extent) * f32(dense_percent)
    mask = (grad > f32(grad_threshold)) & (max_exp_scale(gs.scales) < gamma)
    new = {k: _sel(gs, k, mask) for k in PARAMS}
    densification_postfix(strategy, gs, optimizers, new)
    return mask


def densify_split(strategy, gs, optimizers, grad, grad_threshold, extent, dense_percent, seed):
    """densification.jl:64-119 (+ _add_split_noise! :121-135)"""
    n, n_split = len(gs), 2
    padded = np.zeros(n, f32)
    padded[:grad.shape[0]] = grad
    gamma = f32(extent) * f32(dense_percent)
    mask = (padded >= f32(grad_threshold)) & (max_exp_scale(gs.scales) > gamma)
    stds = np.tile(np.exp(gs.scales[mask], dtype=f32), (n_split, 1))             # repeat(..., 1, n_split): block repeat
    new = {k: _sel(gs, k, mask, n_split) for k in PARAMS}
    new["scales"] = np.log(stds / (f32(0.8) * f32(n_split)), dtype=f32)
    m = new["points"].shape[0]
    if m > 0:
        xi = stds * randn3(seed, m)                                                   # σ .* randn3 (isotropic: σ broadcasts)
        R = unnorm_quat2rot(new["rotations"])
        step = (R[:, :, 0] * xi[:, 0:1] + R[:, :, 1] * xi[:, 1:2]) + R[:, :, 2] * xi[:, 2:3]
        new["points"] = (new["points"] + step).astype(f32)
    densification_postfix(strategy, gs, optimizers, new)
    valid = np.concatenate([~mask, np.ones(m, bool)])
    prune_points(strategy, gs, optimizers, valid)
    return mask


def densify_and_prune(strategy: Strategy, gs: Model, optimizers, extent, pruning_extent, max_screen_size, seed=0):
    """densification.jl:1-27"""
    with np.errstate(invalid="ignore", divide="ignore"):
        grad = (strategy.accum_grad_means_2d / strategy.denom).astype(f32)
    grad[np.isnan(grad)] = 0
    masks = {}
    masks["clone"] = densify_clone(strategy, gs, optimizers, grad, strategy.densify_grad_threshold, extent, strategy.dense_percent)
    masks["split"] = densify_split(strategy, gs, optimizers, grad, strategy.densify_grad_threshold, extent,
                                   strategy.dense_percent, seed)
    valid = sigmoid(gs.opacities).reshape(-1) > f32(strategy.min_opacity)
    if max_screen_size > 0:
        gamma = f32(0.1) * f32(pruning_extent)
        valid &= (strategy.max_radii < max_screen_size) & (max_exp_scale(gs.scales) < gamma)
    prune_points(strategy, gs, optimizers, valid)
    masks["valid"] = valid
    return masks


def reset_opacity(gs: Model):
    """gaussians.jl:115-126"""
    gs.opacities = inverse_sigmoid(np.minimum(f32(0.1), sigmoid(gs.opacities))).astype(f32)


def post_train_step(strategy: Strategy, gs: Model, optimizers, radii, grad_means_2d, resolution, step, extent, seed=0):
    """strategy.jl:78-105.  Returns what happened: (densified, reset)."""
    if step > strategy.densify_until_iter:
        return False, False
    vis = radii > 0                                                                   # _update_stats!, strategy.jl:118-136
    strategy.max_radii[vis] = np.maximum(strategy.max_radii[vis], radii[vis])
    g = np.asarray(grad_means_2d, f32)
    gx = g[:, 0] * f32(resolution[0]) * f32(0.5); gy = g[:, 1] * f32(resolution[1]) * f32(0.5)
    strategy.accum_grad_means_2d[vis] += np.sqrt(gx * gx + gy * gy, dtype=f32)[vis]
    strategy.denom[vis] += f32(1)
    densified = step >= strategy.densify_from_iter and step % strategy.densification_interval == 0
    if densified:
        mss = 20 if step > strategy.opacity_reset_interval else 0
        densify_and_prune(strategy, gs, optimizers, extent, extent, mss, seed)
    reset = step % strategy.opacity_reset_interval == 0
    if reset:
        reset_opacity(gs)
        o = optimizers["opacities"]                                                   # NU.reset!
        o["mu"][:] = 0; o["nu"][:] = 0; o["step"] = 0
    return densified, reset
