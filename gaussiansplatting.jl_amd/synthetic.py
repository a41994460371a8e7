"""Synthetic scenes for parity tests and bench.py.

numpy only; no oracle, no torch.  Layouts are the reference's boundary layouts
(SURVEY.md A.13): Julia `(3,N)` column-major == numpy `(N,3)` C-order, shs
`(3,K,N)` == numpy `(N,K,3)`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace

import numpy as np


@dataclass
class Scene:
    means: np.ndarray        # (N,3)
    scales_raw: np.ndarray   # (N,3) log-scales (pre-activation)
    rotations: np.ndarray    # (N,4) w,x,y,z un-normalised
    opacities_raw: np.ndarray  # (N,) logits
    shs: np.ndarray          # (N,K,3)
    sh_degree: int
    width: int
    height: int
    focal: tuple
    principal: tuple = (0.5, 0.5)

    @property
    def n(self):
        return self.means.shape[0]

    @property
    def scales(self):
        """exp activation (rasterizer.jl:236-237)"""
        return np.exp(self.scales_raw).astype(np.float32)

    @property
    def opacities(self):
        """sigmoid activation (rasterizer.jl:228-229)"""
        return (1.0 / (1.0 + np.exp(-self.opacities_raw.astype(np.float64)))).astype(np.float32)


def make_scene(n: int, width: int, height: int, sh_degree: int = 3, seed: int = 1002,
               sigma_px: float = 3.0, K: int | None = None) -> Scene:
    rng = np.random.default_rng(seed)
    f32 = np.float32
    fx = 0.5 * width / math.tan(math.radians(30.0))
    fy = fx
    z = rng.uniform(2.0, 12.0, n)
    u = rng.uniform(-1.0, 1.0, n)
    v = rng.uniform(-1.0, 1.0, n)
    x = 1.1 * z * u * width / (2.0 * fx)
    y = 1.1 * z * v * height / (2.0 * fy)
    means = np.stack([x, y, z], 1).astype(f32)
    n0 = rng.standard_normal(n)
    nk = rng.standard_normal((n, 3))
    log_s = np.log(sigma_px * z / fx)[:, None] + 0.35 * n0[:, None] + 0.3 * nk
    rotations = rng.standard_normal((n, 4)).astype(f32)
    opac = rng.normal(-1.0, 1.0, n).astype(f32)
    if K is None:
        K = (sh_degree + 1) ** 2
    shs = np.empty((n, K, 3), f32)
    shs[:, 0, :] = rng.normal(0.0, 0.5, (n, 3))
    if K > 1:
        shs[:, 1:, :] = rng.normal(0.0, 0.1, (n, K - 1, 3))
    return Scene(means, log_s.astype(f32), rotations, opac, shs, sh_degree, width, height, (f32(fx), f32(fy)))


def view_pose(j: int, n_views: int = 8):
    """Multi-view batch of config 4: R = R_y(5°·(j-3.5)), t = -R·(0.3·(j-3.5),0,0).
    Returns row-major R (3,3) and t (3,) float32."""
    c = j - (n_views - 1) / 2.0
    a = math.radians(5.0 * c)
    R = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    t = -R @ np.array([0.3 * c, 0.0, 0.0])
    return R.astype(np.float32), t.astype(np.float32)


def make_target(width: int, height: int, seed: int) -> np.ndarray:
    """U(0,1) target image, (3,H,W) C-order == Julia (W,H,3)."""
    return np.random.default_rng(seed + 7919).uniform(0.0, 1.0, (3, height, width)).astype(np.float32)


def make_vpixels(width: int, height: int, channels: int, seed: int) -> np.ndarray:
    """Cotangent for loss-free fwd+bwd configs: N(0,1)/(C·P), (H,W,C)."""
    p = width * height
    g = np.random.default_rng(seed + 104729).standard_normal((height, width, channels))
    return (g / (channels * p)).astype(np.float32)


def morton_order(means: np.ndarray, bits: int = 10) -> np.ndarray:
    """Permutation that sorts points along a 3-D Morton (Z-order) curve: neighbours in the array are neighbours in space,
    so the Gaussians of a wave project into the same few tiles whatever the camera.  What a caller would apply to the
    parameters (and Adam moments) at densification time; the binning atomics of a wave then fall into a few cache lines
    instead of 64 (tools/atomic_rates.hip: 129 vs 23 G atomics/s)."""
    lo, hi = means.min(0), means.max(0)
    q = np.clip(((means - lo) / np.maximum(hi - lo, 1e-30) * ((1 << bits) - 1)).astype(np.uint64), 0, (1 << bits) - 1)
    code = np.zeros(means.shape[0], np.uint64)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + d)
    return np.argsort(code, kind="stable")


def reorder(scene: Scene, perm: np.ndarray) -> Scene:
    """The same scene with its Gaussians permuted (every per-Gaussian array)."""
    return replace(scene, means=scene.means[perm], scales_raw=scene.scales_raw[perm], rotations=scene.rotations[perm],
                   opacities_raw=scene.opacities_raw[perm], shs=scene.shs[perm])


def add_skew(scene: Scene, kind: str, seed: int = 7) -> Scene:
    """Skewed variants of the synthetic scene (tile-list length far from uniform), for the skew bench / tests:
      "hot:K"      K extra Gaussians whose means project into ONE tile (the central one): a list of ~K instances
      "dense:P:F"  a fraction P of the tiles (random) gets F x the mean density (extra Gaussians centred in them)
    Extra Gaussians share the distribution of everything else (scales for sigma_px = 3, random rotations, ...)."""
    rng = np.random.default_rng(seed)
    W, H = scene.width, scene.height
    fx, fy = float(scene.focal[0]), float(scene.focal[1])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    parts = kind.split(":")
    if parts[0] == "hot":
        k = int(parts[1])
        tiles = np.full(k, (gy // 2) * gx + gx // 2)
    elif parts[0] == "dense":
        frac, factor = float(parts[1]), float(parts[2])
        hot = rng.choice(gx * gy, max(1, int(frac * gx * gy)), replace=False)
        per_tile = int(factor * scene.n * 4.5 / (gx * gy) / 4.5)  # factor x the mean number of Gaussian centres per tile
        tiles = np.repeat(hot, per_tile)
        k = tiles.shape[0]
    else:
        raise ValueError(f"unknown skew kind {kind!r}")
    tx, ty = tiles % gx, tiles // gx
    px = tx * 16 + rng.uniform(0.0, 16.0, k)
    py = ty * 16 + rng.uniform(0.0, 16.0, k)
    z = rng.uniform(2.0, 12.0, k)
    x = (px - 0.5 * W) * z / fx
    y = (py - 0.5 * H) * z / fy
    extra = make_scene(k, W, H, scene.sh_degree, seed + 1, K=scene.shs.shape[1])
    means = np.stack([x, y, z], 1).astype(np.float32)
    log_s = (np.log(3.0 * z / fx)[:, None] + (extra.scales_raw - np.log(3.0 * extra.means[:, 2:3] / fx))).astype(np.float32)
    cat = lambda a, b: np.ascontiguousarray(np.concatenate([a, b], 0))  # noqa: E731
    return Scene(cat(scene.means, means), cat(scene.scales_raw, log_s), cat(scene.rotations, extra.rotations),
                 cat(scene.opacities_raw, extra.opacities_raw), cat(scene.shs, extra.shs), scene.sh_degree, W, H, scene.focal,
                 scene.principal)
