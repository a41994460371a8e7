"""Synthetic scenes for parity tests and bench.py.

numpy only; no oracle, no torch.  Layouts are the reference's boundary layouts
(SURVEY.md A.13): Julia `(3,N)` column-major == numpy `(N,3)` C-order, shs
`(3,K,N)` == numpy `(N,K,3)`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

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
