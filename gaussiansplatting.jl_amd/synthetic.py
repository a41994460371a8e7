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


def _quat_to_mat(q: np.ndarray) -> np.ndarray:
    """(n,4) w,x,y,z (any norm) -> (n,3,3) rotation matrices (projection.jl:259-275 convention)."""
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _quat_from_z_to(nrm: np.ndarray, spin: np.ndarray) -> np.ndarray:
    """Unit quaternions (w,x,y,z) rotating the local z axis onto `nrm` (n,3), after a spin about z by `spin` radians."""
    nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    # shortest arc z -> nrm: q = (1 + z.n, z x n), normalised; n = -z handled by a half turn about x
    w = 1.0 + nrm[:, 2]
    q = np.stack([w, -nrm[:, 1], nrm[:, 0], np.zeros_like(w)], 1)
    flip = w < 1e-9
    q[flip] = np.array([0.0, 1.0, 0.0, 0.0])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    s = np.stack([np.cos(0.5 * spin), np.zeros_like(spin), np.zeros_like(spin), np.sin(0.5 * spin)], 1)
    # Hamilton product q * s
    w1, x1, y1, z1 = q.T
    w2, x2, y2, z2 = s.T
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1)


def make_trained_like(n: int, width: int, height: int, sh_degree: int = 3, seed: int = 1010,
                      sigma_px: float = 4.0, K: int | None = None, subclip_fraction: float = 0.3) -> Scene:
    """A procedural scene with the statistics of a TRAINED capture rather than of `make_scene`'s uniform cloud
    (what the reference trains on and benchmarks with: benchmark/pipeline.jl:19-39, Mip-NeRF360 "bicycle"):

      * Gaussians sit on a few SURFACES seen from the origin along +z — a ground plane running to the horizon (35 %),
        a back wall (20 %), a side wall (10 %), a sphere shell in front (25 %) — plus 10 % floaters in the frustum;
      * they are FLAT and anisotropic: two in-plane axes log-normal around `sigma_px` pixels at their depth, the axis
        along the surface normal 10–100 times thinner (log-uniform); oriented along the surface (+ jitter); the ground
        plane at grazing angles therefore projects to needles, and its density per tile grows towards the horizon;
      * opacity is BIMODAL (55 % logit ~ N(3,1), 45 % logit ~ N(-2.5,1));
      * `subclip_fraction` of the splats are tiny (projected radius <= radius_clip = 3 px: culled by project!,
        projection.jl:104) — what a trained model is full of between two prune rounds;
      * rows are in DENSIFICATION order: a seed set in random order, then generations of children appended in their
        parents' order right where densify_clone!/densify_split! put them (densification.jl:47,90): a child sits within
        its parent's extent, half of the children are splits (scale / 1.6).
    numpy only, deterministic in (n, width, height, sh_degree, seed)."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    fx = 0.5 * width / math.tan(math.radians(30.0))
    fy = fx
    if K is None:
        K = (sh_degree + 1) ** 2
    gens, growth = 3, 1.8
    n0 = max(1, int(math.ceil(n / growth ** gens)))
    # --- seed set: positions + normals on the surfaces
    kind = rng.choice(5, n0, p=[0.35, 0.20, 0.10, 0.25, 0.10])
    pos = np.zeros((n0, 3))
    nrm = np.zeros((n0, 3))
    u, v = rng.uniform(0, 1, n0), rng.uniform(0, 1, n0)
    hx = 0.5 * width / fx   # tan of the half field of view
    hy = 0.5 * height / fy
    g = kind == 0           # ground: y = 1.2 below the camera, from 1.5 to 30 units ahead, wider than the frustum
    zg = 1.5 + 28.5 * u[g] ** 1.5
    pos[g] = np.stack([(2 * v[g] - 1) * 1.15 * hx * zg, np.full(g.sum(), 1.2), zg], 1)
    nrm[g] = [0.0, -1.0, 0.0]
    b = kind == 1           # back wall at z = 14
    pos[b] = np.stack([(2 * u[b] - 1) * 1.15 * hx * 14.0, -1.1 * hy * 14.0 + (1.2 + 1.1 * hy * 14.0) * v[b], np.full(b.sum(), 14.0)], 1)
    nrm[b] = [0.0, 0.0, -1.0]
    w = kind == 2           # side wall at x = -3.5, from z = 3 to 14
    zw = 3.0 + 11.0 * u[w]
    pos[w] = np.stack([np.full(w.sum(), -3.5), -3.0 + 4.2 * v[w], zw], 1)
    nrm[w] = [1.0, 0.0, 0.0]
    s = kind == 3           # sphere shell, radius 0.9, centre (0.6, 0.3, 4.5); the far side is occluded, as in a capture
    d = rng.standard_normal((s.sum(), 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos[s] = np.array([0.6, 0.3, 4.5]) + 0.9 * d
    nrm[s] = d
    f = kind == 4           # floaters anywhere in the frustum
    zf = rng.uniform(1.0, 14.0, f.sum())
    pos[f] = np.stack([(2 * u[f] - 1) * hx * zf, (2 * v[f] - 1) * hy * zf, zf], 1)
    d = rng.standard_normal((f.sum(), 3))
    nrm[f] = d / np.linalg.norm(d, axis=1, keepdims=True)
    nrm += 0.08 * rng.standard_normal((n0, 3))   # orientation jitter
    # --- flat anisotropic scales
    z = np.maximum(pos[:, 2], 0.5)
    base = np.log(sigma_px * z / fx)
    ls = np.empty((n0, 3))
    ls[:, 0] = base + 0.6 * rng.standard_normal(n0)
    ls[:, 1] = base + 0.6 * rng.standard_normal(n0)
    ls[:, 2] = np.minimum(ls[:, 0], ls[:, 1]) - rng.uniform(math.log(10.0), math.log(100.0), n0)
    ls[f, 2] = ls[f, 0] - rng.uniform(0.0, math.log(10.0), f.sum())   # floaters are blobs
    quat = _quat_from_z_to(nrm, rng.uniform(0, 2 * math.pi, n0))
    opac = np.where(rng.uniform(0, 1, n0) < 0.55, rng.normal(3.0, 1.0, n0), rng.normal(-2.5, 1.0, n0))
    shs = np.empty((n0, K, 3))
    # low-frequency albedo per surface + noise
    shs[:, 0, :] = 0.6 * np.sin(pos @ rng.normal(0.0, 0.7, (3, 3)) + kind[:, None]) + rng.normal(0.0, 0.15, (n0, 3))
    if K > 1:
        shs[:, 1:, :] = rng.normal(0.0, 0.05, (n0, K - 1, 3))
    # --- densification generations: children appended in their parents' order
    for gen in range(gens):
        cur = pos.shape[0]
        want = n if gen == gens - 1 else int(round(n0 * growth ** (gen + 1)))
        k = max(0, min(cur, want - cur))
        sel = np.sort(rng.choice(cur, k, replace=False))
        R = _quat_to_mat(quat[sel])
        xi = rng.standard_normal((k, 3))
        child_pos = pos[sel] + np.einsum("nij,nj->ni", R, np.exp(ls[sel]) * xi)
        split = rng.uniform(0, 1, k) < 0.5
        child_ls = ls[sel] - np.where(split, math.log(1.6), 0.0)[:, None]
        ls[sel[split]] -= math.log(1.6)
        child_quat = quat[sel] + 0.02 * rng.standard_normal((k, 4))
        child_op = opac[sel] + rng.normal(0.0, 0.3, k)
        child_sh = shs[sel] + rng.normal(0.0, 0.03, (k, K, 3))
        pos = np.concatenate([pos, child_pos]); ls = np.concatenate([ls, child_ls]); quat = np.concatenate([quat, child_quat])
        opac = np.concatenate([opac, child_op]); shs = np.concatenate([shs, child_sh])
    if pos.shape[0] < n:   # tiny n: pad with copies of the first rows
        pad = np.arange(n - pos.shape[0]) % pos.shape[0]
        pos, ls, quat, opac, shs = (np.concatenate([a, a[pad]]) for a in (pos, ls, quat, opac, shs))
    # --- sub-radius_clip splats: projected sigma ~ 0.25-0.6 px whatever they were
    tiny = rng.uniform(0, 1, n) < subclip_fraction
    zt = np.maximum(pos[tiny, 2], 0.5)
    ls[tiny] = np.log(rng.uniform(0.25, 0.6, tiny.sum()) * zt / fx)[:, None] + rng.normal(0.0, 0.1, (tiny.sum(), 3))
    quat = quat * rng.uniform(0.5, 2.0, (n, 1))   # the stored rotations are un-normalised (rasterizer.jl:232-233)
    return Scene(np.ascontiguousarray(pos[:n], f32), np.ascontiguousarray(ls[:n], f32), np.ascontiguousarray(quat[:n], f32),
                 np.ascontiguousarray(opac[:n], f32), np.ascontiguousarray(shs[:n], f32), sh_degree, width, height,
                 (f32(fx), f32(fy)))


def scene_by_name(name: str, n: int, width: int, height: int, sh_degree: int = 3, seed: int = 1002, K: int | None = None,
                  sigma_px: float | None = None) -> Scene:
    """`uniform` (make_scene) or `trained` (make_trained_like): what bench.py's --scene takes (`sigma_px`: the in-plane splat size
    in pixels, None = the generator's default — 3 for the uniform cloud, 4 for the trained-like scene)."""
    if name == "uniform":
        return make_scene(n, width, height, sh_degree, seed, K=K, **({} if sigma_px is None else {"sigma_px": sigma_px}))
    if name == "trained":
        return make_trained_like(n, width, height, sh_degree, seed, K=K, **({} if sigma_px is None else {"sigma_px": sigma_px}))
    raise ValueError(f"unknown scene kind {name!r}")
