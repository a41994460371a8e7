"""Input scenes of the reference's integration tests, restated as data
(reference test/runtests.jl:697-758, 799-822; src/sky_dome.jl:57-71,120-141;
src/gaussians.jl:22-56,133-137).  No oracle / product imports."""
import math

import numpy as np

from oracle.oracle import Camera  # plain dataclass: camera fields only

SH0 = np.float32(0.28209479177387814)


def rgb_2_sh(c):
    return ((np.asarray(c, np.float32) - np.float32(0.5)) * (np.float32(1) / SH0)).astype(np.float32)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-np.float64(x)))).astype(np.float32) if isinstance(x, np.ndarray) else np.float32(
        1.0 / (1.0 + math.exp(-x)))


def _model(points, colors, scales_log, opacity_act):
    n = points.shape[0]
    rots = np.zeros((n, 4), np.float32)
    rots[:, 0] = 1.0
    shs = rgb_2_sh(colors).reshape(n, 1, 3)
    return dict(means=points.astype(np.float32), shs=shs, opac=np.full(n, opacity_act, np.float32),
                scales=np.exp(scales_log).astype(np.float32), rots=rots)


def grid_scene_rgbdn(seed=0):
    xs = np.linspace(-0.6, 0.6, 8, dtype=np.float32)
    pts = np.array([(x, y, 3.0) for y in xs for x in xs], np.float32)
    n = pts.shape[0]
    colors = np.random.default_rng(seed).uniform(size=(n, 3)).astype(np.float32)
    sl = np.tile(np.log(np.array([0.2, 0.2, 0.01], np.float32)), (n, 1))
    return _model(pts, colors, sl, sigmoid(5.0)), Camera(64, 48, (100.0, 100.0))


def sky_test_scene(seed=1, opacity=0.5):
    xs = np.linspace(-0.6, 0.6, 6, dtype=np.float32)
    pts = np.array([(x, y, 3.0) for y in xs for x in xs], np.float32)
    n = pts.shape[0]
    colors = np.random.default_rng(seed).uniform(size=(n, 3)).astype(np.float32)
    sl = np.full((n, 3), np.log(np.float32(0.1)), np.float32)
    # inverse_sigmoid then sigmoid round trip
    return _model(pts, colors, sl, sigmoid(math.log(opacity / (1 - opacity)))), Camera(64, 48, (100.0, 100.0))


def fibonacci_sphere(n):
    i = np.arange(1, n + 1, dtype=np.float32)
    ga = np.float32(math.pi * (3.0 - math.sqrt(5.0)))
    z = np.float32(1) - np.float32(2) * (i - np.float32(0.5)) / np.float32(n)
    r = np.sqrt(np.maximum(np.float32(1) - z * z, np.float32(0)))
    th = ga * (i - np.float32(1))
    return np.stack([r * np.cos(th), r * np.sin(th), z], 1).astype(np.float32), np.float32(
        math.sqrt(4 * math.pi / n))


def sky_dome_scene(n=8192, radius=50.0):
    dirs, spacing = fibonacci_sphere(n)
    pts = dirs * np.float32(radius)
    colors = np.tile(np.array([0.2, 0.4, 0.9], np.float32), (n, 1))
    sl = np.full((n, 3), np.log(np.float32(radius) * spacing), np.float32)
    cam = Camera(64, 48, (100.0, 100.0), far_plane=4 * radius)
    return _model(pts, colors, sl, sigmoid(math.log(0.99 / 0.01))), cam
