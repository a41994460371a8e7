"""Pins the oracle's pipeline with the reference's known-answer / property tests
re-expressed (reference test/runtests.jl:486-520,
697-853) plus our own pin P1: a float64 dense autograd model for the image and
all five gradients.
"""
import math

import numpy as np
import pytest
import torch

import f64_model as fm
import scenes

DT = torch.float64


# K2 — runtests.jl:486-494
def test_tile_ranges(orc):
    keys = np.array([0 << 32, 0 << 32, 1 << 32, 2 << 32, 3 << 32], np.uint64)
    r = orc.identify_tile_range(keys, 4)
    assert r.tolist() == [[0, 2], [2, 3], [3, 4], [4, 5]]


def test_sort_is_stable_by_id(orc):
    rng = np.random.default_rng(3)
    keys = (rng.integers(0, 5, 1000).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 4, 1000).astype(np.uint64)
    vals = np.arange(1000, dtype=np.uint32)
    ks, vs = orc.sort_pairs(keys, vals)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ks, keys[order]) and np.array_equal(vs, vals[order])


# K13 — runtests.jl:697-742
def test_rgbdn_normal_channel(orc):
    sc, cam = scenes.grid_scene_rgbdn()
    st = orc.forward(sc["means"], sc["shs"], sc["opac"], sc["scales"], sc["rots"], cam, 0, mode="rgbdn")
    img = st.image
    assert img.shape == (48, 64, 8)
    alpha = img[:, :, 4]
    covered = alpha > 0.5
    assert covered.any()
    assert np.abs(img[:, :, 5]).max() < 1e-4
    assert np.abs(img[:, :, 6]).max() < 1e-4
    assert np.allclose(img[:, :, 7][covered], -alpha[covered], atol=1e-3)
    # the normal channel's cotangent must reach the rotations
    w = np.random.default_rng(0).standard_normal((48, 64, 3)).astype(np.float32)
    vp = np.zeros_like(img)
    vp[:, :, 5:8] = w
    g = orc.backward(st, vp, sc["means"], sc["shs"], sc["opac"], sc["scales"], sc["rots"], cam, 0)
    assert g.vrots.shape == sc["rots"].shape
    assert np.isfinite(g.vrots).all() and np.abs(g.vrots).max() > 0


# K14 — runtests.jl:760-797
def test_background_composite_identity(orc):
    sc, cam = scenes.sky_test_scene()
    bg = np.array([0.2, 0.7, 0.4], np.float32)

    def render(b):
        return orc.forward(sc["means"], sc["shs"], sc["opac"], sc["scales"], sc["rots"], cam, 0, background=b,
                           mode="rgbd").image

    in_kernel = render(bg)[:, :, :3]
    zeroed = render(np.zeros(3, np.float32))
    alpha = zeroed[:, :, 4]
    comp = zeroed[:, :, :3] + (1 - alpha)[:, :, None] * bg
    assert alpha.min() < 1e-3
    assert ((alpha > 0.05) & (alpha < 0.95)).any()
    assert alpha.max() > 0.3
    assert np.abs(in_kernel - comp).max() < 1e-5


# K15 — runtests.jl:799-832
def test_sky_dome_shell(orc):
    sc, cam = scenes.sky_dome_scene()
    st = orc.forward(sc["means"], sc["shs"], sc["opac"], sc["scales"], sc["rots"], cam, 0, mode="rgbd")
    alpha = st.image[:, :, 4]
    assert alpha.min() > 0.98
    opaque = alpha > 0.99
    assert opaque.any()
    for c, e in enumerate((0.2, 0.4, 0.9)):
        assert np.allclose(st.image[:, :, c][opaque], e, atol=1e-2)
    w = np.random.default_rng(1).standard_normal((48, 64, 3)).astype(np.float32)
    vp = np.zeros_like(st.image)
    vp[:, :, :3] = w
    g = orc.backward(st, vp, sc["means"], sc["shs"], sc["opac"], sc["scales"], sc["rots"], cam, 0)
    assert np.isfinite(g.vshs).all() and np.abs(g.vshs).max() > 0


# K16 — runtests.jl:43-77, 496-520
def conv_ssim(x, ref):
    g = torch.tensor([math.exp(-((i - 5) ** 2) / (2 * 1.5 ** 2)) for i in range(11)], dtype=DT)
    g = g / g.sum()
    w2 = (g[:, None] * g[None, :])
    ch = x.shape[1]
    w = w2.expand(ch, 1, 11, 11).contiguous()
    conv = lambda a: torch.nn.functional.conv2d(a, w, padding=5, groups=ch)  # noqa: E731
    mu1, mu2 = conv(x), conv(ref)
    s1 = conv(x * x) - mu1 ** 2
    s2 = conv(ref * ref) - mu2 ** 2
    s12 = conv(x * ref) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))).mean()


def test_ssim_known_answers(orc):
    ones = np.ones((1, 3, 16, 16), np.float32)
    zeros = np.zeros((1, 3, 16, 16), np.float32)
    assert abs(orc.ssim_forward(ones, zeros, train=False)[0].mean()) < 1e-4
    assert abs(orc.ssim_forward(ones, ones, train=False)[0].mean() - 1) < 1e-6
    x = np.zeros((1, 3, 16, 16), np.float32)  # numpy (B,C,H,W); Julia x[w, h, :, :]
    x[:, :, 0:4, 0:4] = 0.25
    x[:, :, 0:4, 4:8] = 0.5
    x[:, :, 12:16, 8:12] = 0.75
    x[:, :, 12:16, 12:16] = 1.0
    assert abs(orc.ssim_forward(x, ones, train=False)[0].mean() - 0.1035) < 1e-3


def test_ssim_matches_conv_value_and_gradient(orc):
    rng = np.random.default_rng(5)
    x = rng.uniform(size=(2, 3, 128, 128)).astype(np.float32)
    ref = rng.uniform(size=(2, 3, 128, 128)).astype(np.float32)
    m, d0, d1, d2 = orc.ssim_forward(x, ref, train=True)
    xt = torch.tensor(x, dtype=DT, requires_grad=True)
    y = conv_ssim(xt, torch.tensor(ref, dtype=DT))
    y.backward()
    assert abs(float(m.mean(dtype=np.float64)) - float(y.detach())) < 1e-5
    g = orc.ssim_backward(x, ref, np.full_like(x, 1.0 / x.size), d0, d1, d2)
    gt = xt.grad.numpy()
    assert np.linalg.norm(g - gt) <= 1e-3 * np.linalg.norm(gt)


def test_ssim_ragged_size(orc):
    """W,H not multiples of 16 (the kernel's partial tiles, fused_ssim.jl:202,385)."""
    rng = np.random.default_rng(6)
    x = rng.uniform(size=(1, 3, 37, 53)).astype(np.float32)
    ref = rng.uniform(size=(1, 3, 37, 53)).astype(np.float32)
    m = orc.ssim_forward(x, ref, train=False)[0]
    y = conv_ssim(torch.tensor(x, dtype=DT), torch.tensor(ref, dtype=DT))
    assert abs(float(m.mean(dtype=np.float64)) - float(y.detach())) < 1e-5


# P1 — float64 dense autograd model: image + all five gradients (+ pose gradient)
@pytest.mark.parametrize("mode,deg,seed", [("rgb", 3, 11), ("rgbd", 1, 12), ("rgbdn", 2, 13), ("rgb", 0, 14)])
def test_full_pipeline_vs_f64_autograd(orc, pkg, mode, deg, seed):
    W, H, n = 64, 48, 120
    s = pkg.synthetic.make_scene(n, W, H, deg, seed, sigma_px=4.0)
    R, t = pkg.synthetic.view_pose(2)
    cam = orc.Camera(W, H, s.focal, R=R, t=t)
    bg = np.array([0.3, 0.1, 0.6], np.float32)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, background=bg, mode=mode)
    assert st.n_rendered > 0
    C = st.image.shape[2]
    vp = np.random.default_rng(seed).standard_normal((H, W, C)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, background=bg,
                     pose_grad=True)
    tt = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=DT, requires_grad=True)  # noqa: E731
    means, shs, opac, scales, rots = tt(s.means), tt(s.shs), tt(s.opacities), tt(s.scales), tt(s.rotations)
    Rt, tv = tt(R), tt(t)
    img = fm.render_dense(means, shs, opac, scales, rots, cam, deg, bg, mode, st.values_sorted, st.ranges,
                          st.radii, R_w2c=Rt, t_w2c=tv)
    ref = img.detach().numpy()
    bad = np.abs(st.image - ref) > 1e-4 * np.maximum(1.0, np.abs(ref))
    assert bad.mean() < 1e-3, (bad.mean(), np.abs(st.image - ref).max())
    (img * torch.tensor(vp, dtype=DT)).sum().backward()
    vis = st.radii > 0

    def rel(a, b):
        return np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)

    # camera_center depends on R,t in the f64 model only through `cam` (constant), like the reference
    assert rel(g.vmeans[vis], means.grad.numpy()[vis]) < 2e-3
    assert rel(g.vshs, shs.grad.numpy()) < 2e-3
    assert rel(g.vopacities, opac.grad.numpy()) < 2e-3
    assert rel(g.vscales[vis], scales.grad.numpy()[vis]) < 2e-3
    assert rel(g.vrots[vis], rots.grad.numpy()[vis]) < 2e-3
    vR = g.vR.reshape(3, 3).T  # column-major -> row-major
    assert rel(vR, Rt.grad.numpy()) < 5e-3
    assert rel(g.vt, tv.grad.numpy()) < 5e-3
    # culled Gaussians get exact zeros (projection.jl:172-176)
    assert not g.vscales[~vis].any() and not g.vrots[~vis].any()


def test_empty_scene_returns_zero_image(orc, pkg):
    """rasterizer.jl:338: n_rendered == 0 -> all-zero image, background NOT applied."""
    s = pkg.synthetic.make_scene(50, 64, 48, 0, 1)
    means = s.means.copy()
    means[:, 2] = -5.0  # behind the camera
    cam = orc.Camera(64, 48, s.focal)
    st = orc.forward(means, s.shs, s.opacities, s.scales, s.rotations, cam, 0, background=(1, 1, 1))
    assert st.n_rendered == 0 and not st.image.any()
    g = orc.backward(st, np.ones((48, 64, 3), np.float32), means, s.shs, s.opacities, s.scales, s.rotations, cam, 0)
    assert not g.vmeans.any() and not g.vshs.any()


def test_partial_tiles_1080_rows(orc, pkg):
    """Height not a multiple of 16: the bottom half-tile row is masked."""
    W, H = 64, 40
    s = pkg.synthetic.make_scene(200, W, H, 1, 21, sigma_px=4.0)
    cam = orc.Camera(W, H, s.focal)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1)
    assert st.ranges.shape[0] == 4 * 3 and st.image.shape == (H, W, 3)
    assert st.n_contrib[32:, :].max() > 0


def test_morton_order_is_a_permutation_that_keeps_neighbours_together_and_the_image(pkg, orc):
    """synthetic.morton_order / reorder (bench.py --order morton): a permutation of the Gaussians; consecutive
    Gaussians end up close in space; the rendered image does not depend on the order (equal-depth ties aside)."""
    W, H, deg, n = 96, 64, 1, 700
    s = pkg.synthetic.make_scene(n, W, H, deg, 11, sigma_px=4.0)
    perm = pkg.synthetic.morton_order(s.means)
    assert sorted(perm.tolist()) == list(range(n))
    s2 = pkg.synthetic.reorder(s, perm)
    step = lambda m: np.linalg.norm(np.diff(m, axis=0), axis=1).mean()
    assert step(s2.means) < 0.4 * step(s.means)
    cam = orc.Camera(W, H, s.focal)
    a = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    b = orc.forward(s2.means, s2.shs, s2.opacities, s2.scales, s2.rotations, cam, deg)
    assert a.n_rendered == b.n_rendered
    np.testing.assert_allclose(a.image, b.image, rtol=0, atol=1e-5)


def test_parallel_truth_backward_equals_the_serial_one(orc, pkg):
    """oracle/gsr_oracle.c orc_render_bwd, deterministic = 2 (OpenMP tile loop, atomic adds on the DOUBLE accumulators —
    what makes the truth backward affordable at 5 M Gaussians / 4K) against deterministic = 1 (serial tile loop): every
    term is the same float, only the order of the double additions differs."""
    W, H, n, deg = 320, 240, 20_000, 2
    s = pkg.synthetic.make_scene(n, W, H, deg, 77, sigma_px=4.0)
    cam = orc.Camera(W, H, s.focal)
    for mode in ("rgb", "rgbd"):
        st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, mode=mode)
        C = st.image.shape[2]
        vp = pkg.synthetic.make_vpixels(W, H, C, 78) * 1e3
        a = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, deterministic=True)
        b = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, deterministic="parallel")
        for name in ("vmeans", "vshs", "vopacities", "vscales", "vrots", "vmeans2d", "vconics", "vfeatures"):
            x, y = getattr(a, name), getattr(b, name)
            assert np.abs(x).max() > 0
            # identical up to last-bit ties in the double -> float rounding (and what ∇project makes of them)
            assert np.allclose(x, y, rtol=2e-6, atol=1e-12 * np.abs(x).max()), (mode, name)
            assert (x != y).mean() < 1e-3, (mode, name)
