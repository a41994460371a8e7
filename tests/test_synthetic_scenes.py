"""CPU: the procedural scenes bench.py's extra_configs.scenes and tests/test_gpu_scenes.py render (synthetic.make_trained_like,
synthetic.add_skew, synthetic.scene_by_name) — deterministic, and with the statistics that make them what they claim to be (checked
with the oracle's forward)."""
import numpy as np
import pytest


def test_trained_like_generator_statistics(pkg, orc):
    """The generator does what its docstring says (these numbers are what makes the scene "trained-like")."""
    W, H, n = 1920, 1080, 200_000
    s = pkg.synthetic.make_trained_like(n, W, H, 3, 1010)
    s2 = pkg.synthetic.make_trained_like(n, W, H, 3, 1010)
    assert all(np.array_equal(getattr(s, k), getattr(s2, k)) for k in ("means", "scales_raw", "rotations", "opacities_raw", "shs"))
    assert s.means.shape == (n, 3) and s.shs.shape == (n, 16, 3) and s.means.dtype == np.float32
    sc = np.sort(s.scales, 1)
    flat = sc[:, 1] / sc[:, 0]
    assert np.median(flat) > 5.0, "flat splats: the thin axis is well below the in-plane ones"
    o = s.opacities
    assert (o > 0.8).mean() > 0.35 and (o < 0.2).mean() > 0.25 and ((o > 0.35) & (o < 0.65)).mean() < 0.15, "bimodal opacity"
    cam = orc.Camera(W, H, s.focal)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 3)
    z = s.means[:, 2]
    fx = float(s.focal[0])
    px, py = s.means[:, 0] / z * fx + W / 2, s.means[:, 1] / z * fx + H / 2
    inside = (z > 0.2) & (px > 0) & (px < W) & (py > 0) & (py < H)
    culled_inside = ((st.radii == 0) & inside).sum() / inside.sum()
    assert 0.2 <= culled_inside <= 0.45, ("~30 % of the on-screen splats are below radius_clip", culled_inside)
    ln = st.ranges[:, 1].astype(np.int64) - st.ranges[:, 0]
    assert ln.max() > 3.5 * ln.mean(), "tile lists are far from uniform"


def test_skewed_scenes_have_the_lists_they_promise(pkg, orc):
    W, H, n = 640, 360, 20_000
    base = pkg.synthetic.make_scene(n, W, H, 0, 7)
    hot = pkg.synthetic.add_skew(base, "hot:3000", 7)
    assert hot.n == n + 3000 and np.array_equal(hot.means[:n], base.means)
    cam = orc.Camera(W, H, base.focal)
    st = orc.forward(hot.means, hot.shs, hot.opacities, hot.scales, hot.rotations, cam, 0)
    ln = st.ranges[:, 1].astype(np.int64) - st.ranges[:, 0]
    gx = (W + 15) // 16
    assert int(np.argmax(ln)) == ((H + 15) // 16 // 2) * gx + gx // 2 and ln.max() > 2500
    dense = pkg.synthetic.add_skew(base, "dense:0.05:20", 7)
    st = orc.forward(dense.means, dense.shs, dense.opacities, dense.scales, dense.rotations, cam, 0)
    ln = st.ranges[:, 1].astype(np.int64) - st.ranges[:, 0]
    assert (ln > 5 * np.median(ln)).mean() > 0.02
    with pytest.raises(ValueError):
        pkg.synthetic.add_skew(base, "lumpy:3", 7)


def test_scene_by_name_and_sigma(pkg, orc):
    a = pkg.synthetic.scene_by_name("uniform", 1000, 320, 240, 1, 5)
    b = pkg.synthetic.make_scene(1000, 320, 240, 1, 5)
    assert np.array_equal(a.means, b.means) and np.array_equal(a.scales_raw, b.scales_raw)
    t4 = pkg.synthetic.scene_by_name("trained", 5000, 320, 240, 1, 5)
    t8 = pkg.synthetic.scene_by_name("trained", 5000, 320, 240, 1, 5, sigma_px=8.0)
    assert np.array_equal(t4.means[:100], pkg.synthetic.make_trained_like(5000, 320, 240, 1, 5).means[:100])
    assert np.exp(t8.scales_raw).max(1).mean() > 1.5 * np.exp(t4.scales_raw).max(1).mean()
    with pytest.raises(ValueError):
        pkg.synthetic.scene_by_name("fractal", 10, 64, 64)
