"""-m gpu parity tests: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.

Tolerances:
  * integer / index outputs (radii, tile rects, tile ranges, sorted ids): EXACT — the
    per-Gaussian kernels are compiled without FMA contraction and evaluate the same fp32
    expressions as the oracle;
  * per-Gaussian floats (means2d, conics, depths, rgbs): |Δ| <= 1e-6·|x| + 1e-7;
  * image / final T: |Δ| <= 1e-4 on >= 99.99 % of values (exp() differs by ~1 ulp, which
    can flip the α<1/255 and T<1e-4 decisions of single (pixel, splat) pairs);
  * n_contrib: <= 1e-3 of pixels may differ on tiny scenes;
  * gradients: ‖Δ‖₂/‖g‖₂ <= 1e-4 per tensor (fp32 atomics / reassociation against the
    oracle's double-precision deterministic accumulation).
"""
import ctypes as C

import os

import numpy as np
import pytest
import torch

import scenes
from hip_helpers import HipRun, blend_boundary_pixels, dev, frac_bad, rel_l2

pytestmark = pytest.mark.gpu


def _scene(pkg, orc, n, W, H, deg, seed, sigma_px=3.0, view=None):
    s = pkg.synthetic.make_scene(n, W, H, deg, seed, sigma_px=sigma_px)
    if view is None:
        cam = orc.Camera(W, H, s.focal)
    else:
        R, t = pkg.synthetic.view_pose(view)
        cam = orc.Camera(W, H, s.focal, R=R, t=t)
    return s, cam


def _compare_forward(st, run, img, opacities=None):
    geo = {k: v.cpu().numpy() for k, v in run.rast.geometry().items()}
    radii = run.rast.radii.cpu().numpy()
    assert np.array_equal(radii, st.radii), "radii must match exactly"
    vis = st.radii > 0
    assert run.rast.stats.n_visible == int(vis.sum())
    assert run.rast.stats.n_rendered == st.n_rendered
    for name, ref in (("means2d", st.means2d), ("conics", st.conics), ("depths", st.depths), ("rgbs", st.rgbs)):
        assert frac_bad(geo[name][vis], ref[vis], 1e-6, 1e-7) == 0.0, name
    cb = geo["clamped_bits"][vis]
    ref_bits = (st.clamped[vis].astype(np.int32) * np.array([1, 2, 4])).sum(1)
    assert np.array_equal(cb, ref_bits)
    rect = geo["rect"].astype(np.int64)
    tiles = (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])
    assert np.array_equal(tiles[vis], st.tiles_touched[vis]), "tile counts per Gaussian must match exactly"
    assert not st.tiles_touched[~vis].any()  # records of culled Gaussians are stale, as in the reference
    if st.normals is not None:
        assert frac_bad(geo["normals"][vis], st.normals[vis], 1e-6, 1e-7) == 0.0
    if st.n_rendered > 0:
        assert np.array_equal(run.rast.ranges.cpu().numpy().astype(np.uint32), st.ranges)
        assert np.array_equal(run.rast.values_sorted.cpu().numpy().astype(np.uint32), st.values_sorted)
    im = img.cpu().numpy()
    assert im.shape == st.image.shape
    T = run.rast.accum_alpha.cpu().numpy()
    nc = run.rast.n_contrib.cpu().numpy().astype(np.uint32)
    # (the last contributor moves by one where the saturation test T' < 1e-4 is decided by the last bits of T:
    #  one such pixel is already more than 1e-3 of a 25 x 32 image)
    sat = (nc != st.n_contrib).reshape(im.shape[:2])
    assert sat.sum() <= max(1, 1e-3 * nc.size)
    if opacities is not None and im.shape[0] * im.shape[1] < 20000:
        # (one pixel of a small image is already more than the 1e-4 outlier fraction: pixels with a pair ON the blend-test
        #  boundary, where the decision is the last bit of an exp, do not count — hip_helpers.blend_boundary_pixels — and
        #  neither does the at most one pixel whose walk stopped one entry apart, fuzz sweep case 14113)
        edge = blend_boundary_pixels(st, opacities, im.shape[1], im.shape[0])
        keep = ~(edge | sat)
        # a handful of pixels, not a tenth of the image (round-3 verdict / ADVICE: the measured worst case of 9 000 fuzzed
        # scenes was 17 of 1 634 pixels = 1.04 %, under 50 763 splats per pixel)
        assert (~keep).sum() <= max(4, 0.02 * keep.size), ((~keep).sum(), keep.size)
        assert frac_bad(im[keep], st.image[keep], 0.0, 1e-4) <= 1e-4
        assert frac_bad(T[keep], st.accum_alpha[keep], 0.0, 1e-4) <= 1e-4
        fmax = float(max(1.0, np.abs(st.image).max()))
        if edge.any():
            # ... and what is excluded may only differ by ONE flipped pair: its blend weight is alpha·T <= 1/255 (+ the
            # renormalisation of what lies behind it), times the largest feature value
            assert np.abs(im[edge] - st.image[edge]).max() <= 2.0 / 255.0 * fmax
            assert np.abs(T[edge] - st.accum_alpha[edge]).max() <= 2.0 / 255.0
        if (sat & ~edge).any():
            # ... or by the ONE entry blended on one side only: at most the transmittance the other side stopped at
            left = float((1.0 - np.minimum(T, st.accum_alpha.reshape(T.shape))[sat & ~edge]).max())
            assert np.abs(im[sat & ~edge] - st.image[sat & ~edge]).max() <= 2.0 * (left + 1e-4) * fmax
            assert np.abs(T[sat & ~edge] - st.accum_alpha.reshape(T.shape)[sat & ~edge]).max() <= 2.0 * (left + 1e-4)
    else:
        assert frac_bad(im, st.image, 0.0, 1e-4) <= 1e-4, np.abs(im - st.image).max()
        assert frac_bad(T, st.accum_alpha, 0.0, 1e-4) <= 1e-4


def _compare_backward(g, out, vis):
    vm, vs, vo, vsc, vr, vR, vt = [None if o is None else o.cpu().numpy() for o in out]
    assert rel_l2(vm, g.vmeans) <= 1e-4
    assert rel_l2(vs, g.vshs) <= 1e-4
    assert rel_l2(vo.reshape(-1), g.vopacities) <= 1e-4
    assert rel_l2(vsc, g.vscales) <= 1e-4
    assert rel_l2(vr, g.vrots) <= 1e-4
    # culled Gaussians: exact zeros
    assert not vm[~vis].any() and not vs[~vis].any() and not vsc[~vis].any() and not vr[~vis].any()
    return vR, vt


@pytest.mark.parametrize("mode,deg,seed,W,H,n", [
    ("rgb", 3, 101, 64, 48, 300), ("rgb", 0, 102, 64, 48, 300), ("rgbd", 1, 103, 80, 64, 400),
    ("rgbdn", 2, 104, 64, 48, 300), ("rgb", 3, 105, 200, 120, 3000), ("rgb", 2, 106, 64, 40, 500),
])
def test_forward_backward_vs_oracle(pkg, orc, mode, deg, seed, W, H, n):
    s, cam = _scene(pkg, orc, n, W, H, deg, seed, sigma_px=4.0, view=3)
    bg = (0.3, 0.1, 0.6)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, background=bg, mode=mode)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, bg, mode)
    img = run.forward()
    _compare_forward(st, run, img, s.opacities)
    C = st.image.shape[2]
    vp = np.random.default_rng(seed).standard_normal((H, W, C)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, background=bg)
    out = run.backward(vp)
    _compare_backward(g, out, st.radii > 0)
    assert rel_l2(run.rast.grad_means_2d.cpu().numpy(), g.vmeans2d) <= 1e-4


@pytest.mark.parametrize("case", range(12))
def test_randomised_sweep_vs_oracle(pkg, orc, case):
    """Seeded sweep over modes, SH degrees, ragged resolutions, views, footprint sizes and
    opacity ranges (both list modes): every stage against the oracle."""
    import fuzz_scenes
    fs = fuzz_scenes.sweep_scene(pkg, case)
    s, opac, cam, deg, mode, bg, rng = fs, fs.opac, fs.cam, fs.deg, fs.mode, fs.bg, fs.rng
    W, H = cam.width, cam.height
    st = orc.forward(s.means, s.shs, opac, s.scales, s.rotations, cam, deg, background=bg, mode=mode)
    run = HipRun(pkg, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, bg, mode)
    img = run.forward().clone()
    _compare_forward(st, run, img, opac)
    C = st.image.shape[2]
    vp = rng.standard_normal((H, W, C)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, background=bg)
    if st.n_rendered > 0 and np.linalg.norm(g.vmeans) > 0:
        _compare_backward(g, run.backward(vp), st.radii > 0)
    cul = HipRun(pkg, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, bg, mode, exact_tile_cull=True)
    assert torch.equal(cul.forward(), img)
    if st.n_rendered > 0 and np.linalg.norm(g.vmeans) > 0:
        _compare_backward(g, cul.backward(vp), st.radii > 0)


def test_degenerate_inputs(pkg, orc):
    """Zero / tiny opacities, Gaussians behind the camera or far off-screen, a zero Gaussian count."""
    W, H = 64, 48
    s = pkg.synthetic.make_scene(200, W, H, 1, 17, sigma_px=4.0)
    cam = orc.Camera(W, H, s.focal)
    opac = s.opacities.copy()
    opac[:50] = 0.0
    opac[50:100] = 1e-4
    means = s.means.copy()
    means[100:120, 2] *= -1.0
    means[120:140, 0] += 500.0
    st = orc.forward(means, s.shs, opac, s.scales, s.rotations, cam, 1, background=(0.5, 0.5, 0.5))
    for cull in (False, True):
        run = HipRun(pkg, means, s.shs, opac, s.scales, s.rotations, cam, 1, (0.5, 0.5, 0.5), exact_tile_cull=cull)
        img = run.forward()
        assert frac_bad(img.cpu().numpy(), st.image, 0, 1e-4) <= 1e-4
        vp = np.random.default_rng(1).standard_normal((H, W, 3)).astype(np.float32)
        g = orc.backward(st, vp, means, s.shs, opac, s.scales, s.rotations, cam, 1, background=(0.5, 0.5, 0.5))
        out = run.backward(vp)
        _compare_backward(g, out, st.radii > 0)
        assert all(torch.isfinite(o).all() for o in out[:5])
    # n = 0
    e = np.zeros((0, 3), np.float32)
    run0 = HipRun(pkg, e, np.zeros((0, 4, 3), np.float32), np.zeros((0,), np.float32), e, np.zeros((0, 4), np.float32),
                  cam, 1, (1, 0, 0))
    assert not run0.forward().cpu().numpy().any() and run0.rast.stats.n_rendered == 0


def test_k_padded_sh_storage(pkg, orc):
    """sh_degree below the stored band count (training ramps the degree, training.jl:577-585):
    bands above the active degree are ignored forward and get zero gradient."""
    s = pkg.synthetic.make_scene(300, 64, 48, 1, 7, sigma_px=4.0, K=16)
    cam = orc.Camera(64, 48, s.focal)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1)
    _compare_forward(st, run, run.forward())
    vp = pkg.synthetic.make_vpixels(64, 48, 3, 7) * 1e4
    g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1)
    out = run.backward(vp)
    _compare_backward(g, out, st.radii > 0)
    assert not out[1][:, 4:, :].any()


def test_pose_gradient_and_device_pose(pkg, orc):
    s, cam = _scene(pkg, orc, 300, 64, 48, 2, 33, sigma_px=4.0, view=1)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 2, mode="rgbd")
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 2, mode="rgbd", pose_dev=True)
    _compare_forward(st, run, run.forward())
    vp = np.random.default_rng(3).standard_normal((48, 64, 5)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 2, pose_grad=True)
    vR, vt = _compare_backward(g, run.backward(vp), st.radii > 0)
    # torch (3,3) row-major memory == the ABI's column-major 9 floats
    assert rel_l2(vR.reshape(-1), g.vR) <= 1e-4
    assert rel_l2(vt, g.vt) <= 1e-4


def test_covisibilities_and_uncertainties(pkg, orc):
    s, cam = _scene(pkg, orc, 400, 64, 48, 0, 44, sigma_px=4.0)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 0, want_covis=True, want_uncert=True)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 0, want_covis=True, want_uncert=True)
    _compare_forward(st, run, run.forward())
    assert (run.covis.cpu().numpy() != st.covisibilities).mean() <= 5e-3
    assert frac_bad(run.unc.cpu().numpy(), st.uncertainties, 0, 1e-4) <= 1e-4


def test_empty_scene(pkg, orc):
    """rasterizer.jl:338: nothing visible -> all-zero image (no background), zero gradients."""
    s = pkg.synthetic.make_scene(64, 64, 48, 0, 5)
    means = s.means.copy()
    means[:, 2] = -3.0
    cam = orc.Camera(64, 48, s.focal)
    run = HipRun(pkg, means, s.shs, s.opacities, s.scales, s.rotations, cam, 0, (1, 1, 1))
    img = run.forward()
    assert run.rast.stats.n_rendered == 0 and not img.cpu().numpy().any()
    out = run.backward(np.ones((48, 64, 3), np.float32))
    assert all(not o.cpu().numpy().any() for o in out[:5])


@pytest.mark.parametrize("n,floor", [(2500, 1024), (6000, 4096), (10000, 8192), (21000, 16384), (70000, 65536)])
def test_long_tile_lists_take_the_larger_sort_tiers(pkg, orc, n, floor):
    """Lists beyond 1024 / 4096 / 8192 instances in one tile: the 32 KB and 64 KB LDS sorts and, past
    8192, the chunked LDS sort + merge passes (1, 2 and 4 merge passes here) — sorted ids must stay exactly
    the oracle's."""
    rng = np.random.default_rng(9)
    s = pkg.synthetic.make_scene(n, 32, 32, 0, 9)
    means = np.stack([rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n), rng.uniform(2, 8, n)], 1).astype(np.float32)
    cam = orc.Camera(32, 32, s.focal)
    opac = np.full(n, 0.02, np.float32)
    st = orc.forward(means, s.shs, opac, s.scales * 3, s.rotations, cam, 0)
    assert (st.ranges[:, 1] - st.ranges[:, 0]).max() > floor
    run = HipRun(pkg, means, s.shs, opac, s.scales * 3, s.rotations, cam, 0)
    _compare_forward(st, run, run.forward())
    assert run.rast.stats.max_tile_instances > floor


def test_tile_bin_overflow_regrows_and_repeats(pkg, orc):
    """The per-tile key bins start at a capacity estimated from N / T.  A view that piles far more
    instances into one tile than that overflows them: the forward must notice (max count in the
    scan totals), grow the bins and repeat the pass — lists, ids and image still exact."""
    W, H, n = 640, 480, 3000   # T = 1200 tiles -> initial capacity 8*3000/1200 + 64 -> 128 keys per tile
    rng = np.random.default_rng(19)
    s = pkg.synthetic.make_scene(n, W, H, 1, 19)
    means = np.stack([rng.uniform(-0.02, 0.02, n), rng.uniform(-0.02, 0.02, n), rng.uniform(2, 8, n)], 1).astype(np.float32)
    cam = orc.Camera(W, H, s.focal)
    opac = np.full(n, 0.05, np.float32)
    st = orc.forward(means, s.shs, opac, s.scales, s.rotations, cam, 1)
    assert (st.ranges[:, 1] - st.ranges[:, 0]).max() > 1000
    run = HipRun(pkg, means, s.shs, opac, s.scales, s.rotations, cam, 1)
    _compare_forward(st, run, run.forward())
    assert run.rast.stats.max_tile_instances > 1000
    assert run.rast.stats.compact_binning == 1, "an overflowing view is finished in compact mode, nothing is repeated"
    _compare_forward(st, run, run.forward())
    assert run.rast.stats.compact_binning == 0, "the next view has bins of the right capacity"
    # and a second, sparse view on the same (grown) handle
    s2, _ = _scene(pkg, orc, n, W, H, 1, 20)
    st2 = orc.forward(s2.means, s2.shs, s2.opacities, s2.scales, s2.rotations, cam, 1)
    run.t = [dev(s2.means), dev(s2.shs), dev(s2.opacities.reshape(-1, 1)), dev(s2.scales), dev(s2.rotations)]
    _compare_forward(st2, run, run.forward())


@pytest.mark.parametrize("sigma_px", [4.0, 45.0])
@pytest.mark.parametrize("exact", [False, True])
def test_compact_binning_mode_is_bit_identical_to_the_bins(pkg, orc, exact, sigma_px):
    """gsr_config.bins_budget_bytes: with a budget the fixed-capacity bins cannot meet, every view is binned
    count -> scan -> scatter (compact mode).  Lists, ids, image and gradients must be bit-identical to the fast
    mode's (the per-tile sort erases the arrival order), and equal to the oracle's lists in reference-list mode."""
    # sigma_px = 45: footprints of well over 48 tiles — the wave-cooperative emit path of both binning modes
    W, H, n, deg = 320, 208, (12000 if sigma_px < 10 else 1500), 1
    s, cam = _scene(pkg, orc, n, W, H, deg, 61, sigma_px=sigma_px)
    fast = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, exact_tile_cull=exact)
    comp = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, exact_tile_cull=exact)
    comp.rast.close()
    comp.rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", exact_tile_cull=exact, bins_budget_bytes=1)
    vp = np.random.default_rng(3).standard_normal((H, W, 3)).astype(np.float32)
    for it in range(2):  # second view: steady state of both modes
        ia, ib = fast.forward().clone(), comp.forward().clone()
        # (a first view that overflows the initial capacity estimate is finished in compact mode, by design)
        assert (it == 0 or fast.rast.stats.compact_binning == 0) and comp.rast.stats.compact_binning == 1
        assert comp.rast.stats.bins_bytes == 8 * comp.rast.stats.n_rendered
        assert torch.equal(ia, ib)
        assert torch.equal(fast.rast.values_sorted, comp.rast.values_sorted) and torch.equal(fast.rast.ranges, comp.rast.ranges)
        ga, gb = fast.backward(vp), comp.backward(vp)
        assert all(torch.equal(x, y) for x, y in zip(ga[:5], gb[:5]))
    if not exact:
        st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
        assert np.array_equal(comp.rast.values_sorted.cpu().numpy().astype(np.uint32), st.values_sorted)


@pytest.mark.parametrize("form", ["direct", "aggregating"])
@pytest.mark.parametrize("exact", [False, True])
def test_overflow_tiles_keep_the_rest_of_the_view_in_its_bins(pkg, orc, exact, form):
    """Round 5: a budget that holds bins of 1536 keys while some lists are far longer (tiles of every tier: (1024, 4096],
    (4096, 8192], beyond).  The view stays in its bins (gsr_stats.compact_binning == 2): only the lists beyond the capacity are
    scattered a second time, every sort takes a tile's keys from where they are complete.  Lists, ids, image and gradients must
    be bit-identical to a handle whose bins hold everything and to the compact mode's, and the lists equal to the oracle's."""
    W, H, n, deg = 640, 416, 30000, 1
    s = pkg.synthetic.make_scene(n, W, H, deg, 67)
    for i, k in enumerate(("dense:0.01:70", "dense:0.005:190", "hot:9000")):   # ~2 000 / ~5 500 / 9 000 extra centres per tile
        s = pkg.synthetic.add_skew(s, k, seed=68 + i)
    cam = orc.Camera(W, H, s.focal)
    T = 40 * 26
    runs = [HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, exact_tile_cull=exact) for _ in range(3)]
    for r, budget in zip(runs, (0, (T + 1) * 8 * 1536, 1)):
        r.rast.close()
        r.rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", exact_tile_cull=exact, bins_budget_bytes=budget,
                                                   preprocess_form=form)
    full, some, comp = runs
    vp = np.random.default_rng(5).standard_normal((H, W, 3)).astype(np.float32)
    for it in range(3):
        imgs = [r.forward().clone() for r in runs]
        assert comp.rast.stats.compact_binning == 1 and (it == 0 or full.rast.stats.compact_binning == 0)
        # (the first view starts from a capacity estimate below 1024 keys and is finished in compact mode, by design)
        assert it == 0 or some.rast.stats.compact_binning == 2, some.rast.stats.compact_binning
        assert some.rast.stats.max_tile_instances > 8192
        assert it == 0 or some.rast.stats.bins_bytes == (T + 1) * 1536 * 8 + 8 * some.rast.stats.n_rendered
        for r in (some, comp):
            assert torch.equal(imgs[0], r.forward())
            assert torch.equal(full.rast.values_sorted, r.rast.values_sorted) and torch.equal(full.rast.ranges, r.rast.ranges)
        g = [r.backward(vp) for r in runs]
        for gb in g[1:]:
            assert all(torch.equal(x, y) for x, y in zip(g[0][:5], gb[:5]))
    lens = (full.rast.ranges[:, 1].long() - full.rast.ranges[:, 0].long()).cpu().numpy()
    assert ((lens > 1536) & (lens <= 4096)).any() and ((lens > 4096) & (lens <= 8192)).any() and (lens <= 1024).sum() > T // 2
    if not exact:
        st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
        assert np.array_equal(some.rast.values_sorted.cpu().numpy().astype(np.uint32), st.values_sorted)
        assert np.array_equal(some.rast.ranges.cpu().numpy().astype(np.uint32), st.ranges)


def test_views_with_and_without_tier_tiles_alternate_on_one_handle(pkg, orc):
    """gsr_forward holds its fused sort + forward launch when the PREVIOUS view had tiles of more than 1024 instances (their walk
    then runs beside it on the second stream) and sends it out early otherwise.  Every transition — early -> held, held -> held,
    held with nothing to wait for -> early — must give the images and gradients of a fresh handle, bit for bit."""
    W, H, deg = 320, 208, 1
    base = pkg.synthetic.make_scene(6000, W, H, deg, 81)
    scenes = [base, pkg.synthetic.add_skew(base, "hot:3000", seed=82), pkg.synthetic.add_skew(base, "hot:9500", seed=83)]
    cam = orc.Camera(W, H, base.focal)
    vp = np.random.default_rng(7).standard_normal((H, W, 3)).astype(np.float32)
    tensors = [[dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)] for s in scenes]
    fresh = []
    for s, t in zip(scenes, tensors):
        r = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, exact_tile_cull=True, want_covis=True,
                   want_uncert=True)
        img = r.forward().clone()
        fresh.append((img, [g.clone() for g in r.backward(vp)[:5]], int(r.rast.stats.max_tile_instances), r.covis.clone(),
                      r.unc.clone()))
        r.rast.close()
    assert fresh[0][2] <= 1024 < fresh[1][2] <= 4096 and fresh[2][2] > 8192
    one = HipRun(pkg, base.means, base.shs, base.opacities, base.scales, base.rotations, cam, deg, exact_tile_cull=True)
    one.unc = torch.zeros(H, W, device="cuda")
    # (1 -> 2 at its first occurrence: a held launch whose buffers turn out too small — it is not sent, the unfused path runs)
    for k in (0, 1, 1, 2, 0, 2, 2, 1, 0, 0, 2):
        one.t = tensors[k]
        one.covis = torch.zeros(scenes[k].means.shape[0], dtype=torch.uint8, device="cuda")   # (side outputs: written by both launches of a held view)
        img = one.forward()
        assert torch.equal(img, fresh[k][0]), k
        assert torch.equal(one.covis, fresh[k][3]) and torch.equal(one.unc, fresh[k][4]), k
        assert all(torch.equal(a, b) for a, b in zip(one.backward(vp)[:5], fresh[k][1])), k


def test_hot_tile_scene_stays_within_the_bins_budget(pkg, orc):
    """A skewed scene — one tile 100x deeper than the rest — must not cost O(tiles x longest list) memory: the bins are sized for
    the other tiles, the deep tile's keys are scattered again (8 B per instance), and the scratch stays bounded by the instance
    count."""
    W, H, deg = 1920, 1080, 0
    base = pkg.synthetic.make_scene(100_000, W, H, deg, 71)
    s = pkg.synthetic.add_skew(base, "hot:40000", seed=72)
    cam = orc.Camera(W, H, s.focal)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    for _ in range(2):
        img = run.forward()
    st_ = run.rast.stats
    T = 120 * 68
    assert st_.max_tile_instances > 20000 and st_.compact_binning == 2
    would_be = (T + 1) * int(st_.max_tile_instances * 1.25) * 8
    assert would_be > 2 * 2 ** 30, "fixed-capacity bins that hold the deep tile would need gigabytes here"
    assert st_.bins_bytes <= 100 * 2 ** 20 + 8 * st_.n_rendered, st_.bins_bytes
    assert run.rast.memory_usage() < 400 * 2 ** 20 + 300 * st_.n_rendered, run.rast.memory_usage()
    ref = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    assert frac_bad(img.cpu().numpy(), ref.image, 0, 1e-4) <= 1e-4
    assert np.array_equal(run.rast.radii.cpu().numpy(), ref.radii)


def test_footprint_masks_are_conservative(pkg, orc):
    """The per-instance row / quadrant masks only ever skip work: every pixel that passes the exact
    test (sigma >= 0, alpha >= 1/255; render.jl:92-95) must lie in a flagged row AND a flagged
    quadrant of its instance — for isotropic, elongated and large splats, both list modes."""
    W, H, n = 160, 96, 2500
    s = pkg.synthetic.make_scene(n, W, H, 0, 29, sigma_px=5.0)
    rng = np.random.default_rng(29)
    scales = (s.scales * np.exp(rng.normal(0, 0.8, (n, 3)))).astype(np.float32)  # strong anisotropy
    cam = orc.Camera(W, H, s.focal)
    ys, xs = np.mgrid[0:16, 0:16]
    gx = (W + 15) // 16
    for exact in (False, True):
        run = HipRun(pkg, s.means, s.shs, s.opacities, scales, s.rotations, cam, 0, exact_tile_cull=exact)
        run.forward()
        masks = run.rast.instance_masks.cpu().numpy().astype(np.int64)
        ids = run.rast.values_sorted.cpu().numpy()
        ranges = run.rast.ranges.cpu().numpy().reshape(-1, 2)
        geo = {k: v.cpu().numpy() for k, v in run.rast.geometry().items()}
        m2, con, op = geo["means2d"], geo["conics"], s.opacities
        checked = active_rows = flagged_rows = 0
        for t in range(ranges.shape[0]):
            X0, Y0 = (t % gx) * 16, (t // gx) * 16
            for p in range(ranges[t, 0], ranges[t, 1]):
                g, m = ids[p], masks[p]
                dx, dy = m2[g, 0] - (X0 + xs), m2[g, 1] - (Y0 + ys)
                sig = 0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) + con[g, 1] * dx * dy
                act = (sig >= 0) & (np.minimum(0.99, op[g] * np.exp(-sig)) >= 1.0 / 255.0)
                rows = act.any(1)
                assert not (rows & ~((m >> np.arange(16)) & 1).astype(bool)).any(), (t, p, hex(m))
                for q in range(4):
                    if act[8 * (q >> 1):8 * (q >> 1) + 8, 8 * (q & 1):8 * (q & 1) + 8].any():
                        assert (m >> (16 + q)) & 1, (t, p, q, hex(m))
                checked += 1; active_rows += int(rows.sum()); flagged_rows += bin(m & 0xFFFF).count("1")
        assert checked > 3000
        assert flagged_rows <= 1.25 * active_rows + 16  # and they are tight: few rows flagged beyond the active ones


def test_large_footprints_and_deterministic_gradients(pkg, orc):
    """Splats covering hundreds of tiles take the wave-cooperative row-sum path of the
    per-Gaussian backward; gradients are summed in a fixed order, so two runs agree bit for bit."""
    W, H, n = 320, 240, 400
    s = pkg.synthetic.make_scene(n, W, H, 1, 91, sigma_px=4.0)
    scales = s.scales.copy()
    scales[:40] *= 25.0  # radius of a few hundred pixels -> area >> 48 tiles
    opac = s.opacities.copy()
    opac[:40] = 0.05
    cam = orc.Camera(W, H, s.focal)
    st = orc.forward(s.means, s.shs, opac, scales, s.rotations, cam, 1, mode="rgbd")
    assert st.tiles_touched.max() > 200
    run = HipRun(pkg, s.means, s.shs, opac, scales, s.rotations, cam, 1, mode="rgbd")
    _compare_forward(st, run, run.forward())
    vp = np.random.default_rng(4).standard_normal((H, W, 5)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, opac, scales, s.rotations, cam, 1)
    out1 = [o.clone() for o in run.backward(vp)[:5]]
    _compare_backward(g, out1 + [None, None], st.radii > 0)
    out2 = run.backward(vp)[:5]
    for a, b in zip(out1, out2):
        assert torch.equal(a, b), "backward must be bit-deterministic"


@pytest.mark.parametrize("mode,deg,seed,W,H,n", [("rgb", 3, 301, 200, 120, 3000), ("rgbdn", 1, 302, 96, 80, 800)])
def test_exact_tile_cull_changes_lists_not_results(pkg, orc, mode, deg, seed, W, H, n):
    """GSR_FLAG_EXACT_TILE_CULL: instances that cannot reach alpha >= 1/255 anywhere in their tile
    are not emitted.  Image and final T are BIT-identical to the reference-list mode, gradients
    agree with the oracle, every culled tile list is a subsequence of the reference's."""
    s, cam = _scene(pkg, orc, n, W, H, deg, seed, sigma_px=4.0, view=5)
    opac = s.opacities.copy()
    opac[::3] *= 0.05  # plenty of faint splats: small effective footprints
    bg = (0.1, 0.2, 0.3)
    st = orc.forward(s.means, s.shs, opac, s.scales, s.rotations, cam, deg, background=bg, mode=mode)
    ref = HipRun(pkg, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, bg, mode)
    cul = HipRun(pkg, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, bg, mode, exact_tile_cull=True)
    img_r, img_c = ref.forward().clone(), cul.forward().clone()
    assert torch.equal(img_r, img_c)
    assert torch.equal(ref.rast.accum_alpha, cul.rast.accum_alpha)
    Dr, Dc = ref.rast.stats.n_rendered, cul.rast.stats.n_rendered
    assert Dc < Dr == st.n_rendered
    rr, rc = ref.rast.ranges.cpu().numpy(), cul.rast.ranges.cpu().numpy()
    vr, vc = ref.rast.values_sorted.cpu().numpy(), cul.rast.values_sorted.cpu().numpy()
    for t in range(rr.shape[0]):
        a, b = vr[rr[t, 0]:rr[t, 1]], vc[rc[t, 0]:rc[t, 1]]
        keep = np.isin(a, b)
        assert np.array_equal(a[keep], b), "culled list must be an order-preserving subsequence"
    C = st.image.shape[2]
    vp = np.random.default_rng(seed).standard_normal((H, W, C)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, background=bg)
    out = cul.backward(vp)
    _compare_backward(g, out, st.radii > 0)
    assert rel_l2(cul.rast.grad_means_2d.cpu().numpy(), g.vmeans2d) <= 1e-4


# ---- the reference's own integration scenes (K13-K15) through the HIP path ----
def test_k13_rgbdn_grid_scene(pkg, orc):
    sc, cam = scenes.grid_scene_rgbdn()
    run = HipRun(pkg, sc["means"], sc["shs"], sc["opac"], sc["scales"], sc["rots"], cam, 0, mode="rgbdn")
    img = run.forward().cpu().numpy()
    assert img.shape == (48, 64, 8)
    alpha = img[:, :, 4]
    cov = alpha > 0.5
    assert cov.any() and np.abs(img[:, :, 5]).max() < 1e-4 and np.abs(img[:, :, 6]).max() < 1e-4
    assert np.allclose(img[:, :, 7][cov], -alpha[cov], atol=1e-3)
    vp = np.zeros_like(img)
    vp[:, :, 5:8] = np.random.default_rng(0).standard_normal((48, 64, 3))
    vrot = run.backward(vp)[4].cpu().numpy()
    assert np.isfinite(vrot).all() and np.abs(vrot).max() > 0


def test_k14_background_identity(pkg, orc):
    sc, cam = scenes.sky_test_scene()
    bg = np.array([0.2, 0.7, 0.4], np.float32)

    def render(b):
        return HipRun(pkg, sc["means"], sc["shs"], sc["opac"], sc["scales"], sc["rots"], cam, 0, b,
                      "rgbd").forward().cpu().numpy().copy()

    in_kernel = render(bg)[:, :, :3]
    zeroed = render((0, 0, 0))
    alpha = zeroed[:, :, 4]
    assert alpha.min() < 1e-3 and ((alpha > 0.05) & (alpha < 0.95)).any() and alpha.max() > 0.3
    assert np.abs(in_kernel - (zeroed[:, :, :3] + (1 - alpha)[:, :, None] * bg)).max() < 1e-5


def test_k15_sky_dome(pkg, orc):
    sc, cam = scenes.sky_dome_scene()
    run = HipRun(pkg, sc["means"], sc["shs"], sc["opac"], sc["scales"], sc["rots"], cam, 0, mode="rgbd")
    img = run.forward().cpu().numpy()
    alpha = img[:, :, 4]
    assert alpha.min() > 0.98
    op = alpha > 0.99
    for c, e in enumerate((0.2, 0.4, 0.9)):
        assert np.allclose(img[:, :, c][op], e, atol=1e-2)


# ---- SSIM + loss head ----
@pytest.mark.parametrize("shape", [(2, 3, 128, 128), (1, 3, 37, 53), (1, 1, 16, 16)])
def test_ssim_vs_oracle(pkg, orc, shape):
    """The exact twin of ssim.hip (gsr_ssim_precision(1)): bit for bit the oracle's maps and pullback."""
    rng = np.random.default_rng(5)
    x = rng.uniform(size=shape).astype(np.float32)
    y = rng.uniform(size=shape).astype(np.float32)
    m, d0, d1, d2 = orc.ssim_forward(x, y, train=True)
    F = pkg.fused_ssim
    with F.exact_arithmetic():
        hm, h0, h1, h2 = F._fused_ssim(dev(x), dev(y), train=True)
        for a, b in ((hm, m), (h0, d0), (h1, d1), (h2, d2)):
            assert np.array_equal(a.cpu().numpy(), b), "SSIM maps are bit-exact (no FMA contraction)"
        dl = rng.standard_normal(shape).astype(np.float32)
        g = orc.ssim_backward(x, y, dl, d0, d1, d2)
        hg = F.fused_ssim_bwd(dev(x), dev(y), dev(dl), h0, h1, h2)
        assert np.array_equal(hg.cpu().numpy(), g)


@pytest.mark.parametrize("shape", [(2, 3, 128, 128), (1, 3, 37, 53), (1, 1, 16, 16)])
def test_default_ssim_arithmetic_vs_oracle_at_fp32_tolerance(pkg, orc, shape):
    """The library's DEFAULT SSIM path: multiply-adds contracted, the formula's six divisions over two reciprocals (what a GPU
    compiler makes of fused_ssim.jl).  Against the oracle's fp32-as-written evaluation: the SSIM map to
    1e-5 absolute (its values lie in [-1, 1]; sigma² = E[x²] - mu² cancels, so an ulp of a moment is several ulps of the
    quotient; mean 5e-7), the derivative maps and the pullback to 2e-5 of their L2 norm and element-wise to 1e-4 relative +
    1e-5 of the map's scale on 99.9 % of the values."""
    rng = np.random.default_rng(5)
    x = rng.uniform(size=shape).astype(np.float32)
    y = rng.uniform(size=shape).astype(np.float32)
    m, d0, d1, d2 = orc.ssim_forward(x, y, train=True)
    F = pkg.fused_ssim
    hm, h0, h1, h2 = F._fused_ssim(dev(x), dev(y), train=True)
    dm = np.abs(hm.cpu().numpy() - m)
    print("ssim map max / mean |diff|:", dm.max(), dm.mean())
    assert dm.max() <= 1e-5 and dm.mean() <= 5e-7
    for a, b in ((h0, d0), (h1, d1), (h2, d2)):
        a = a.cpu().numpy()
        print("   derivative map rel-L2:", rel_l2(a, b), "frac bad:", frac_bad(a, b, 1e-4, 1e-5 * float(np.abs(b).max())))
        assert rel_l2(a, b) <= 2e-5
        assert frac_bad(a, b, 1e-4, 1e-5 * float(np.abs(b).max())) <= 1e-3
    dl = rng.standard_normal(shape).astype(np.float32)
    g = orc.ssim_backward(x, y, dl, d0, d1, d2)
    hg = F.fused_ssim_bwd(dev(x), dev(y), dev(dl), h0, h1, h2).cpu().numpy()   # (its own derivative maps: the whole chain)
    print("   pullback rel-L2:", rel_l2(hg, g))
    assert rel_l2(hg, g) <= 2e-5
    assert frac_bad(hg, g, 1e-4, 1e-5 * float(np.abs(g).max())) <= 1e-3
    assert not np.array_equal(hm.cpu().numpy(), m) or shape[-1] <= 16  # (it IS the other build: not bit-identical at size)


def test_ssim_known_answers_and_autograd(pkg):
    F = pkg.fused_ssim
    ones, zeros = torch.ones(1, 3, 16, 16).cuda(), torch.zeros(1, 3, 16, 16).cuda()
    assert abs(float(F.fused_ssim(ones, zeros).mean())) < 1e-4
    assert abs(float(F.fused_ssim(ones, ones).mean()) - 1) < 1e-5   # (runtests.jl:496-520 asks for ≈, i.e. 3e-4)
    x = torch.zeros(1, 3, 16, 16)
    x[:, :, 0:4, 0:4] = 0.25; x[:, :, 0:4, 4:8] = 0.5; x[:, :, 12:16, 8:12] = 0.75; x[:, :, 12:16, 12:16] = 1.0
    assert abs(float(F.fused_ssim(x.cuda(), ones).mean()) - 0.1035) < 1e-3
    xr = torch.rand(2, 3, 64, 64).cuda().requires_grad_(True)
    ref = torch.rand(2, 3, 64, 64).cuda()
    F.fused_ssim(xr, ref).mean().backward()
    assert xr.grad is not None and torch.isfinite(xr.grad).all() and float(xr.grad.abs().max()) > 0


@pytest.mark.parametrize("mode", ["rgb", "rgbd"])
def test_loss_head_vs_oracle(pkg, orc, mode):
    W, H = 72, 40
    s, cam = _scene(pkg, orc, 500, W, H, 1, 77, sigma_px=4.0)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1, mode=mode)
    tgt = pkg.synthetic.make_target(W, H, 77)
    loss, vp = orc.loss_head(st.image, tgt)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1, mode=mode)
    # feed the oracle's image so only the loss head is compared; default (contracted) arithmetic at tolerance, then the exact
    # twin, whose pullback is the oracle's bit for bit
    hl, hv = pkg.fused_ssim.l1_ssim_loss(run.rast, dev(st.image), dev(tgt))
    torch.cuda.synchronize()
    assert abs(float(hl) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    assert rel_l2(hv.cpu().numpy(), vp) <= 1e-5
    assert not hv.cpu().numpy()[:, :, 3:].any()
    with pkg.fused_ssim.exact_arithmetic():
        hl2, hv2 = pkg.fused_ssim.l1_ssim_loss(run.rast, dev(st.image), dev(tgt))
        torch.cuda.synchronize()
    assert abs(float(hl2) - float(loss)) <= 1e-6 and np.array_equal(hv2.cpu().numpy(), vp)


@pytest.mark.parametrize("bg", [(0.0, 0.0, 0.0), (0.2, 0.5, 0.1)])
@pytest.mark.parametrize("mode", ["rgbd", "rgbdn", "rgb"])
def test_color_cotangent_flag_gives_the_unflagged_gradients(pkg, orc, mode, bg):
    """GSR_GRADS_COLOR_COTANGENT: the caller says that channels >= 3 of vpixels are zeros (the loss head's cotangent) and the
    backward of :rgbd / :rgbdn runs the :rgb arithmetic on the mode's stream.  Same gradients as the unflagged call up to the
    association of fp32 sums, the oracle's within the suite's tolerance; lists beyond 1024 instances included (their launch
    keeps the full kernel); ignored in :rgb mode; unknown flag bits are refused.  ABI 6: the promise is CHECKED —
    the flag is honoured only for the very buffer this handle's gsr_loss_l1_ssim wrote for this forward; a copy of it, or the
    same buffer after another forward, is GSR_E_INVALID_ARG; GSR_CHECK_COLOR_COTANGENT=1 also looks INTO the buffer."""
    W, H, deg = 200, 120, 2
    base = pkg.synthetic.make_scene(4000, W, H, deg, 91, sigma_px=5.0)
    s = pkg.synthetic.add_skew(base, "hot:2500", seed=92)
    cam = orc.Camera(W, H, s.focal)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, background=bg, mode=mode)
    tgt = pkg.synthetic.make_target(W, H, 93)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, bg, mode)
    img = run.forward()
    _, vp = pkg.fused_ssim.l1_ssim_loss(run.rast, img, dev(tgt))   # (the loss head's own buffer: what the flag is valid for)
    assert mode == "rgb" or not vp[:, :, 3:].any()
    args = (vp, *run.t, run.camera, deg, run.bg)
    plain = [g.clone() for g in run.rast.backward_raw(*args)[:5]]
    flagged = [g.clone() for g in run.rast.backward_raw(*args, color_cotangent=True)[:5]]
    torch.cuda.synchronize()
    if mode != "rgb":
        # any other buffer with the flag set is refused — even a bit-identical copy: the library cannot know what was added to it
        with pytest.raises(pkg._lib.GsrError, match="only valid for the cotangent gsr_loss_l1_ssim wrote"):
            run.rast.backward_raw(vp.clone(), *args[1:], color_cotangent=True)
        # the debug check looks into the buffer: something added in place to the depth channel after the loss head
        os.environ["GSR_CHECK_COLOR_COTANGENT"] = "1"
        try:
            run.rast.backward_raw(*args, color_cotangent=True)       # untouched: passes
            vp[3, 5, 3] += 1e-3
            with pytest.raises(pkg._lib.GsrError, match="non-zero values above the colour channels"):
                run.rast.backward_raw(*args, color_cotangent=True)
            vp[3, 5, 3] = 0.0
        finally:
            del os.environ["GSR_CHECK_COLOR_COTANGENT"]
    else:
        run.rast.backward_raw(vp.clone(), *args[1:], color_cotangent=True)   # :rgb: nothing above the colour, the flag says nothing
    for a, b in zip(plain, flagged):
        assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) <= (0.0 if mode == "rgb" else 2e-6)
    g = orc.backward(st, vp.cpu().numpy(), s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, background=bg)
    _compare_backward(g, flagged + [None, None], st.radii > 0)
    assert int(run.rast.stats.max_tile_instances) > 1024
    L = pkg._lib
    gr = L.Grads(*(t.data_ptr() for t in plain), None, None, None, None, 0, 0x2, 0)
    inp = run.rast._inputs(*run.t, deg, run.bg)
    cs = run.rast._camera(run.camera, None, None)
    assert L.load().gsr_backward(run.rast._h, C.byref(inp), C.byref(cs), vp.data_ptr(), C.byref(gr), None) == L.GSR_E_INVALID_ARG
    if mode != "rgb":
        # ... and the loss head's buffer is only good for the forward it was computed from
        run.forward()
        with pytest.raises(pkg._lib.GsrError, match="only valid for the cotangent"):
            run.rast.backward_raw(*args, color_cotangent=True)


def test_functor_autograd_end_to_end(pkg, orc):
    """rast(points, opacities, scales, rotations, f_dc, f_rest; camera, sh_degree) under autograd:
    raw-parameter gradients = activated gradients x activation derivatives (rasterizer.jl:200-253)."""
    W, H, deg = 64, 48, 2
    s, cam = _scene(pkg, orc, 300, W, H, deg, 55, sigma_px=4.0)
    camera = pkg.Camera(W, H, tuple(s.focal))
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb")
    p = [dev(s.means), dev(s.opacities_raw.reshape(-1, 1)), dev(s.scales_raw), dev(s.rotations), dev(s.shs[:, :1]),
         dev(s.shs[:, 1:])]
    for t in p:
        t.requires_grad_(True)
    img = rast(*p, camera=camera, sh_degree=deg)
    w = dev(np.random.default_rng(1).standard_normal((H, W, 3)))
    (img * w).sum().backward()
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    g = orc.backward(st, w.cpu().numpy(), s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    o = s.opacities.astype(np.float64)
    assert rel_l2(p[0].grad.cpu().numpy(), g.vmeans) <= 1e-4
    assert rel_l2(p[1].grad.cpu().numpy().reshape(-1), g.vopacities * o * (1 - o)) <= 1e-4
    assert rel_l2(p[2].grad.cpu().numpy(), g.vscales * s.scales) <= 1e-4
    assert rel_l2(p[3].grad.cpu().numpy(), g.vrots) <= 1e-4
    assert rel_l2(p[4].grad.cpu().numpy(), g.vshs[:, :1]) <= 1e-4
    assert rel_l2(p[5].grad.cpu().numpy(), g.vshs[:, 1:]) <= 1e-4


def test_factored_sh_gradient_exchange_form(pkg, orc):
    """backward_raw(factored_sh=True) returns the colour cotangent vc instead of ∇shs;
    sh_grad_from_views rebuilds ∇shs — bit-identically for one view, and for a batch of views it
    equals the sum of the per-view ∇shs (what the all-reduce of the full arena would give)."""
    W, H, deg, n, V = 96, 64, 3, 800, 3
    s = pkg.synthetic.make_scene(n, W, H, deg, 77, sigma_px=4.0)
    K = s.shs.shape[1]
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb")
    p = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
    full_sum, vcs, centers, small_sum = None, [], [], None
    for v in range(V):
        R, t = pkg.synthetic.view_pose(v, V)
        cam = pkg.Camera(W, H, tuple(s.focal), (0.5, 0.5), R, t)
        vp = dev(pkg.synthetic.make_vpixels(W, H, 3, 300 + v) * 1e3)
        rast.forward_raw(*p, cam, deg, (0, 0, 0))
        full = [x.clone() for x in rast.backward_raw(vp, *p, cam, deg, (0, 0, 0))[:5]]
        rast.forward_raw(*p, cam, deg, (0, 0, 0))
        fac = [x.clone() for x in rast.backward_raw(vp, *p, cam, deg, (0, 0, 0), factored_sh=True)[:5]]
        for k in (0, 2, 3, 4):  # vmeans, vopacities, vscales, vrot: untouched by the form
            assert torch.equal(full[k], fac[k])
        assert fac[1].shape == (n, 3)
        one = pkg.rasterizer.sh_grad_from_views(p[0], fac[1].view(1, n, 3), dev(cam.camera_center.reshape(1, 3)), K, deg)
        assert torch.equal(one, full[1]), "V = 1 must reproduce the fused ∇shs bit for bit"
        vcs.append(fac[1]); centers.append(cam.camera_center)
        full_sum = full[1].double() if full_sum is None else full_sum + full[1].double()
    got = pkg.rasterizer.sh_grad_from_views(p[0], torch.stack(vcs).contiguous(), dev(np.stack(centers)), K, deg)
    torch.cuda.synchronize()
    assert full_sum.abs().max() > 0
    assert rel_l2(got.cpu().numpy(), full_sum.cpu().numpy()) <= 1e-6
    # and against the oracle's ∇SH for the first view
    R, t = pkg.synthetic.view_pose(0, V)
    ocam = orc.Camera(W, H, s.focal, R=R, t=t)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, ocam, deg)
    g = orc.backward(st, pkg.synthetic.make_vpixels(W, H, 3, 300) * 1e3, s.means, s.shs, s.opacities, s.scales,
                     s.rotations, ocam, deg)
    vc0 = g.vfeatures[:, :3] * (1.0 - st.clamped.astype(np.float32))
    assert rel_l2(vcs[0].cpu().numpy(), vc0) <= 1e-4


def test_update_stats_vs_oracle(pkg, orc):
    """strategy.jl:107-136 `_update_stats!` on the side outputs of a forward/backward pair."""
    s, cam = _scene(pkg, orc, 500, 96, 64, 1, 61, sigma_px=4.0)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1)
    run.forward()
    rng = np.random.default_rng(2)
    mr = rng.integers(0, 30, 500).astype(np.int32)
    acc = rng.uniform(size=500).astype(np.float32)
    den = rng.integers(0, 5, 500).astype(np.float32)
    d_mr, d_acc, d_den = dev(mr, torch.int32), dev(acc), dev(den)
    with pytest.raises(pkg._lib.GsrError):
        run.rast.update_stats(d_mr, d_acc, d_den)  # no backward yet
    vp = np.random.default_rng(3).standard_normal((64, 96, 3)).astype(np.float32)
    run.backward(vp)
    run.rast.update_stats(d_mr, d_acc, d_den)
    vm2 = run.rast.grad_means_2d.cpu().numpy()
    orc.update_stats(mr, acc, den, st.radii, vm2, 96, 64)
    assert np.array_equal(d_mr.cpu().numpy(), mr)
    assert np.array_equal(d_den.cpu().numpy(), den)
    assert np.allclose(d_acc.cpu().numpy(), acc, rtol=1e-6, atol=1e-7)


def test_side_stream_two_handles_and_buffer_growth(pkg, orc):
    """Work is enqueued on the caller's stream (not the null stream); handles are independent;
    scratch grows when N / D grow between calls."""
    W, H = 96, 64
    side = torch.cuda.Stream()
    scenes_ = [pkg.synthetic.make_scene(n, W, H, 1, 70 + i, sigma_px=sp) for i, (n, sp) in
               enumerate([(100, 3.0), (900, 6.0), (300, 4.0)])]
    cam_o = orc.Camera(W, H, scenes_[0].focal)
    cam = pkg.Camera(W, H, tuple(scenes_[0].focal))
    ra = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", exact_tile_cull=False)
    rb = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", exact_tile_cull=True)
    vp = np.random.default_rng(5).standard_normal((H, W, 3)).astype(np.float32)
    with torch.cuda.stream(side):
        for s in scenes_:
            t = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
            ia = ra.forward_raw(*t, cam, 1, (0, 0, 0)).clone()
            ib = rb.forward_raw(*t, cam, 1, (0, 0, 0)).clone()  # interleaved second handle
            ga = ra.backward_raw(dev(vp), *t, cam, 1, (0, 0, 0))
            gb = rb.backward_raw(dev(vp), *t, cam, 1, (0, 0, 0))
            side.synchronize()
            st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam_o, 1)
            g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam_o, 1)
            assert torch.equal(ia, ib)
            assert frac_bad(ia.cpu().numpy(), st.image, 0, 1e-4) <= 1e-4
            _compare_backward(g, ga, st.radii > 0)
            _compare_backward(g, gb, st.radii > 0)
            assert ra.stats.n_rendered == st.n_rendered
    ra.profile(True)
    with torch.cuda.stream(side):
        ra.forward_raw(*t, cam, 1, (0, 0, 0))
        ra.backward_raw(dev(vp), *t, cam, 1, (0, 0, 0))
    prof = ra.profile_read()
    # (the forward of the tiles is the fused sort + forward launch in the common case)
    assert prof["sort_composite_fwd"][1] + prof["composite_fwd"][1] >= 1 and prof["sort_composite_fwd"][0] + prof["composite_fwd"][0] > 0
    assert prof["composite_bwd"][1] == 1 and prof["composite_bwd"][0] > 0
    ra.profile(False)


def test_reserve_makes_the_following_forwards_allocation_free(pkg):
    """gsr_reserve (ABI 6): after reserving for a model and an instance count, a forward of a scene inside those sizes neither
    reallocates a buffer nor changes memory_usage, and its image equals a cold handle's bit for bit."""
    small, big = pkg.synthetic.make_scene(3000, 320, 192, 0, 5), pkg.synthetic.make_scene(40000, 320, 192, 0, 6)
    cam = pkg.Camera(320, 192, tuple(big.focal))
    tens = lambda s: [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
    cold = pkg.rasterizer.GaussianRasterizer(320, 192, mode="rgbd")
    ref = cold.forward_raw(*tens(big), cam, 0, (0, 0, 0)).clone()
    d_big = int(cold.stats.n_rendered)
    rast = pkg.rasterizer.GaussianRasterizer(320, 192, mode="rgbd")
    rast.forward_raw(*tens(small), cam, 0, (0, 0, 0))
    g0 = int(rast.stats.scratch_regrowths)
    rast.reserve(40000, d_big)
    mem = rast.memory_usage()
    img = rast.forward_raw(*tens(big), cam, 0, (0, 0, 0))
    assert torch.equal(img, ref)
    assert int(rast.stats.scratch_regrowths) == g0, (g0, int(rast.stats.scratch_regrowths))
    bins_bytes = lambda r: 8 * int(r.stats.bin_capacity) * ((320 + 15) // 16) * ((192 + 15) // 16)
    # (the key bins are a policy of their own — and so is the compact binning a view falls back to when they overflowed, which this
    #  first large view after a small one does: its sort buffers are first allocations, not regrowths)
    assert rast.memory_usage() <= mem + bins_bytes(rast) + 40 * d_big, (rast.memory_usage(), mem, d_big)
    # ... and a backward after it works on the reserved buffers
    vp = torch.ones_like(img)
    out = rast.backward_raw(vp, *tens(big), cam, 0, (0, 0, 0))
    want = cold.backward_raw(vp, *tens(big), cam, 0, (0, 0, 0))
    for a, b in zip(out[:5], want[:5]):
        assert torch.equal(a, b)
    with pytest.raises(pkg._lib.GsrError):
        rast.reserve(-1, 0)
    rast.reserve(0, 0)   # no-op
    rast.close(); cold.close()


def test_state_errors(pkg):
    """gsr_backward without a matching forward -> GSR_E_STATE; bad shapes -> ValueError."""
    s = pkg.synthetic.make_scene(32, 64, 48, 0, 3)
    cam = pkg.Camera(64, 48, tuple(s.focal))
    rast = pkg.rasterizer.GaussianRasterizer(64, 48, mode="rgb")
    t = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
    with pytest.raises(pkg._lib.GsrError) as e:
        rast.backward_raw(torch.zeros(48, 64, 3).cuda(), *t, cam, 0, (0, 0, 0))
    assert e.value.code == pkg._lib.GSR_E_STATE
    with pytest.raises(ValueError):
        rast.forward_raw(t[0][:, :2].contiguous(), *t[1:], cam, 0, (0, 0, 0))
    with pytest.raises(pkg._lib.GsrError):
        rast.forward_raw(*t, cam, 2, (0, 0, 0))  # K=1 cannot hold degree 2
    assert rast.memory_usage() > 0
    rast.release_scene_buffers()
    rast.forward_raw(*t, cam, 0, (0, 0, 0))
    # forward generation: a backward that names a superseded forward is refused (an eval render between a
    # training forward and its pullback must not silently yield the wrong view's gradients)
    gen = int(rast.stats.generation)
    rast.forward_raw(*t, cam, 0, (0, 0, 0))
    assert int(rast.stats.generation) == gen + 1
    with pytest.raises(pkg._lib.GsrError) as e:
        rast.backward_raw(torch.zeros(48, 64, 3).cuda(), *t, cam, 0, (0, 0, 0), forward_generation=gen)
    assert e.value.code == pkg._lib.GSR_E_STATE
    rast.backward_raw(torch.zeros(48, 64, 3).cuda(), *t, cam, 0, (0, 0, 0), forward_generation=gen + 1)


def test_gstate_is_filled_by_the_library(pkg, orc):
    """The drop-in contract of strategy.jl:85-86: after a step the caller reads `rast.gstate.radii` and
    `rast.gstate.∇means_2d` — arrays the rasterizer OBJECT owns (rasterizer.jl:275-278), written in place by
    gsr_forward / gsr_backward through gsr_aux.radii / gsr_grads.vmeans2d (no copies, no accessor calls);
    they survive growth of the model and feed gsr_update_stats."""
    W, H = 96, 64
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb")
    for n, seed in ((200, 5), (700, 6), (300, 7)):  # grows, then shrinks: gstate is grow-only
        s, cam = _scene(pkg, orc, n, W, H, 1, seed, sigma_px=4.0)
        camera = pkg.Camera(W, H, tuple(s.focal))
        t = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
        st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1)
        vp = np.random.default_rng(seed).standard_normal((H, W, 3)).astype(np.float32)
        g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, 1)
        rast.forward_raw(*t, camera, 1, (0, 0, 0))
        p0 = rast.gstate.radii.data_ptr()
        rast.backward_raw(dev(vp), *t, camera, 1, (0, 0, 0))
        torch.cuda.synchronize()
        assert rast.gstate.radii.data_ptr() == p0 and len(rast.gstate) >= n
        assert rast.gstate.radii.shape == (n,) and rast.gstate.grad_means_2d.shape == (n, 2)
        assert np.array_equal(rast.gstate.radii.cpu().numpy(), st.radii)
        assert rel_l2(rast.gstate.grad_means_2d.cpu().numpy(), g.vmeans2d) <= 1e-4
        # the library reports the caller's arrays as the state buffers, and update_stats reads them
        ptr, sz = C.c_void_p(), C.c_size_t()
        pkg._lib.check(rast._lib.gsr_buffer(rast._h, pkg._lib.BUF_RADII, C.byref(ptr), C.byref(sz)))
        assert ptr.value == p0 and sz.value == 4 * n
        mr = torch.zeros(n, dtype=torch.int32, device="cuda"); acc = torch.zeros(n, device="cuda"); den = torch.zeros(n, device="cuda")
        rast.update_stats(mr, acc, den)
        mo, ao, do = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        orc.update_stats(mo, ao, do, st.radii, rast.gstate.grad_means_2d.cpu().numpy(), W, H)
        assert np.array_equal(mr.cpu().numpy(), mo) and np.array_equal(den.cpu().numpy(), do)
        assert np.allclose(acc.cpu().numpy(), ao, rtol=1e-6, atol=0)


def test_early_sort_pass_hits_and_misses_give_the_same_result(pkg):
    """The sort's main pass is enqueued before the host has read the instance count, against the capacity earlier
    views left behind (grow-only, 25 % slack); a view that needs more makes that pass a no-op and the host repeats
    it after growing the buffers.  Hit (a little more, within the slack), miss (much more), hit (far less): image,
    lists and gradients equal those of a fresh handle, bit for bit."""
    W, H, deg = 160, 112, 1
    cam = pkg.Camera(W, H, tuple(pkg.synthetic.make_scene(10, W, H, deg, 1).focal))
    vp = dev(np.random.default_rng(8).standard_normal((H, W, 3)).astype(np.float32))
    shared = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb")
    seen = []
    for n, seed in [(2000, 31), (2300, 32), (5200, 33), (700, 34), (5200, 33)]:
        s = pkg.synthetic.make_scene(n, W, H, deg, seed, sigma_px=4.0)
        t = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
        fresh = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb")
        out = []
        for r in (shared, fresh):
            img = r.forward_raw(*t, cam, deg, (0, 0, 0)).clone()
            g = r.backward_raw(vp, *t, cam, deg, (0, 0, 0))
            torch.cuda.synchronize()
            out.append((img, [x.clone() for x in g[:5]], r.stats.n_rendered, r.values_sorted.clone(), r.ranges.clone()))
        a, b = out
        assert a[2] == b[2] and a[2] > 0
        assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
        for x, y in zip(a[1], b[1]):
            assert torch.equal(x, y)
        seen.append(a[2])
    assert seen[1] < 1.25 * seen[0] < seen[2] and seen[3] < seen[0]   # the sequence did exercise hit / miss / hit


@pytest.mark.parametrize("mode,n_hot", [("rgb", 1500), ("rgb", 5000), ("rgbd", 2200), ("rgbdn", 1300), ("rgb", 9000)])
def test_backward_of_long_lists_runs_four_waves_per_tile_and_matches_the_oracle(pkg, orc, mode, n_hot):
    """Tiles whose list exceeds 1024 instances are left out by the one-wave-per-tile backward and walked on the handle's
    second stream (all three tier lists here: (1024, 4096], (4096, 8192], > 8192), next to ordinary tiles — round 2: by four
    waves per tile (pixel strips), round 5: by up to 32 single-wave workgroups per tile, each a SEGMENT of the list over all
    256 pixels, in two launches (composite_bwd_long_kernel).  Gradients and gstate.∇means_2d equal the oracle's, and a second
    backward-capable step on the same handle (no long tile any more) still does.  The same scenes exercise the tier sorts
    (register runs + LDS merges up to 8192 keys, many workgroups beyond)."""
    W, H, deg, n = 96, 64, 1, 600
    base = pkg.synthetic.make_scene(n, W, H, deg, 55, sigma_px=3.0)
    rng = np.random.default_rng(56)
    hot = pkg.synthetic.make_scene(n_hot, W, H, deg, 57, sigma_px=2.0)
    # the extra Gaussians sit inside one tile (pixels 32..47 x 16..31), faint enough that deep ones still contribute
    z = rng.uniform(2.0, 9.0, n_hot)
    fx = base.focal[0]
    u = rng.uniform(33.0, 46.0, n_hot) - W / 2.0
    v = rng.uniform(17.0, 30.0, n_hot) - H / 2.0
    hot.means[:] = np.stack([u * z / fx, v * z / base.focal[1], z], 1).astype(np.float32)
    means = np.concatenate([base.means, hot.means]); shs = np.concatenate([base.shs, hot.shs])
    opac = np.concatenate([base.opacities, np.full(n_hot, 0.004 + 40.0 / n_hot, np.float32)])
    scales = np.concatenate([base.scales, hot.scales * 0.5]); rots = np.concatenate([base.rotations, hot.rotations])
    cam = orc.Camera(W, H, base.focal)
    bg = (0.2, 0.4, 0.1)
    st = orc.forward(means, shs, opac, scales, rots, cam, deg, background=bg, mode=mode)
    lens = st.ranges[:, 1].astype(np.int64) - st.ranges[:, 0]
    assert lens.max() > 1024 and (lens > 0).sum() > 4 and np.median(lens[lens > 0]) < 1024
    run = HipRun(pkg, means, shs, opac, scales, rots, cam, deg, bg, mode)
    _compare_forward(st, run, run.forward())
    vp = np.random.default_rng(58).standard_normal(st.image.shape).astype(np.float32)
    g = orc.backward(st, vp, means, shs, opac, scales, rots, cam, deg, background=bg)
    out = run.backward(vp)
    _compare_backward(g, out, st.radii > 0)
    assert rel_l2(run.rast.grad_means_2d.cpu().numpy(), g.vmeans2d) <= 1e-4
    # same handle, an ordinary scene afterwards: nothing of the fork is left behind
    st2 = orc.forward(base.means, base.shs, base.opacities, base.scales, base.rotations, cam, deg, background=bg, mode=mode)
    run.t = [dev(base.means), dev(base.shs), dev(base.opacities.reshape(-1, 1)), dev(base.scales), dev(base.rotations)]
    _compare_forward(st2, run, run.forward())
    g2 = orc.backward(st2, vp, base.means, base.shs, base.opacities, base.scales, base.rotations, cam, deg, background=bg)
    _compare_backward(g2, run.backward(vp), st2.radii > 0)


def test_profile_intervals_are_the_launch_to_launch_times_of_a_stage(pkg):
    """gsr_profile_read_intervals: K launches of a stage give K-1 begin-to-begin intervals (bench.py's per-step times),
    each at least as long as the stage itself; unknown stages are refused."""
    W, H, deg, n = 128, 96, 1, 800
    s = pkg.synthetic.make_scene(n, W, H, deg, 91, sigma_px=4.0)
    cam = pkg.Camera(W, H, tuple(s.focal))
    t = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb")
    vp = torch.ones(H, W, 3, device="cuda")
    rast.profile(True, stages=["composite_bwd"])
    for _ in range(4):
        rast.forward_raw(*t, cam, deg, (0, 0, 0))
        rast.backward_raw(vp, *t, cam, deg, (0, 0, 0))
    torch.cuda.synchronize()
    iv = rast.profile_intervals("composite_bwd")
    prof = rast.profile_read()
    assert len(iv) == 3 and prof["composite_bwd"][1] == 4 and prof["composite_fwd"][1] == 0 and prof["sort_composite_fwd"][1] == 0
    assert all(x > 0 for x in iv) and min(iv) >= 0.5 * prof["composite_bwd"][0] / 4
    assert rast.profile_intervals("composite_fwd") == []
    with pytest.raises(ValueError):
        rast.profile_intervals("no_such_stage")
    rast.profile(False)
