"""-m gpu: the two forms of preprocess's binning (gsr_preprocess_form) — direct (one returning global atomic per instance
pair) and aggregating (a workgroup of 512 Gaussians adds its requests up per counter word in LDS, one global atomic per word
in address order; the default for large scenes).  Only the arbitrary order of the unsorted keys inside a tile's bin may differ:
records, sorted lists, ranges, image, final T, radii and every gradient must be bit-identical, on every binning path
(fixed-capacity bins, first view, bin overflow -> compact, forced compact, wave-emitted large rects, odd grids, both cull modes)."""
import contextlib

import numpy as np
import pytest
import torch

from hip_helpers import HipRun

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def form(pkg, f):
    L = pkg._lib
    L.check(L.load().gsr_preprocess_form(f))
    try:
        yield
    finally:
        L.check(L.load().gsr_preprocess_form(-1))


def _views(pkg, orc, s, W, H, deg, mode, exact, f, n_views=3, vp_seed=3, **kw):
    """n_views forward + backward passes on one handle under form f; returns everything comparable per view."""
    cam = orc.Camera(W, H, s.focal)
    out = []
    with form(pkg, f):
        run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, (0.1, 0.3, 0.2), mode,
                     exact_tile_cull=exact, **kw)
        C = run.rast.channels
        vp = np.random.default_rng(vp_seed).standard_normal((H, W, C)).astype(np.float32)
        for _ in range(n_views):
            img = run.forward().clone()
            st = run.rast.stats
            rec = dict(img=img, T=run.rast.accum_alpha.clone(), nc=run.rast.n_contrib.clone(), radii=run.rast.radii.clone(),
                       ranges=run.rast.ranges.clone(), ids=run.rast.values_sorted.clone(),
                       stats=(st.n_rendered, st.n_visible, st.max_tile_instances, st.compact_binning))
            rec["grads"] = [g.clone() for g in run.backward(vp)[:5]]
            out.append(rec)
        run.rast.close()
    return out


def _assert_same(a, b):
    assert len(a) == len(b)
    for va, vb in zip(a, b):
        assert va["stats"] == vb["stats"]
        for k in ("img", "T", "nc", "radii", "ranges", "ids"):
            assert torch.equal(va[k], vb[k]), k
        for ga, gb in zip(va["grads"], vb["grads"]):
            assert torch.equal(ga, gb)


@pytest.mark.parametrize("mode", ["rgb", "rgbdn"])
@pytest.mark.parametrize("exact", [True, False])
def test_both_forms_give_identical_lists_images_and_gradients(pkg, orc, mode, exact):
    W, H, n, deg = 328, 200, 30000, 2   # 21 x 13 tiles: odd grid, the last tile pair is half outside the counters' pairs
    s = pkg.synthetic.make_scene(n, W, H, deg, 5100, sigma_px=5.0)
    _assert_same(_views(pkg, orc, s, W, H, deg, mode, exact, 0), _views(pkg, orc, s, W, H, deg, mode, exact, 1))


def test_forms_agree_on_large_rects_hot_tiles_and_compact_binning(pkg, orc):
    W, H, deg = 256, 160, 1
    base = pkg.synthetic.make_scene(20000, W, H, deg, 5200, sigma_px=4.0)
    # one tile beyond 4096 instances (tier sorts), and Gaussians whose rects exceed the per-thread walk (wave-emitted)
    s = pkg.synthetic.add_skew(base, "hot:6000", seed=5201)
    import dataclasses
    big = s.scales_raw.copy()
    big[:40] += np.log(30.0).astype(np.float32)   # rects of hundreds of tiles: beyond the per-thread walk
    s = dataclasses.replace(s, scales_raw=big)
    a = _views(pkg, orc, s, W, H, deg, "rgb", True, 0)
    b = _views(pkg, orc, s, W, H, deg, "rgb", True, 1)
    _assert_same(a, b)
    assert a[-1]["stats"][2] > 4096
    # compact binning (count -> scan -> scatter) forced by a 1-byte bins budget: the aggregating form only counts
    a = _views(pkg, orc, s, W, H, deg, "rgb", True, 0, n_views=2, bins_budget_bytes=1)
    b = _views(pkg, orc, s, W, H, deg, "rgb", True, 1, n_views=2, bins_budget_bytes=1)
    _assert_same(a, b)
    assert a[-1]["stats"][3] == 1


def test_default_form_by_size_and_argument_check(pkg, orc):
    L = pkg._lib
    lib = L.load()
    assert lib.gsr_preprocess_form(2) == L.GSR_E_INVALID_ARG and lib.gsr_preprocess_form(-2) == L.GSR_E_INVALID_ARG
    # 300 k Gaussians at 640 x 360: the default picks the aggregating form; same outputs as the direct form forced
    W, H, n, deg = 640, 360, 300_000, 1
    s = pkg.synthetic.make_scene(n, W, H, deg, 5300, sigma_px=1.5)
    _assert_same(_views(pkg, orc, s, W, H, deg, "rgb", True, -1, n_views=2), _views(pkg, orc, s, W, H, deg, "rgb", True, 0, n_views=2))


def test_sixteen_bit_words_on_a_large_grid(pkg, orc):
    """Grids whose 64-bit counter words do not fit the LDS three times per CU (beyond ~10 700 tiles) take the aggregating form
    with 2 x 16-bit words up to ~21 500 tiles (positions clamp at "not stored"): 2560 x 1440 = 14 400 tiles here; a hot tile far beyond the first
    view's bin capacity exercises the clamp; the direct form is the reference, bit for bit."""
    W, H, deg = 2560, 1440, 1
    base = pkg.synthetic.make_scene(60000, W, H, deg, 5400, sigma_px=6.0)
    s = pkg.synthetic.add_skew(base, "hot:9000", seed=5401)
    _assert_same(_views(pkg, orc, s, W, H, deg, "rgb", True, 0), _views(pkg, orc, s, W, H, deg, "rgb", True, 1))


@pytest.mark.parametrize("size", [(3840, 2160), (3848, 2168)])
def test_banded_aggregating_form_on_4k_grids(pkg, orc, size):
    """Round 5 (round-4 verdict, next #3): grids beyond ~21 500 tiles take the aggregating form in horizontal BANDS of the tile
    grid (4K: two bands of 2 x 16-bit words; the odd 241 x 136 grid makes rows alternate between two pair counts AND puts a band
    boundary inside a counter word).  Lists, images, gradients bit-identical to the direct form; rects that straddle the band
    boundary, wave-emitted large rects, a hot tile beyond the first view's capacity (clamped positions), the overflow -> compact
    path and the forced compact mode (whose SCATTER pass is the same kernel) are all in the scene."""
    import dataclasses
    W, H = size
    deg = 1
    base = pkg.synthetic.make_scene(120_000, W, H, deg, 5500, sigma_px=7.0)
    s = pkg.synthetic.add_skew(base, "hot:5000", seed=5501)
    big = s.scales_raw.copy()
    big[:30] += np.log(40.0).astype(np.float32)
    s = dataclasses.replace(s, scales_raw=big)
    L = pkg._lib
    a = _views(pkg, orc, s, W, H, deg, "rgb", True, 0)
    b = _views(pkg, orc, s, W, H, deg, "rgb", True, 1)
    _assert_same(a, b)
    # the form that ran is reported: 3 = banded
    with form(pkg, 1):
        cam = orc.Camera(W, H, s.focal)
        run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
        run.forward(); run.forward()
        assert int(run.rast.stats.preprocess_form) == 3
        run.rast.close()
    a = _views(pkg, orc, s, W, H, deg, "rgb", False, 0, n_views=2, bins_budget_bytes=1)   # forced compact: count + SCATTER pass
    b = _views(pkg, orc, s, W, H, deg, "rgb", False, 1, n_views=2, bins_budget_bytes=1)
    _assert_same(a, b)
    assert a[-1]["stats"][3] == 1


def test_compact_scatter_pass_with_lists_beyond_sixteen_bit_positions(pkg, orc):
    """A tile of > 65 535 instances: the scatter pass cannot hand out 16-bit positions and takes 2 x 32-bit LDS words (in bands
    where they do not fit); ids and ranges equal the oracle's, image within tolerance."""
    from hip_helpers import frac_bad
    W, H, deg = 640, 480, 0
    base = pkg.synthetic.make_scene(5000, W, H, deg, 5600, sigma_px=3.0)
    s = pkg.synthetic.add_skew(base, "hot:70000", seed=5601)
    cam = orc.Camera(W, H, s.focal)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)   # reference lists
    img = run.forward().cpu().numpy()
    assert int(run.rast.stats.compact_binning) == 1 and int(run.rast.stats.max_tile_instances) > 65535
    assert np.array_equal(run.rast.values_sorted.cpu().numpy().astype(np.uint32), st.values_sorted)
    assert np.array_equal(run.rast.ranges.cpu().numpy().astype(np.uint32), st.ranges)
    assert frac_bad(img, st.image, 0, 1e-4) <= 1e-4


def test_default_form_on_a_4k_grid_is_measured_once_and_kept(pkg, orc):
    """Grids that need bands (4K) with scenes of the aggregating form's size: neither form wins everywhere, so the handle times
    two views in each (views 3 .. 6: the first two grow buffers) and keeps the faster.  gsr_stats.preprocess_form shows the
    sequence; every view gives the same image and lists whatever form ran."""
    W, H, deg, n = 3840, 2160, 0, 260_000
    s = pkg.synthetic.make_scene(n, W, H, deg, 5700, sigma_px=3.0)
    cam = orc.Camera(W, H, s.focal)
    run = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    forms, first = [], None
    for it in range(12):
        img = run.forward()
        forms.append(int(run.rast.stats.preprocess_form))
        if first is None:
            first = (img.clone(), run.rast.values_sorted.clone(), run.rast.ranges.clone())
        else:
            assert torch.equal(img, first[0]) and torch.equal(run.rast.values_sorted, first[1]) and torch.equal(run.rast.ranges, first[2])
    assert forms[:6] == [0, 0, 0, 3, 0, 3], forms      # untimed default (uniform scene: direct), then direct / banded timed twice
    assert forms[7] in (0, 3) and forms[7:] == [forms[7]] * 5, forms   # decided at view 7 or 8, and kept
    # a forced form is not touched by any of this
    with form(pkg, 1):
        run.forward()
        assert int(run.rast.stats.preprocess_form) == 3
    with form(pkg, 0):
        run.forward()
        assert int(run.rast.stats.preprocess_form) == 0
    run.rast.close()
