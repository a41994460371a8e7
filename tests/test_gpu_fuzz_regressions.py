"""-m gpu: the diagnosed failures of round 3's fuzz campaign (profiles/r03/fuzz_parity.txt: 20 of 9 000 randomised / deep
scenes and 44 of 2 000 hostile ones missed a HIP-vs-oracle gradient criterion) as ASSERTIONS, arbitrated by the float64
autograd model (tests/f64_model.py) — the only evidence independent of both fp32 evaluations while the oracle cannot be pinned
on a live reference (runtests.jl:95-306 pins the adjoints with Float64 finite differences the same way).

A gradient tensor of a recorded case must satisfy ONE of
  (a) the suite's own criterion: rel-L2(HIP, oracle) <= 1e-4;
  (b) conditioning: the mean / scale / rotation adjoints of a Gaussian with axis ratio r pass through differences of s_i^2
      (render.jl:302-366): an input error is amplified by ~r^2, for ANY fp32 evaluation (round 3: 600 : 1 needles put the ORACLE
      2e-4 .. 6e-2 from float64; and the oracle sums its per-pixel terms in double, so on such Gaussians it is closer to float64
      than any fp32 accumulation — the reference's float atomics included — can be).  The tensor restricted to the
      well-conditioned Gaussians (axis ratio <= 10) meets (a), and on the others HIP stays inside the conditioning bound
      rel-L2(HIP, f64) <= 2e-6 * r_max^2 (capped at 0.1), or is no farther from float64 than 4 x the oracle (+ 1e-4);
  (c) a boundary pair: the largest contributor to ||HIP - oracle||^2 owns a (pixel, splat) pair within 4 ulps of the blend
      boundary alpha = 1/255 (render.jl:95) — such a pair is decided by the last bit of sigma / exp on either side —, at least
      80 % of the squared difference sits on the Gaussians that blend into those boundary pixels (a flipped pair changes the
      transmittance of everything behind it and the accum_rec recursion of everything in front of it AT THAT PIXEL,
      render.jl:237-258), the boundary pixels are a handful (<= max(4, 2 % of the image), the bound of the suite's image
      comparison), one of the two evaluations agrees with float64 (<= 1e-4: it meets (a) against the float64 model), and
      without those Gaussians the tensor meets (a).
Anything else — a difference spread over many Gaussians, or concentrated on one that is nowhere near the boundary and well
conditioned — fails: that would be a kernel bug."""
import numpy as np
import pytest
import torch

import f64_model as fm
import fuzz_scenes
from hip_helpers import HipRun, blend_boundary_pixels

pytestmark = pytest.mark.gpu
DT = torch.float64
NAMES = ("vmeans", "vshs", "vopacities", "vscales", "vrots")


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def three_way(pkg, orc, fs, exact_tile_cull=False, grad_precision=None):
    """oracle, HIP and float64 gradients of one scene -> {tensor: (oracle, hip, f64) as (N, -1) float64 arrays}, oracle state."""
    st = orc.forward(fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, mode=fs.mode)
    vp = fs.cotangent()
    g = orc.backward(st, vp, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg)
    run = HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, fs.mode, exact_tile_cull=exact_tile_cull,
                 grad_precision=grad_precision)
    run.forward()
    out = [o.cpu().numpy() for o in run.backward(vp)[:5]]
    tt = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=DT, requires_grad=True)  # noqa: E731
    leaves = [tt(fs.means), tt(fs.shs), tt(fs.opac), tt(fs.scales), tt(fs.rots)]
    img = fm.render_dense(*leaves, fs.cam, fs.deg, np.asarray(fs.bg, np.float32), fs.mode, st.values_sorted, st.ranges, st.radii)
    (img * torch.tensor(vp, dtype=DT)).sum().backward()
    n = fs.means.shape[0]
    ref = (g.vmeans, g.vshs, g.vopacities, g.vscales, g.vrots)
    res = {}
    for nm, o, r, leaf in zip(NAMES, out, ref, leaves):
        res[nm] = tuple(np.asarray(x, np.float64).reshape(n, -1) for x in (r, o, leaf.grad.numpy()))
    return res, st


def reference_form_distance(orc, fs, st, res, runs=5):
    """The reference's OWN accumulation form as a fourth column (round-5 verdict, next #2): ∇render!'s per-pixel contributions
    added with float32 atomics in whatever order the threads arrive (render.jl:242,262-282) — `orc.backward(deterministic=False)`:
    fp32 `#pragma omp atomic` accumulators under an OpenMP tile loop.  `runs` evaluations (the order differs from run to run and
    with the thread count); returns {tensor: (best, worst) rel-L2 against float64 over the visible Gaussians}.  DESIGN.md §4.2 had
    ASSERTED that "the reference's per-pixel fp32 atomics have the same noise or more" than HIP's fp32 wave reduction; this
    measures it.  `fs`: a FRESH scene object of the case (the campaign's cotangent is its rng's first draw)."""
    vis = st.radii > 0
    vp = fs.cotangent()
    n = fs.means.shape[0]
    out = {nm: [] for nm in NAMES}
    threads0 = orc.num_threads()
    try:
        for r in range(runs):
            orc.set_num_threads(max(2, threads0 >> (r % 3)))     # (another thread count = another arrival order)
            g = orc.backward(st, vp, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, deterministic=False)
            for nm, a in zip(NAMES, (g.vmeans, g.vshs, g.vopacities, g.vscales, g.vrots)):
                out[nm].append(_rel(np.asarray(a, np.float64).reshape(n, -1)[vis], res[nm][2][vis]))
    finally:
        orc.set_num_threads(threads0)
    return {nm: (min(v), max(v)) for nm, v in out.items()}


def arbitrate(res, st, fs):
    """Apply (a) / (b) / (c) to every tensor; returns {tensor: verdict string}; raises AssertionError with the numbers."""
    vis = st.radii > 0
    W, H = fs.cam.width, fs.cam.height
    owners = touched = None
    few_pixels = False
    verdicts = {}
    for nm, (orc_g, hip_g, f64_g) in res.items():
        e_ho, e_o, e_h = _rel(hip_g[vis], orc_g[vis]), _rel(orc_g[vis], f64_g[vis]), _rel(hip_g[vis], f64_g[vis])
        if e_ho <= 1e-4:
            verdicts[nm] = f"(a) {e_ho:.1e}"
            continue
        sc = np.abs(np.asarray(fs.scales, np.float64))
        ratio = sc.max(1) / np.maximum(sc.min(1), 1e-30)
        ill = ratio > 10.0
        well = vis & ~ill
        bad = vis & ill
        if bad.any() and _rel(hip_g[well], orc_g[well]) <= 1e-4:
            e_hb, e_ob = _rel(hip_g[bad], f64_g[bad]), _rel(orc_g[bad], f64_g[bad])
            bound = min(2e-6 * float(ratio[bad].max()) ** 2, 0.1)
            if e_hb <= bound or e_hb <= 4.0 * e_ob + 1e-4:
                verdicts[nm] = (f"(b) {int(bad.sum())} Gaussians beyond 10 : 1 (worst {ratio[bad].max():.0f} : 1, bound {bound:.1e}): HIP-f64 "
                                f"{e_hb:.1e}, oracle-f64 {e_ob:.1e}; the other {int(well.sum())}: HIP-oracle {_rel(hip_g[well], orc_g[well]):.1e}")
                continue
        d2 = ((hip_g - orc_g) ** 2).sum(1)
        if owners is None:
            bmask, owners, touched = blend_boundary_pixels(st, fs.opac, W, H, with_ids=True)
            few_pixels = int(bmask.sum()) <= max(4, 0.02 * W * H)
        top = int(np.argmax(d2))
        sel = np.zeros(vis.shape[0], bool)
        sel[list(touched)] = True
        share = float(d2[sel].sum() / max(d2.sum(), 1e-300))
        rest = vis & ~sel
        e_rest = _rel(hip_g[rest], orc_g[rest])
        frac = sel[vis].mean()
        ok = top in owners and share >= 0.8 and few_pixels and min(e_o, e_h) <= 1e-4 and e_rest <= 1e-4
        assert ok, (f"{nm}: HIP-oracle {e_ho:.2e}, oracle-f64 {e_o:.2e}, HIP-f64 {e_h:.2e}; largest contributor {top} (axis ratio "
                    f"{ratio[top]:.1f}, radius {int(st.radii[top])} px, {100 * d2[top] / max(d2.sum(), 1e-300):.0f} % of it; owns a "
                    f"boundary pair: {top in owners}); {100 * share:.0f} % of the squared difference on the {int(sel.sum())} Gaussians "
                    f"({100 * frac:.0f} % of the visible ones) that blend into the boundary pixels, the rest {e_rest:.2e}")
        verdicts[nm] = (f"(c) pair of Gaussian {top}: {100 * share:.0f} % on {int(sel.sum())} Gaussians of the boundary pixels, rest "
                        f"{e_rest:.1e}, oracle-f64 {e_o:.1e}, HIP-f64 {e_h:.1e}")
    return verdicts


# randomised sweep: the five cases tools/dbg_sweep3.py looked at in round 3 + the two of the first 2 000
@pytest.mark.parametrize("case", [137, 1160, 3127, 5124, 6928, 3285, 4629])
def test_sweep_failures_are_boundary_pairs_or_conditioning(pkg, orc, case):
    fs = fuzz_scenes.sweep_scene(pkg, case)
    res, st = three_way(pkg, orc, fs)
    print(f"sweep {case}", arbitrate(res, st, fs))


# hostile inputs: needles (27 : 1 stacked on screen-filling / scene spread: beyond 30 : 1) and one faint large Gaussian (1651)
@pytest.mark.parametrize("case", [1518, 1575, 1651, 1847, 1959, 1981, 1990])
def test_edge_failures_are_conditioning_or_boundary_pairs(pkg, orc, case):
    fs = fuzz_scenes.edge_scene(pkg, case)
    res, st = three_way(pkg, orc, fs)
    print(f"edge {case}", arbitrate(res, st, fs))


# round 5, last campaign: the two single-pixel IMAGE differences of 1 900 scenes, both on the plain fused path.  Sweep 14113: one
# pixel whose walk stops one entry apart (n_contrib 103 / 102: the saturation test T' < 1e-4 decided by the last bits of T).
# Edge 7117: one pair 17 ulps of tau from the blend-test boundary, 61 px from the centre of a large anisotropic footprint, where
# sigma = 5.3 is the sum of three terms of +-100 (0.5 ulp of the largest): hip_helpers.blend_boundary_pixels now measures its
# window in ulps of the largest term.
@pytest.mark.parametrize("family,case", [("sweep", 14113), ("edge", 7117)])
def test_single_pixel_image_differences_are_boundary_decisions(pkg, orc, family, case):
    import test_gpu_parity as T
    fs = fuzz_scenes.sweep_scene(pkg, case) if family == "sweep" else fuzz_scenes.edge_scene(pkg, case)
    st = orc.forward(fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, mode=fs.mode)
    run = HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, fs.mode)
    img = run.forward().clone()
    T._compare_forward(st, run, img, fs.opac)
    cul = HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, fs.mode, exact_tile_cull=True)
    assert torch.equal(cul.forward(), img)


# round 5 (round-4 verdict "weak #2"): the three hostile scenes of the last campaign in which HIP — not the oracle — was the side
# far from float64 on ∇rotations (profiles/r04/fuzz_parity_last.txt: HIP-f64 2.1e-4 / 5.2e-4 / 2.7e-4 against oracle-f64
# 1.3e-5 / 2.9e-4 / 3.9e-5).  Cause: the fp32 chain ∇inverse -> ... -> ∇unnorm_quat2rot loses a needle's thin
# eigen-direction, so ANY fp32 evaluation turns the last bit of vconic into 1e-3 of ∇rotations; which side looks better is
# luck.  pergauss_bwd now evaluates that chain in float64 from the raw inputs (scales_rots_bwd_f64): ∇scales and ∇rotations
# must meet the suite's 1e-4 directly AGAINST FLOAT64, whatever the fp32 oracle does.
@pytest.mark.parametrize("case", [5315, 5378, 5457])
def test_needle_rotation_gradients_meet_the_tolerance_against_float64(pkg, orc, case):
    fs = fuzz_scenes.edge_scene(pkg, case)
    res, st = three_way(pkg, orc, fs)
    vis = st.radii > 0
    for nm in ("vscales", "vrots"):
        o, h, t = res[nm]
        e_h, e_o = _rel(h[vis], t[vis]), _rel(o[vis], t[vis])
        print(f"edge {case} {nm}: HIP-f64 {e_h:.2e}, oracle-f64 {e_o:.2e}")
        assert e_h <= 1e-4, (nm, e_h, e_o)
    print(f"edge {case}", arbitrate(res, st, fs))


# round 6 (round-5 verdict, next #2): ∇means of needle-shaped splats.  On edge 8498 / 8112 HIP — not the oracle — is the far side
# from float64 (4.1e-4 / 2.1e-4 against 2.3e-5 / 9.4e-5).  DESIGN.md §4.2 (round 5) had ASSERTED that "the reference's per-pixel
# fp32 atomics have the same noise or more"; measured now with the oracle's like-for-like mode (`deterministic=False`: fp32 atomics
# per pixel, render.jl:242,262-282, five arrival orders): 2e-5 .. 3e-4 on 8498 — the same order, HIP up to 3 x worse than its
# worst.  And the source is NOT the wave reduction round 5 suspected: it is the two fast hardware functions of the per-pixel
# body (v_exp_f32 on the rounded product sigma x log2 e, v_rcp_f32 through the T recursion) — with libm exp + IEEE division
# (gsr_config.grad_precision = GSR_GRAD_ACCURATE, +12 % of ∇render!) 8498 drops to 2.3e-5, the oracle's own figure
# (profiles/r06/experiments/needle_*.txt).  Asserted: ACCURATE meets 1e-4 against float64 or sits on the fp32 oracle; the DEFAULT
# stays within 3 x the worst arrival order of the reference's own accumulation form, or 5e-4 (a regression bound: measured 4.1e-4).
@pytest.mark.parametrize("case", [8498, 8112, 5315, 5457])
def test_needle_means_gradient_default_and_accurate_arithmetic(pkg, orc, case):
    fs = fuzz_scenes.edge_scene(pkg, case)
    res, st = three_way(pkg, orc, fs)
    res_acc, _ = three_way(pkg, orc, fuzz_scenes.edge_scene(pkg, case), grad_precision="accurate")
    ref = reference_form_distance(orc, fuzz_scenes.edge_scene(pkg, case), st, res)   # (a fresh scene: the cotangent is its rng's FIRST draw)
    vis = st.radii > 0
    for nm in ("vmeans", "vopacities", "vshs", "vscales", "vrots"):
        o, h, t = res[nm]
        e_h, e_o, e_a = _rel(h[vis], t[vis]), _rel(o[vis], t[vis]), _rel(res_acc[nm][1][vis], t[vis])
        print(f"edge {case} {nm}: float64 distance of HIP default {e_h:.2e}, HIP accurate {e_a:.2e}, oracle (double sums) {e_o:.2e}, "
              f"reference form (fp32 atomics) {ref[nm][0]:.2e} .. {ref[nm][1]:.2e}")
        if nm in ("vmeans", "vopacities", "vshs"):   # (∇scales / ∇rotations: test_needle_rotation_gradients_... above)
            assert e_a <= 1e-4 or e_a <= 1.5 * e_o + 2e-5, (nm, e_a, e_o)
            assert e_h <= 1e-4 or e_h <= max(3.0 * ref[nm][1], 5e-4), (nm, e_h, ref[nm], e_o)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_600_to_1_needles_hip_is_as_close_to_float64_as_the_oracle(pkg, orc, seed):
    """No 1e-4 criterion between two fp32 evaluations holds on 600 : 1 needles (round 3: 2.2e-4 / 3.2e-4 / 4.7e-4); what must
    hold is that the kernels are not the worse of the two."""
    fs = fuzz_scenes.needle_scene(pkg, seed)
    res, st = three_way(pkg, orc, fs)
    print(f"needle {seed}", arbitrate(res, st, fs))
    m = fs.note["needles"] & (st.radii > 0)
    assert m.any()
    for nm in ("vmeans", "vscales", "vrots"):   # the needles alone, where the conditioning bites
        o, h, t = res[nm]
        assert _rel(h[m], t[m]) <= 4.0 * _rel(o[m], t[m]) + 1e-4, nm
    # round 5: ∇scales / ∇rotations come out of a float64 chain — on the needles they now meet the suite's tolerance against
    # float64 outright (round 3 measured 2.2e-4 .. 4.7e-4 for the fp32 chain, the oracle's fp32 chain 2e-4 .. 6e-2)
    for nm in ("vscales", "vrots"):
        o, h, t = res[nm]
        print(f"needle {seed} {nm}: HIP-f64 {_rel(h[m], t[m]):.2e}, oracle-f64 {_rel(o[m], t[m]):.2e}")
        assert _rel(h[m], t[m]) <= 1e-4, (nm, _rel(h[m], t[m]))


def test_deep_case_523_differs_on_one_boundary_gaussian_only(pkg, orc):
    """41 514 instances in one tile, opacity 0.0066: v_opacities was 1.2e-4 off from ONE pair on the boundary.  (No float64
    arbitration here — the dense model of a 41 k list needs tens of GB — but the structure of the difference is asserted.)"""
    fs = fuzz_scenes.deep_scene(pkg, 523)
    st = orc.forward(fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, mode=fs.mode)
    vp = fs.cotangent()
    g = orc.backward(st, vp, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg)
    run = HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, fs.mode)
    run.forward()
    out = [o.cpu().numpy() for o in run.backward(vp)[:5]]
    vis = st.radii > 0
    _, owners, touched = blend_boundary_pixels(st, fs.opac, fs.cam.width, fs.cam.height, with_ids=True)
    n = fs.means.shape[0]
    for nm, o, r in zip(NAMES, out, (g.vmeans, g.vshs, g.vopacities, g.vscales, g.vrots)):
        o, r = np.asarray(o, np.float64).reshape(n, -1), np.asarray(r, np.float64).reshape(n, -1)
        e = _rel(o[vis], r[vis])
        if e <= 1e-4:
            continue
        d2 = ((o - r) ** 2).sum(1)
        w = int(np.argmax(d2))
        sel = np.zeros(n, bool); sel[list(touched)] = True
        rest = vis & ~sel
        assert w in owners and d2[sel].sum() >= 0.8 * d2.sum() and _rel(o[rest], r[rest]) <= 1e-4, (nm, e, w, w in owners)


@pytest.mark.parametrize("mode", ["rgb", "rgbd"])
def test_every_instance_culled_still_shows_the_background(pkg, orc, mode):
    """Final campaign of round 4, edge case 4434: ONE Gaussian of opacity 0 over a non-zero background.  The reference renders its
    15 instances (each blends nothing) and the pixels show the background; the exact footprint cull drops all 15 — and the
    library then took the reference's "no instance at all" exit (all-zero image, rasterizer.jl:283,338).  With a visible Gaussian
    whose rect holds a tile the view IS rendered: background, T = 1, no contributor — bit-identical to the reference-lists mode."""
    import fuzz_scenes
    fs = fuzz_scenes.edge_scene(pkg, 4434)
    assert fs.means.shape[0] == 1 and float(fs.opac[0]) == 0.0 and any(b != 0 for b in fs.bg)
    ref = HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, mode)
    cul = HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, mode, exact_tile_cull=True)
    a, b = ref.forward().clone(), cul.forward().clone()
    assert ref.rast.stats.n_rendered == 15 and cul.rast.stats.n_rendered == 0 and cul.rast.stats.n_visible == 1
    assert torch.equal(a, b) and float(a.abs().max()) > 0
    assert torch.equal(ref.rast.accum_alpha, cul.rast.accum_alpha) and torch.equal(ref.rast.n_contrib, cul.rast.n_contrib)
    st = orc.forward(fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, mode=mode)
    assert np.array_equal(b.cpu().numpy().reshape(st.image.shape), st.image)
    # a backward on it is all zeros, in both modes
    C = cul.rast.channels
    vp = np.random.default_rng(0).standard_normal((fs.cam.height, fs.cam.width, C)).astype(np.float32)
    for g in cul.backward(vp)[:5]:
        assert not g.cpu().numpy().any()
    # and a view with NO visible Gaussian at all keeps the reference's all-zero image
    behind = fs.means.copy(); behind[:, 2] = -5.0
    none = HipRun(pkg, behind, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, mode, exact_tile_cull=True)
    assert not none.forward().cpu().numpy().any() and none.rast.stats.n_visible == 0
