"""-m gpu: GSR_FORWARD_ONLY (gsr_aux.flags, ABI 4) — the render of the reference's non-AD branch (rasterizer.jl:214-248:
`validate` training.jl:501-504, GUI gui/worker.jl:654-657, scripts/render-views.jl).  Same image / final T / n_contrib / radii /
ranges bit for bit as a training forward; no sorted stream, ids or gradient rows behind it; a backward after it is GSR_E_STATE."""
import numpy as np
import pytest
import torch

from hip_helpers import HipRun, dev

pytestmark = pytest.mark.gpu


def _pair(pkg, orc, s, cam, deg, mode, exact, bg=(0.2, 0.5, 0.1), **kw):
    full = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, bg, mode, exact_tile_cull=exact, **kw)
    only = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, bg, mode, exact_tile_cull=exact, **kw)
    return full, only


def _fwd_only(run):
    img = run.rast.forward_raw(*run.t, run.camera, run.deg, run.bg, run.Rd, run.td, run.covis, run.unc, forward_only=True)
    torch.cuda.synchronize()
    return img


def _same_outputs(full, only):
    a, b = full.forward().clone(), _fwd_only(only).clone()
    assert torch.equal(a, b)
    assert torch.equal(full.rast.accum_alpha, only.rast.accum_alpha)
    assert torch.equal(full.rast.n_contrib, only.rast.n_contrib)
    assert torch.equal(full.rast.radii, only.rast.radii)
    assert torch.equal(full.rast.ranges, only.rast.ranges)
    assert only.rast.stats.n_rendered == full.rast.stats.n_rendered and only.rast.stats.n_visible == full.rast.stats.n_visible
    return a


@pytest.mark.parametrize("mode", ["rgb", "rgbd", "rgbdn"])
@pytest.mark.parametrize("exact", [True, False])
def test_forward_only_is_bit_identical_and_keeps_no_backward_state(pkg, orc, mode, exact):
    W, H, n, deg = 320, 208, 12000, 2
    s = pkg.synthetic.make_scene(n, W, H, deg, 4100, sigma_px=4.0)
    cam = orc.Camera(W, H, s.focal)
    full, only = _pair(pkg, orc, s, cam, deg, mode, exact)
    for _ in range(2):  # first view (buffers grow, maybe compact mode) and steady state
        _same_outputs(full, only)
    L = pkg._lib
    # nothing per instance was stored ...
    with pytest.raises(RuntimeError):
        only.rast.values_sorted
    with pytest.raises(RuntimeError):
        only.rast.instance_masks
    # ... so a backward on that forward is refused, by both entry points
    C = only.rast.channels
    with pytest.raises(L.GsrError) as e:
        only.backward(np.zeros((H, W, C), np.float32))
    assert e.value.code == L.GSR_E_STATE and "FORWARD_ONLY" in str(e.value)
    # a handle that only ever rendered forward-only reserves no gradient rows (and, unless a view went through the rare
    # paths, no stream): never more than the training handle
    assert only.rast.memory_usage() < full.rast.memory_usage()
    # and a training forward on the same handle afterwards is a normal one
    img = only.forward().clone()
    vp = np.random.default_rng(1).standard_normal((H, W, C)).astype(np.float32)
    ga, gb = full.backward(vp), only.backward(vp)
    assert all(torch.equal(x, y) for x, y in zip(ga[:5], gb[:5])) and torch.equal(img, full.forward())


def test_forward_only_with_side_outputs_long_tiles_and_compact_binning(pkg, orc):
    """The rare paths still go through the stream: tiles beyond 1024 instances (tier sorts + a forward launch over the tier
    lists) and views binned in compact mode; covisibilities / uncertainties are written as in a training forward."""
    W, H, deg = 256, 160, 1
    base = pkg.synthetic.make_scene(20000, W, H, deg, 4200, sigma_px=4.0)
    s = pkg.synthetic.add_skew(base, "hot:6000", seed=4201)   # one tile with > 4096 instances
    cam = orc.Camera(W, H, s.focal)
    full, only = _pair(pkg, orc, s, cam, deg, "rgb", True, want_covis=True, want_uncert=True)
    for it in range(3):
        _same_outputs(full, only)
        assert torch.equal(full.covis, only.covis) and torch.equal(full.unc, only.unc)
    assert full.rast.stats.max_tile_instances > 4096
    # compact mode forced by a 1-byte bins budget
    for r in (full, only):
        r.rast.close()
        r.rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", bins_budget_bytes=1)
    _same_outputs(full, only)
    assert only.rast.stats.compact_binning == 1


def test_forward_only_empty_scene_and_flag_validation(pkg, orc):
    import ctypes as C
    s = pkg.synthetic.make_scene(64, 64, 48, 0, 5)
    means = s.means.copy()
    means[:, 2] = -3.0
    cam = orc.Camera(64, 48, s.focal)
    run = HipRun(pkg, means, s.shs, s.opacities, s.scales, s.rotations, cam, 0, (1, 1, 1))
    img = _fwd_only(run)
    assert run.rast.stats.n_rendered == 0 and not img.cpu().numpy().any()
    # unknown aux flag bits are rejected, not ignored
    L = pkg._lib
    inp = run.rast._inputs(*run.t, 0, (0, 0, 0))
    cs = run.rast._camera(run.camera, None, None)
    aux = L.Aux(None, None, None, 2, 0)
    rc = run.rast._lib.gsr_forward(run.rast._h, C.byref(inp), C.byref(cs), C.c_void_p(run.rast.image.data_ptr()), C.byref(aux),
                                   None, None)
    assert rc == L.GSR_E_INVALID_ARG


def test_rasterize_outside_ad_renders_forward_only(pkg, orc):
    """The mirror of the reference's `within_gradient` test (rasterizer.jl:214-215): under torch.no_grad(), or when no argument
    requires a gradient, `rasterize` keeps no backward state; under autograd it does."""
    W, H, n, deg = 128, 96, 2000, 1
    s = pkg.synthetic.make_scene(n, W, H, deg, 4300, sigma_px=4.0)
    camera = pkg.Camera(W, H, tuple(s.focal))
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb")
    t = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
    R = pkg.rasterizer
    tr = [x.clone().requires_grad_(True) for x in t]
    img_train = R.rasterize(*tr, rast=rast, camera=camera, sh_degree=deg).clone()
    img_train.sum().backward()
    assert tr[0].grad is not None and float(tr[0].grad.abs().sum()) > 0
    with torch.no_grad():
        img_eval = R.rasterize(*tr, rast=rast, camera=camera, sh_degree=deg).clone()
    assert torch.equal(img_eval, img_train.detach())
    with pytest.raises(pkg._lib.GsrError) as e:
        rast.backward_raw(torch.zeros(H, W, 3).cuda(), *t, camera, deg, (0, 0, 0))
    assert e.value.code == pkg._lib.GSR_E_STATE
    img_plain = R.rasterize(*t, rast=rast, camera=camera, sh_degree=deg)   # no argument requires a gradient
    assert torch.equal(img_plain, img_eval)
    with pytest.raises(pkg._lib.GsrError):
        rast.backward_raw(torch.zeros(H, W, 3).cuda(), *t, camera, deg, (0, 0, 0))


def test_manual_rasterize_grad_rasterize_pair_without_autograd(pkg, orc):
    """the reference's `rasterize` always keeps its backward state, so the manual pair `rasterize` ->
    `∇rasterize` on plain arrays (rasterizer.jl:255,416) works without AD.  The mirror's opt-outs of the forward-only default:
    `forward_only=False` per call, `rast.forward_only_outside_ad = False` per rasterizer; gradients equal autograd's."""
    W, H, n, deg = 128, 96, 2000, 1
    s = pkg.synthetic.make_scene(n, W, H, deg, 4301, sigma_px=4.0)
    camera = pkg.Camera(W, H, tuple(s.focal))
    R = pkg.rasterizer
    rast = R.GaussianRasterizer(W, H, mode="rgb")
    t = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
    vp = dev(pkg.synthetic.make_vpixels(W, H, 3, 9))
    tr = [x.clone().requires_grad_(True) for x in t]
    R.rasterize(*tr, rast=rast, camera=camera, sh_degree=deg).backward(vp)
    want = [x.grad.clone() for x in tr]
    # per call
    img = R.rasterize(*t, rast=rast, camera=camera, sh_degree=deg, forward_only=False).clone()
    got = R.grad_rasterize(vp, t[0], t[1], t[3], t[4], t[2], rast=rast, camera=camera, sh_degree=deg)
    for g, w in zip(got[:5], want):
        assert torch.equal(g.reshape(w.shape), w)
    # per rasterizer, also under no_grad
    rast.forward_only_outside_ad = False
    with torch.no_grad():
        img2 = R.rasterize(*t, rast=rast, camera=camera, sh_degree=deg)
        got = R.grad_rasterize(vp, t[0], t[1], t[3], t[4], t[2], rast=rast, camera=camera, sh_degree=deg)
    assert torch.equal(img2, img) and torch.equal(got[0].reshape(want[0].shape), want[0])
    # forcing the inference render still works, and a differentiated call refuses it
    R.rasterize(*t, rast=rast, camera=camera, sh_degree=deg, forward_only=True)
    with pytest.raises(pkg._lib.GsrError):
        R.grad_rasterize(vp, t[0], t[1], t[3], t[4], t[2], rast=rast, camera=camera, sh_degree=deg)
    with pytest.raises(ValueError):
        R.rasterize(*tr, rast=rast, camera=camera, sh_degree=deg, forward_only=True)
