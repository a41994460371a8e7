"""-m gpu parity of the two streaming passes either side of rasterize(), through the C ABI:
gsr_prologue_forward/backward (rasterizer.jl:200-253) and gsr_adam_step (NU.step!,
training.jl:234-239,778) against the oracle on the same inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

from hip_helpers import dev

pytestmark = pytest.mark.gpu


def _raw(n, kr, iso, seed):
    r = np.random.default_rng(seed)
    return (r.normal(size=(n, 1, 3)).astype(np.float32), r.normal(size=(n, kr, 3)).astype(np.float32) if kr else None,
            r.normal(size=(n, 1)).astype(np.float32) * 3, r.normal(size=(n, 1 if iso else 3)).astype(np.float32))


@pytest.mark.parametrize("n,kr,iso", [(1000, 15, False), (257, 0, False), (4099, 8, True), (1, 3, False)])
def test_prologue_forward_vs_oracle(pkg, orc, n, kr, iso):
    dc, rest, o, s = _raw(n, kr, iso, 21)
    shs, oa, sa = orc.prologue_forward(dc, rest, o, s)
    R = pkg.rasterizer
    h = R.prologue_forward(dev(dc), dev(rest) if kr else None, dev(o), dev(s))
    torch.cuda.synchronize()
    assert np.array_equal(h[0].cpu().numpy(), shs)                       # hcat: bit-exact
    # sigmoid / exp: the device expf and glibc expf are each within 1 ulp of the true value
    np.testing.assert_allclose(h[1].cpu().numpy(), oa, rtol=5e-7, atol=1e-37)
    np.testing.assert_allclose(h[2].cpu().numpy(), sa, rtol=5e-7, atol=1e-37)


@pytest.mark.parametrize("n,kr,iso", [(1000, 15, False), (4099, 8, True), (300, 0, False)])
def test_prologue_backward_vs_oracle_bit_exact(pkg, orc, n, kr, iso):
    dc, rest, o, s = _raw(n, kr, iso, 22)
    shs, oa, sa = orc.prologue_forward(dc, rest, o, s)
    r = np.random.default_rng(9)
    vshs, voa, vsa = (r.normal(size=a.shape).astype(np.float32) for a in (shs, oa, sa))
    want = orc.prologue_backward(oa, sa, vshs, voa, vsa, 1 if iso else 3)
    got = pkg.rasterizer.prologue_backward(dev(oa), dev(sa), dev(vshs), dev(voa), dev(vsa), 1 if iso else 3)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)  # same fp32 expression tree, no contraction


def test_prologue_autograd_matches_torch_ops(pkg):
    """The functor's prologue under autograd == torch.cat / sigmoid / exp under autograd."""
    n, kr = 500, 15
    dc, rest, o, s = _raw(n, kr, False, 5)
    a = [dev(x).requires_grad_(True) for x in (dc, rest, o, s)]
    b = [dev(x).requires_grad_(True) for x in (dc, rest, o, s)]
    P = pkg.rasterizer._Prologue
    ya = P.apply(*a)
    yb = (torch.cat([b[0], b[1]], 1), torch.sigmoid(b[2]), torch.exp(b[3]))
    w = [torch.randn_like(y) for y in yb]
    sum((y * k).sum() for y, k in zip(ya, w)).backward()
    sum((y * k).sum() for y, k in zip(yb, w)).backward()
    for x, y in zip(a, b):
        np.testing.assert_allclose(x.grad.cpu().numpy(), y.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)


def test_adam_vs_oracle_bit_exact_multi_group(pkg, orc):
    """Six parameter groups (training.jl:234-239) in one launch, odd lengths and an unaligned
    slice (scalar tail path), 4 steps: θ, μ, ν bit-identical to the oracle."""
    r = np.random.default_rng(31)
    n = 1237
    shapes = [(n, 3), (n, 1, 3), (n, 15, 3), (n, 1), (n, 3), (n, 4)]
    lrs = [1.6e-4, 2.5e-3, 2.5e-3 / 20, 2.5e-2, 5e-3, 1e-3]
    th = [r.normal(size=s).astype(np.float32) for s in shapes]
    O = pkg.optim
    th_d = [dev(t) for t in th]
    # group 3 lives at an odd element offset of a larger buffer: not 16-byte aligned
    big = torch.zeros(n + 1, device="cuda"); big[1:] = th_d[3].reshape(-1); th_d[3] = big[1:].view(n, 1)
    opts = [O.Adam(t, lr, eps=1e-15) for t, lr in zip(th_d, lrs)]
    mu = [np.zeros(t.size, np.float32) for t in th]; nu = [np.zeros(t.size, np.float32) for t in th]
    for step in range(1, 5):
        gr = [(r.normal(size=s) * 10.0 ** r.uniform(-4, 2)).astype(np.float32) for s in shapes]
        O.step_all(opts, th_d, [dev(g) for g in gr])
        for t, g, m, v, lr in zip(th, gr, mu, nu, lrs):
            orc.adam_step(t.reshape(-1), g.reshape(-1), m, v, step, lr, 0.9, 0.999, 1e-15)
    torch.cuda.synchronize()
    for t, td, m, v, o in zip(th, th_d, mu, nu, opts):
        assert o.current_step == 4
        assert np.array_equal(td.cpu().numpy(), t)
        assert np.array_equal(o.mu.cpu().numpy(), m) and np.array_equal(o.nu.cpu().numpy(), v)


def test_adam_large_single_group_and_reset(pkg, orc):
    r = np.random.default_rng(2)
    n = 3 * 16 * 100_003  # a features_rest-sized array, not a multiple of the 1024-element workgroup
    th = r.normal(size=n).astype(np.float32); g = r.normal(size=n).astype(np.float32)
    td = dev(th); opt = pkg.optim.Adam(td, 2.5e-3 / 20, eps=1e-15)
    opt.step(td, dev(g))
    mu = np.zeros(n, np.float32); nu = np.zeros(n, np.float32)
    orc.adam_step(th, g, mu, nu, 1, 2.5e-3 / 20, 0.9, 0.999, 1e-15)
    assert np.array_equal(td.cpu().numpy(), th)
    opt.reset()  # NU.reset! (strategy.jl:102)
    assert opt.current_step == 0 and not opt.mu.any() and not opt.nu.any()


def test_adam_errors_and_empty_groups(pkg):
    L = pkg._lib
    lib = L.load()
    t = torch.zeros(8, device="cuda"); g = torch.zeros(8, device="cuda")
    grp = (L.AdamGroup * 1)(L.AdamGroup(t.data_ptr(), g.data_ptr(), t.data_ptr(), t.data_ptr(), 8, 0.1, 0))
    assert lib.gsr_adam_step(grp, 1, 0.9, 0.999, 1e-15, None) == L.GSR_E_INVALID_ARG   # step counts from 1
    assert lib.gsr_adam_step(grp, 9, 0.9, 0.999, 1e-15, None) == L.GSR_E_INVALID_ARG   # too many groups
    grp[0].count = 0
    assert lib.gsr_adam_step(grp, 1, 0.9, 0.999, 1e-15, None) == L.GSR_OK              # training.jl:770: empty -> skip
    empty = torch.zeros((0, 3), device="cuda")
    o = pkg.optim.Adam(empty, 0.1)
    o.step(empty, empty.clone())
    assert o.current_step == 0
    with pytest.raises(ValueError):
        pkg.optim.Adam(torch.zeros(4), 0.1)  # CPU tensor: no CPU path
    assert lib.gsr_prologue_forward(4, 0, 2, None, None, None, None, None, None, None, None) == L.GSR_E_INVALID_ARG


@pytest.mark.parametrize("n,p", [(1, 1.0), (1023, 0.5), (1024, 0.0), (100_003, 0.3), (2_000_001, 0.93)])
def test_mask_findall_vs_oracle(pkg, orc, n, p):
    """findall(mask): ascending, exact, any block boundary."""
    r = np.random.default_rng(n)
    mask = r.uniform(size=n) < p
    got = pkg.densification.findall(dev(mask, torch.bool))
    assert np.array_equal(got.cpu().numpy(), orc.findall(mask))


def test_prune_all_parameter_arrays_bit_exact(pkg, orc):
    """prune_points! (densification.jl:138-191): the same valid_mask applied to the six parameters,
    their twelve Adam moments, the three densification statistics and the ids — 22 arrays,
    three launches."""
    r = np.random.default_rng(77)
    n = 50_021
    shapes = [(n, 3), (n, 1, 3), (n, 15, 3), (n, 1), (n, 3), (n, 4)]
    arrs = [r.normal(size=s).astype(np.float32) for s in shapes]
    arrs += [r.normal(size=int(np.prod(s))).astype(np.float32).reshape(s) for s in shapes for _ in range(2)]  # mu, nu
    arrs += [r.integers(0, 50, n).astype(np.int32), r.normal(size=n).astype(np.float32), r.normal(size=n).astype(np.float32)]
    arrs += [np.arange(n, dtype=np.int32)]
    mask = r.uniform(size=n) < 0.8
    out = pkg.densification.prune([dev(a, torch.int32 if a.dtype == np.int32 else torch.float32) for a in arrs],
                                  dev(mask, torch.bool))
    idx = orc.findall(mask)
    for o, a in zip(out, arrs):
        assert np.array_equal(o.cpu().numpy(), orc.select_rows(a, idx))


def test_select_arbitrary_indices_and_errors(pkg, orc):
    """x[:, idxs] with repeated / unordered indices (MCMC relocation samples with replacement)."""
    r = np.random.default_rng(5)
    x = r.normal(size=(1000, 15, 3)).astype(np.float32)
    idx = r.integers(0, 1000, 4321).astype(np.int32)
    (y,) = pkg.densification.select([dev(x)], dev(idx, torch.int32))
    assert np.array_equal(y.cpu().numpy(), orc.select_rows(x, idx))
    (e,) = pkg.densification.select([dev(x)], torch.empty(0, dtype=torch.int32, device="cuda"))
    assert e.shape == (0, 15, 3)
    with pytest.raises(ValueError):
        pkg.densification.findall(torch.zeros(4, dtype=torch.bool))  # CPU tensor
    L = pkg._lib
    assert L.load().gsr_gather_rows(None, 9, None, 1, None) == L.GSR_E_INVALID_ARG


def test_ply_scene_renders_like_the_arrays_it_was_exported_from(pkg, orc, tmp_path):
    """export_ply -> import_ply -> functor: the image of the reloaded scene is bit-identical."""
    W, H, deg, n = 96, 64, 2, 700
    s = pkg.synthetic.make_scene(n, W, H, deg, 5, sigma_px=4.0)
    gm = pkg.ply.GaussianModel(s.means, s.shs[:, :1].copy(), s.shs[:, 1:].copy(), s.scales_raw, s.rotations,
                               s.opacities_raw.reshape(-1, 1), deg, deg)
    path = str(tmp_path / "scene.ply")
    pkg.ply.export_ply(gm, path)
    g2 = pkg.ply.import_ply(path)
    cam = pkg.Camera(W, H, tuple(s.focal))
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb")
    img1 = rast(dev(gm.points), dev(gm.opacities), dev(gm.scales), dev(gm.rotations), dev(gm.features_dc),
                dev(gm.features_rest), camera=cam, sh_degree=deg).clone()
    img2 = rast(dev(g2.points), dev(g2.opacities), dev(g2.scales), dev(g2.rotations), dev(g2.features_dc),
                dev(g2.features_rest), camera=cam, sh_degree=g2.max_sh_degree)
    assert img1.abs().max() > 0 and torch.equal(img1, img2)


@pytest.mark.parametrize("kr,iso", [(15, False), (0, False), (3, True)])
def test_fused_trainer_tail_equals_the_three_kernels_and_the_oracle(pkg, orc, kr, iso):
    """gsr_trainer_tail_step == gsr_prologue_backward + gsr_adam_step (+ gsr_prologue_forward of the
    updated parameters): θ, μ, ν bit-identical, over 3 steps; and equal to the oracle chain."""
    r = np.random.default_rng(41)
    n = 777
    O, R = pkg.optim, pkg.rasterizer
    shapes = dict(points=(n, 3), features_dc=(n, 1, 3), features_rest=(n, kr, 3), opacities=(n, 1),
                  scales=(n, 1 if iso else 3), rotations=(n, 4))
    lrs = dict(points=1.6e-4, features_dc=2.5e-3, features_rest=2.5e-3 / 20, opacities=2.5e-2, scales=5e-3, rotations=1e-3)
    host = {k: r.normal(size=s).astype(np.float32) for k, s in shapes.items()}
    raw_a = {k: dev(v) for k, v in host.items()}          # fused path
    raw_b = {k: dev(v) for k, v in host.items()}          # three kernels
    opt_a = {k: O.Adam(raw_a[k], lrs[k], eps=1e-15) for k in O.GROUPS}
    opt_b = {k: O.Adam(raw_b[k], lrs[k], eps=1e-15) for k in O.GROUPS}
    th_o = {k: v.copy() for k, v in host.items()}         # oracle
    mu_o = {k: np.zeros(v.size, np.float32) for k, v in host.items()}
    nu_o = {k: np.zeros(v.size, np.float32) for k, v in host.items()}
    rest_or_none = lambda d: d["features_rest"] if kr else None  # noqa: E731
    shs, oa, sa = R.prologue_forward(raw_a["features_dc"], rest_or_none(raw_a), raw_a["opacities"], raw_a["scales"])
    for step in range(1, 4):
        g = dict(vmeans=r.normal(size=(n, 3)), vshs=r.normal(size=(n, 1 + kr, 3)), vopacities=r.normal(size=(n, 1)),
                 vscales=r.normal(size=(n, 3)), vrot=r.normal(size=(n, 4)))
        g = {k: v.astype(np.float32) for k, v in g.items()}
        gd = {k: dev(v) for k, v in g.items()}
        # --- three kernels (activated copies of THIS step's parameters) ---
        shs_b, oa_b, sa_b = R.prologue_forward(raw_b["features_dc"], rest_or_none(raw_b), raw_b["opacities"], raw_b["scales"])
        assert torch.equal(shs_b, shs) and torch.equal(oa_b, oa) and torch.equal(sa_b, sa)  # fused left them ready
        vdc, vrest, vo, vs = R.prologue_backward(oa_b, sa_b, gd["vshs"], gd["vopacities"], gd["vscales"], 1 if iso else 3)
        names = [k for k in O.GROUPS if raw_b[k].numel()]
        gmap = dict(points=gd["vmeans"], features_dc=vdc, features_rest=vrest, opacities=vo, scales=vs, rotations=gd["vrot"])
        O.step_all([opt_b[k] for k in names], [raw_b[k] for k in names], [gmap[k] for k in names])
        # --- oracle chain ---
        oa_h, sa_h = oa_b.cpu().numpy(), sa_b.cpu().numpy()   # same activated values as the device used
        o_vdc, o_vrest, o_vo, o_vs = orc.prologue_backward(oa_h, sa_h, g["vshs"], g["vopacities"], g["vscales"], 1 if iso else 3)
        omap = dict(points=g["vmeans"], features_dc=o_vdc, features_rest=o_vrest, opacities=o_vo, scales=o_vs, rotations=g["vrot"])
        for k in names:
            orc.adam_step(th_o[k].reshape(-1), np.ascontiguousarray(omap[k]).reshape(-1), mu_o[k], nu_o[k], step, lrs[k],
                          0.9, 0.999, 1e-15)
        # --- fused ---
        O.trainer_tail_step(opt_a, raw_a, gd, shs, oa, sa)
        torch.cuda.synchronize()
        for k in names:
            assert torch.equal(raw_a[k], raw_b[k]), (k, step)
            assert torch.equal(opt_a[k].mu, opt_b[k].mu) and torch.equal(opt_a[k].nu, opt_b[k].nu), (k, step)
            assert opt_a[k].current_step == step
            assert np.array_equal(raw_a[k].cpu().numpy(), th_o[k]), (k, step)
            assert np.array_equal(opt_a[k].mu.cpu().numpy(), mu_o[k]) and np.array_equal(opt_a[k].nu.cpu().numpy(), nu_o[k])
    if not kr:
        assert opt_a["features_rest"].current_step == 0   # empty group skipped (training.jl:770)


@pytest.mark.parametrize("kr,iso,deg,V", [(15, False, 3, 3), (15, False, 1, 2), (0, False, 0, 4), (3, True, 1, 1)])
def test_multi_view_tail_equals_rebuild_then_tail_step(pkg, kr, iso, deg, V):
    """gsr_sh_grad_from_views_tail (the multi-GPU trainer step after the factored exchange) ==
    gsr_sh_grad_from_views + gsr_trainer_tail_step: θ, μ, ν and the activated copies bit-identical over 3 steps — the
    (N,K,3) ∇shs is never materialised on the fused side.  Views with culled Gaussians (zero cotangents), an active degree
    below the stored one, isotropic scales, no higher bands."""
    r = np.random.default_rng(43)
    n, K = 1031, 1 + kr
    O, R = pkg.optim, pkg.rasterizer
    shapes = dict(points=(n, 3), features_dc=(n, 1, 3), features_rest=(n, kr, 3), opacities=(n, 1),
                  scales=(n, 1 if iso else 3), rotations=(n, 4))
    lrs = dict(points=1.6e-4, features_dc=2.5e-3, features_rest=2.5e-3 / 20, opacities=2.5e-2, scales=5e-3, rotations=1e-3)
    host = {k: r.normal(size=s).astype(np.float32) for k, s in shapes.items()}
    host["points"][:, 2] += 6.0
    raw_a = {k: dev(v) for k, v in host.items()}          # fused: rebuild + tail in one pass
    raw_b = {k: dev(v) for k, v in host.items()}          # rebuild, then gsr_trainer_tail_step
    opt_a = {k: O.Adam(raw_a[k], lrs[k], eps=1e-15) for k in O.GROUPS}
    opt_b = {k: O.Adam(raw_b[k], lrs[k], eps=1e-15) for k in O.GROUPS}
    rest = lambda d: d["features_rest"] if kr else None  # noqa: E731
    act_a = list(R.prologue_forward(raw_a["features_dc"], rest(raw_a), raw_a["opacities"], raw_a["scales"]))
    act_b = list(R.prologue_forward(raw_b["features_dc"], rest(raw_b), raw_b["opacities"], raw_b["scales"]))
    centers = dev(r.normal(size=(V, 3)).astype(np.float32))
    for step in range(1, 4):
        vc = r.normal(size=(V, n, 3)).astype(np.float32)
        vc[r.random((V, n)) < 0.2] = 0.0                  # culled in that view
        small = dict(vmeans=r.normal(size=(n, 3)), vopacities=r.normal(size=(n, 1)), vscales=r.normal(size=(n, 3)),
                     vrot=r.normal(size=(n, 4)))
        small = {k: dev(v.astype(np.float32)) for k, v in small.items()}
        vcd = dev(vc)
        vshs = R.sh_grad_from_views(raw_b["points"], vcd, centers, K, deg)
        O.trainer_tail_step(opt_b, raw_b, dict(small, vshs=vshs), *act_b)
        O.sh_views_tail_step(opt_a, raw_a, small, vcd, centers, deg, *act_a)
        torch.cuda.synchronize()
        for k in O.GROUPS:
            if not raw_a[k].numel():
                continue
            assert torch.equal(raw_a[k], raw_b[k]), (k, step)
            assert torch.equal(opt_a[k].mu, opt_b[k].mu) and torch.equal(opt_a[k].nu, opt_b[k].nu), (k, step)
            assert opt_a[k].current_step == step
        for a, b in zip(act_a, act_b):
            assert torch.equal(a, b), step
    assert float((raw_a["features_dc"] - dev(host["features_dc"])).abs().max()) > 0
    # argument checking: a null gradient is an error, not a crash
    import ctypes as C
    L = pkg._lib
    st, _ = O.tail_state(opt_a, raw_a, *act_a)
    tg = L.TailGrads(None, None, None, None, None)
    assert L.load().gsr_sh_grad_from_views_tail(n, K, deg, V, centers.data_ptr(), vcd.data_ptr(), C.byref(tg), C.byref(st),
                                                None) == L.GSR_E_INVALID_ARG


def test_end_to_end_fit_reduces_the_loss(pkg):
    """The whole chain a trainer step runs — functor prologue, rasterize, L1/DSSIM loss head, ∇rasterize,
    fused trainer tail — fits a perturbed scene back towards the image it was rendered from."""
    W, H, deg, n = 96, 64, 1, 600
    s = pkg.synthetic.make_scene(n, W, H, deg, 123, sigma_px=4.0)
    cam = pkg.Camera(W, H, tuple(s.focal))
    R, O = pkg.rasterizer, pkg.optim
    rast = R.GaussianRasterizer(W, H, mode="rgb")
    gt = [dev(s.means), dev(s.shs[:, :1].copy()), dev(s.shs[:, 1:].copy()), dev(s.opacities_raw.reshape(-1, 1)),
          dev(s.scales_raw), dev(s.rotations)]
    shs, oa, sa = R.prologue_forward(gt[1], gt[2], gt[3], gt[4])
    target_hwc = rast.forward_raw(gt[0], shs, oa, sa, gt[5], cam, deg, (0, 0, 0)).clone()
    target = target_hwc.permute(2, 0, 1).contiguous()            # (3,H,W) == Julia (W,H,3)
    rng = np.random.default_rng(5)
    raw = dict(points=dev(s.means + rng.normal(0, 0.01, s.means.shape).astype(np.float32)),
               features_dc=dev(s.shs[:, :1] + rng.normal(0, 0.3, (n, 1, 3)).astype(np.float32)),
               features_rest=dev(s.shs[:, 1:] * 0.5),
               opacities=dev(s.opacities_raw.reshape(-1, 1) - 0.5), scales=dev(s.scales_raw + 0.1),
               rotations=dev(s.rotations))
    lrs = dict(points=1.6e-4, features_dc=2.5e-2, features_rest=2.5e-3, opacities=5e-2, scales=5e-3, rotations=1e-3)
    opts = {k: O.Adam(raw[k], lrs[k], eps=1e-15) for k in O.GROUPS}
    shs, oa, sa = R.prologue_forward(raw["features_dc"], raw["features_rest"], raw["opacities"], raw["scales"])
    losses = []
    for it in range(60):
        img = rast.forward_raw(raw["points"], shs, oa, sa, raw["rotations"], cam, deg, (0, 0, 0))
        loss, vp = pkg.fused_ssim.l1_ssim_loss(rast, img, target)
        vm, vsh, vo, vsc, vr, _, _ = rast.backward_raw(vp, raw["points"], shs, oa, sa, raw["rotations"], cam, deg, (0, 0, 0))
        O.trainer_tail_step(opts, raw, dict(vmeans=vm, vshs=vsh, vopacities=vo, vscales=vsc, vrot=vr), shs, oa, sa)
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    assert np.mean(losses[-5:]) < np.mean(losses[:5])


def test_ply_device_pack_unpack_matches_host_io(pkg, tmp_path):
    """SURVEY.md §8f rank 4 on the device: gsr_ply_pack_rows / gsr_ply_unpack_rows against the host (numpy) forms of
    export_ply / import_ply (gaussians.jl:157-247) — byte-identical files, bit-identical arrays, degree 0 and 3."""
    for n, kr in ((1000, 15), (257, 0)):
        rng = np.random.default_rng(n)
        f = lambda *s: rng.normal(size=s).astype(np.float32)  # noqa: E731
        deg = int(round(np.sqrt(kr + 1))) - 1
        g = pkg.ply.GaussianModel(f(n, 3), f(n, 1, 3), f(n, kr, 3), f(n, 3), f(n, 4), f(n, 1), deg, deg)
        gd = pkg.densification.GaussianModel(dev(g.points), dev(g.features_dc), dev(g.features_rest), dev(g.scales), dev(g.rotations),
                                             dev(g.opacities))
        a, b = str(tmp_path / f"host_{n}.ply"), str(tmp_path / f"dev_{n}.ply")
        pkg.ply.export_ply(g, a)
        pkg.ply.export_ply_device(gd, b)
        assert open(a, "rb").read() == open(b, "rb").read()
        pts, dc, rest, sc, rot, op, d = pkg.ply.import_ply_device(a)
        assert d == deg
        for t, ref in ((pts, g.points), (dc, g.features_dc), (rest, g.features_rest), (sc, g.scales), (rot, g.rotations), (op, g.opacities)):
            assert np.array_equal(t.cpu().numpy(), ref)


def test_nonfinite_gradient_report(pkg):
    """gsr_count_nonfinite against torch: the GSP_DEBUG guard of step! (training.jl:772-777) and the per-parameter counts of
    nonfinite_gradient_report (:534-552)."""
    n = 5000
    g = torch.Generator(device="cuda").manual_seed(3)
    names = ["points", "features_dc", "features_rest", "opacities", "scales", "rotations"]
    grads = [torch.randn((n, 3), device="cuda", generator=g), torch.randn((n, 1, 3), device="cuda", generator=g),
             torch.randn((n, 15, 3), device="cuda", generator=g), torch.randn((n, 1), device="cuda", generator=g),
             torch.randn((n, 3), device="cuda", generator=g), torch.randn((n, 4), device="cuda", generator=g)]
    assert pkg.optim.nonfinite_gradient_report(names, grads, n) == {}
    grads[2][[17, 4000, 4000, 4999], [3, 0, 14, 7], [1, 2, 0, 2]] = torch.tensor([float("nan"), float("inf"), float("nan"), float("-inf")], device="cuda")
    grads[5][123, 3] = float("inf")
    grads.append(torch.empty((n, 0, 3), device="cuda"))  # an empty features_rest-like array is skipped (training.jl:770)
    rep = pkg.optim.nonfinite_gradient_report(names + ["empty"], grads, n)
    assert rep == {"features_rest": (3, 17), "rotations": (1, 123)}
    for nm, t in zip(names, grads):
        bad = (~torch.isfinite(t.reshape(n, -1))).any(1)
        assert int(bad.sum()) == rep.get(nm, (0, 0))[0]


@pytest.mark.parametrize("n,deg,max_deg,iso,mode,color", [(600, 1, 1, False, "rgb", False), (1500, 3, 3, False, "rgb", False),
                                                           (777, 1, 3, False, "rgbd", False), (300, 0, 0, False, "rgb", False),
                                                           (513, 2, 2, True, "rgb", False), (1100, 3, 3, False, "rgbdn", False),
                                                           (777, 1, 3, False, "rgbd", True), (1100, 3, 3, False, "rgbdn", True)])
def test_backward_with_the_tail_in_its_epilogue_equals_backward_then_tail(pkg, n, deg, max_deg, iso, mode, color):
    """gsr_backward_trainer_tail == gsr_backward + gsr_trainer_tail_step, bit for bit (θ, μ, ν, the
    activated copies, gstate.∇means_2d), over 4 training steps: culled Gaussians (zero gradient, moments
    still decay), an active SH degree below the stored one, isotropic scales, no features_rest, n % 256 != 0,
    every render mode; `color`: a cotangent with zeros above the colour channels, announced to both paths
    (GSR_GRADS_COLOR_COTANGENT: the :rgb arithmetic on the mode's stream)."""
    W, H = 112, 80
    s = pkg.synthetic.make_scene(n, W, H, max_deg, 77, sigma_px=5.0)
    cam = pkg.Camera(W, H, tuple(s.focal))
    R, O = pkg.rasterizer, pkg.optim
    C = pkg._lib.MODES[mode]
    rng = np.random.default_rng(3)
    target = dev(rng.uniform(0, 1, (H, W, C)).astype(np.float32))
    scales_raw = s.scales_raw[:, :1].copy() if iso else s.scales_raw
    s.means[::7, 2] = -1.0   # behind the camera: culled, zero gradient
    host = dict(points=s.means, features_dc=s.shs[:, :1].copy(), features_rest=s.shs[:, 1:].copy(),
                opacities=s.opacities_raw.reshape(-1, 1), scales=scales_raw, rotations=s.rotations)
    lrs = dict(points=1.6e-3, features_dc=2.5e-2, features_rest=2.5e-3, opacities=5e-2, scales=5e-3, rotations=1e-2)
    bg = (0.1, 0.2, 0.3)

    def make():
        raw = {k: dev(np.ascontiguousarray(v)) for k, v in host.items()}
        opts = {k: O.Adam(raw[k], lrs[k], eps=1e-15) for k in O.GROUPS}
        rest = raw["features_rest"] if raw["features_rest"].numel() else None
        act = list(R.prologue_forward(raw["features_dc"], rest, raw["opacities"], raw["scales"]))
        return raw, opts, act, R.GaussianRasterizer(W, H, mode=mode)

    raw_a, opt_a, act_a, rast_a = make()   # backward, then the tail
    raw_b, opt_b, act_b, rast_b = make()   # the tail in the epilogue
    for step in range(1, 5):
        img_a = rast_a.forward_raw(raw_a["points"], *act_a, raw_a["rotations"], cam, deg, bg)
        img_b = rast_b.forward_raw(raw_b["points"], *act_b, raw_b["rotations"], cam, deg, bg)
        assert torch.equal(img_a, img_b), step
        if color:
            # the flag is only valid for the cotangent the handle's OWN loss head wrote for this forward (ABI 6): one each
            tgt3 = target[:, :, :3].permute(2, 0, 1).contiguous()
            _, vp = pkg.fused_ssim.l1_ssim_loss(rast_a, img_a, tgt3)
            _, vp_b = pkg.fused_ssim.l1_ssim_loss(rast_b, img_b, tgt3)
            assert torch.equal(vp, vp_b) and not vp[:, :, 3:].any()
        else:
            vp = (img_a - target) * (2.0 / img_a.numel())     # any cotangent will do; the same one on both sides
            vp_b = vp.clone()
        vm, vsh, vo, vsc, vr, _, _ = rast_a.backward_raw(vp, raw_a["points"], *act_a, raw_a["rotations"], cam, deg, bg,
                                                         color_cotangent=color)
        O.trainer_tail_step(opt_a, raw_a, dict(vmeans=vm, vshs=vsh, vopacities=vo, vscales=vsc, vrot=vr), *act_a)
        O.fused_backward_tail_step(rast_b, vp_b, opt_b, raw_b, *act_b, cam, deg, bg,
                                   forward_generation=rast_b.stats.generation, color_cotangent=color)
        torch.cuda.synchronize()
        assert (rast_a.gstate.radii <= 0).any() and (rast_a.gstate.radii > 0).any()
        for k in O.GROUPS:
            if not raw_a[k].numel():
                continue
            assert torch.equal(raw_a[k], raw_b[k]), (k, step)
            assert torch.equal(opt_a[k].mu, opt_b[k].mu) and torch.equal(opt_a[k].nu, opt_b[k].nu), (k, step)
            assert opt_b[k].current_step == step
        for x, y in zip(act_a, act_b):
            assert torch.equal(x, y), step
        assert torch.equal(rast_a.gstate.grad_means_2d, rast_b.gstate.grad_means_2d), step
    assert not torch.equal(raw_b["points"], dev(host["points"]))   # it did train


def test_fused_step_checks_its_contract(pkg):
    """The fused step updates the forward's inputs in place: foreign arrays are refused, and so is a second
    backward on the same forward."""
    W, H, n, deg = 64, 48, 200, 1
    s = pkg.synthetic.make_scene(n, W, H, deg, 5, sigma_px=4.0)
    cam = pkg.Camera(W, H, tuple(s.focal))
    R, O, L = pkg.rasterizer, pkg.optim, pkg._lib
    raw = dict(points=dev(s.means), features_dc=dev(s.shs[:, :1].copy()), features_rest=dev(s.shs[:, 1:].copy()),
               opacities=dev(s.opacities_raw.reshape(-1, 1)), scales=dev(s.scales_raw), rotations=dev(s.rotations))
    opts = {k: O.Adam(raw[k], 1e-3, eps=1e-15) for k in O.GROUPS}
    shs, oa, sa = R.prologue_forward(raw["features_dc"], raw["features_rest"], raw["opacities"], raw["scales"])
    rast = R.GaussianRasterizer(W, H, mode="rgb")
    img = rast.forward_raw(raw["points"], shs, oa, sa, raw["rotations"], cam, deg, (0, 0, 0))
    vp = torch.ones_like(img)
    st, _ = O.tail_state(opts, raw, shs.clone(), oa, sa)                # NOT the array the forward was given
    with pytest.raises(L.GsrError, match="trainer's own arrays"):
        rast.backward_trainer_tail(vp, st, raw["points"], shs, oa, sa, raw["rotations"], cam, deg, (0, 0, 0))
    assert all(o.current_step == 0 for o in opts.values())          # a refused call moves no counter
    with pytest.raises(L.GsrError, match="forward #"):
        O.fused_backward_tail_step(rast, vp, opts, raw, shs, oa, sa, cam, deg, (0, 0, 0),
                                   forward_generation=rast.stats.generation + 1)
    O.fused_backward_tail_step(rast, vp, opts, raw, shs, oa, sa, cam, deg, (0, 0, 0))
    with pytest.raises(L.GsrError, match="updated in place"):
        rast.backward_raw(vp, raw["points"], shs, oa, sa, raw["rotations"], cam, deg, (0, 0, 0))
    assert rast.gstate.radii.numel() == n                             # the forward's outputs stay readable
