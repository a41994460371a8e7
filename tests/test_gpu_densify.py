"""-m gpu parity tests of the device densification (gsr_densify_* / gsr_compose_rows / gsr_split_transform /
gsr_reset_opacity through the host mirror densification.py) against the CPU restatement oracle/densify.py:
masks, indices, row order, Adam moments and statistics EXACT; parameters bit-equal except the split children's
points / scales (device exp / log / cos / sin vs glibc: 1e-6 relative)."""
import numpy as np
import pytest
import torch

from hip_helpers import dev, rel_l2
from oracle import densify as dz
from test_oracle_densify import fill_stats, make_model

pytestmark = pytest.mark.gpu
f32 = np.float32


def to_device(pkg, m: dz.Model):
    return pkg.densification.GaussianModel(*[dev(getattr(m, k)) for k in dz.PARAMS])


def device_optimizers(pkg, gs_d, opt_o):
    out = {}
    for k in dz.PARAMS:
        a = pkg.optim.Adam(getattr(gs_d, k), 1e-3, eps=1e-15)
        a.mu, a.nu = dev(opt_o[k]["mu"]), dev(opt_o[k]["nu"])
        out[k] = a
    return out


@pytest.mark.parametrize("scale_dims,k_rest,mss", [(3, 15, 0), (1, 3, 20), (3, 0, 20)])
def test_densify_and_prune_matches_oracle(pkg, scale_dims, k_rest, mss):
    Dz = pkg.densification
    n, extent, seed = 5000, 5.0, 99
    gs_o = make_model(n, k_rest, 31, scale_dims)
    st_o = dz.Strategy.for_model(n)
    fill_stats(st_o, 32)
    opt_o = dz.new_optimizers(gs_o)
    rng = np.random.default_rng(33)
    for k in dz.PARAMS:
        opt_o[k]["mu"][:] = rng.normal(size=opt_o[k]["mu"].shape); opt_o[k]["nu"][:] = rng.uniform(size=opt_o[k]["nu"].shape)
    gs_d = to_device(pkg, gs_o)
    # `gs.ids` (use_ids, gaussians.jl:14,46): labels that ride along — here the original index, so the result says where
    # every surviving row came from
    ids0 = np.arange(n, dtype=np.int32) * 3 + 1
    gs_d.ids = dev(ids0, torch.int32)
    opt_d = device_optimizers(pkg, gs_d, opt_o)
    st_d = Dz.DefaultStrategy(gs_d)
    st_d.max_radii, st_d.accum_grad_means_2d, st_d.denom = dev(st_o.max_radii, torch.int32), dev(st_o.accum_grad_means_2d), dev(st_o.denom)
    masks_o = dz.densify_and_prune(st_o, gs_o, opt_o, extent, extent, mss, seed=seed)
    mc, ms, valid = Dz.densify_and_prune(st_d, gs_d, opt_d, extent, extent, mss, seed=seed)
    torch.cuda.synchronize()
    assert np.array_equal(mc.cpu().numpy().astype(bool), masks_o["clone"])
    assert np.array_equal(ms.cpu().numpy().astype(bool), masks_o["split"])
    assert np.array_equal(valid.cpu().numpy().astype(bool), masks_o["valid"])
    assert masks_o["clone"].sum() > 50 and masks_o["split"].sum() > 50 and (~masks_o["valid"]).sum() > 50
    assert len(gs_d) == len(gs_o)
    # which surviving rows are split children (their points / scales went through device transcendentals)
    n1 = n + masks_o["clone"].sum()
    n_keep = n1 - masks_o["split"].sum()
    child = np.zeros(masks_o["valid"].shape[0], bool); child[n_keep:] = True
    child = child[masks_o["valid"]]
    for k in dz.PARAMS:
        a, b = getattr(gs_d, k).cpu().numpy(), getattr(gs_o, k)
        assert a.shape == b.shape, k
        if k in ("points", "scales"):
            assert np.array_equal(a[~child], b[~child]), k
            assert np.allclose(a[child], b[child], rtol=2e-6, atol=2e-6), k
        else:
            assert np.array_equal(a, b), k
        assert np.array_equal(opt_d[k].mu.cpu().numpy(), opt_o[k]["mu"]), k
        assert np.array_equal(opt_d[k].nu.cpu().numpy(), opt_o[k]["nu"]), k
    assert np.array_equal(st_d.max_radii.cpu().numpy(), st_o.max_radii) and st_d.denom.shape[0] == len(gs_o)
    assert not st_d.accum_grad_means_2d.any()
    # ids through the three steps, restated from the reference: clone appends gs.ids[mask] (densification.jl:47,253-257), split
    # appends repeat(gs.ids[mask], 2) and prunes the originals (:90,105-111), prune keeps gs.ids[valid_mask] (:185-188)
    ids = np.concatenate([ids0, ids0[masks_o["clone"]]])
    sp = masks_o["split"]
    ids = np.concatenate([ids[~sp], np.tile(ids[sp], 2)])
    ids = ids[masks_o["valid"]]
    got = gs_d.ids.cpu().numpy()
    assert got.dtype == np.int32 and np.array_equal(got, ids)
    with pytest.raises(ValueError):
        Dz.GaussianModel(*[getattr(gs_d, k) for k in dz.PARAMS], ids=gs_d.ids[:-1].contiguous())


def test_reset_opacity_matches_oracle(pkg):
    gs_o = make_model(3000, 0, 41)
    gs_d = to_device(pkg, gs_o)
    dz.reset_opacity(gs_o)
    pkg.densification.reset_opacity(gs_d)
    assert np.allclose(gs_d.opacities.cpu().numpy(), gs_o.opacities, rtol=2e-6, atol=2e-6)


def test_train_densify_train_chain_matches_oracle_chain(pkg, orc):
    """SURVEY.md §8f rank 3 'done when': steps of forward + loss + backward + Adam + post_train_step! through the C ABI,
    with a densification and an opacity reset in the middle, against the same chain on the oracle (orc.forward /
    loss_head / backward / prologue / adam_step + oracle/densify.py)."""
    R, O, Dz = pkg.rasterizer, pkg.optim, pkg.densification
    W, H, deg, n0 = 96, 64, 1, 500
    s = pkg.synthetic.make_scene(n0, W, H, deg, 123, sigma_px=4.0)
    cam_d, cam_o = pkg.Camera(W, H, tuple(s.focal)), orc.Camera(W, H, s.focal)
    target = pkg.synthetic.make_target(W, H, 5)
    gs_o = dz.Model(s.means.copy(), s.shs[:, :1].copy(), s.shs[:, 1:].copy(), s.scales_raw.copy(), s.rotations.copy(),
                    s.opacities_raw.reshape(-1, 1).copy())
    gs_d = to_device(pkg, gs_o)
    lrs = dict(points=1.6e-4, features_dc=2.5e-3, features_rest=2.5e-3 / 20, opacities=2.5e-2, scales=5e-3, rotations=1e-3)
    opt_o = dz.new_optimizers(gs_o)
    opt_d = {k: O.Adam(getattr(gs_d, k), lrs[k], eps=1e-15) for k in dz.PARAMS}
    kw = dict(densify_from_iter=2, densify_until_iter=100, densification_interval=2, opacity_reset_interval=3,
              densify_grad_threshold=1e-4, dense_percent=0.08)
    st_o = dz.Strategy.for_model(n0, **kw)
    st_d = Dz.DefaultStrategy(gs_d, **kw)
    rast = R.GaussianRasterizer(W, H, mode="rgb")
    tgt_d = dev(target)
    extent, sizes = 5.0, []
    for step in range(1, 5):
        # ---- oracle chain ----
        shs, oa, sa = orc.prologue_forward(gs_o.features_dc, gs_o.features_rest, gs_o.opacities, gs_o.scales)
        st = orc.forward(gs_o.points, shs, oa, sa, gs_o.rotations, cam_o, deg)
        loss_o, vp = orc.loss_head(st.image, target)
        g = orc.backward(st, vp, gs_o.points, shs, oa, sa, gs_o.rotations, cam_o, deg)
        vdc, vrest, vo, vs = orc.prologue_backward(oa, sa, g.vshs, g.vopacities.reshape(-1, 1), g.vscales, 3)
        grads = dict(points=g.vmeans, features_dc=vdc, features_rest=vrest, opacities=vo, scales=vs, rotations=g.vrots)
        for k in dz.PARAMS:
            opt_o[k]["step"] += 1
            theta = getattr(gs_o, k).reshape(-1)
            orc.adam_step(theta, np.ascontiguousarray(grads[k]).reshape(-1), opt_o[k]["mu"], opt_o[k]["nu"], opt_o[k]["step"], lrs[k],
                          0.9, 0.999, 1e-15)
        dz.post_train_step(st_o, gs_o, opt_o, st.radii, g.vmeans2d, (W, H), step, extent, seed=step)
        # ---- HIP chain, through the C ABI ----
        shs_d, oa_d, sa_d = R.prologue_forward(gs_d.features_dc, gs_d.features_rest, gs_d.opacities, gs_d.scales)
        img = rast.forward_raw(gs_d.points, shs_d, oa_d, sa_d, gs_d.rotations, cam_d, deg, (0, 0, 0))
        loss_d, vp_d = pkg.fused_ssim.l1_ssim_loss(rast, img, tgt_d)
        vm, vsh, vo_d, vsc, vr, _, _ = rast.backward_raw(vp_d, gs_d.points, shs_d, oa_d, sa_d, gs_d.rotations, cam_d, deg, (0, 0, 0))
        raw = {k: getattr(gs_d, k) for k in dz.PARAMS}
        O.trainer_tail_step(opt_d, raw, dict(vmeans=vm, vshs=vsh, vopacities=vo_d, vscales=vsc, vrot=vr), shs_d, oa_d, sa_d)
        Dz.post_train_step(st_d, gs_d, opt_d, rast, step, extent, seed=step)
        torch.cuda.synchronize()
        assert abs(float(loss_d) - float(loss_o)) < 1e-5
        assert len(gs_d) == len(gs_o), f"step {step}: {len(gs_d)} vs {len(gs_o)} Gaussians"
        sizes.append(len(gs_o))
        for k in dz.PARAMS:
            assert rel_l2(getattr(gs_d, k).cpu().numpy().reshape(-1), getattr(gs_o, k).reshape(-1)) <= 2e-5, (step, k)
            assert opt_d[k].mu.numel() == opt_o[k]["mu"].size
        assert opt_d["opacities"].current_step == opt_o["opacities"]["step"]
    assert sizes[1] != n0 and sizes[3] != sizes[2], f"densification must have changed the model: {sizes}"


def test_default_split_seed_differs_between_rounds(pkg):
    """ADVICE r2 (densification.py:208): the split noise is a pure function of (seed, row, draw), so a constant default
    seed would hand appended row i the same normal triple at every densification round.  With seed=None the strategy's
    own round counter picks the stream: two rounds on the SAME inputs must give different children (the reference draws
    fresh randn each time, densification.jl:121-135), and an explicit seed must stay reproducible."""
    Dz = pkg.densification
    n, extent = 4000, 5.0
    gs_o = make_model(n, 3, 51, 3)
    st_o = dz.Strategy.for_model(n)
    fill_stats(st_o, 52)

    def one_round(strategy, seed):
        gs_d = to_device(pkg, gs_o)
        opt_d = device_optimizers(pkg, gs_d, dz.new_optimizers(gs_o))
        strategy.max_radii, strategy.accum_grad_means_2d, strategy.denom = (dev(st_o.max_radii, torch.int32),
                                                                              dev(st_o.accum_grad_means_2d), dev(st_o.denom))
        grad = torch.empty(n, dtype=torch.float32, device="cuda")
        L = pkg._lib
        L.check(L.load().gsr_densify_grad_mean(n, strategy.accum_grad_means_2d.data_ptr(), strategy.denom.data_ptr(),
                                               grad.data_ptr(), None))
        m = Dz.densify_split(strategy, gs_d, opt_d, grad, strategy.densify_grad_threshold, extent, strategy.dense_percent, seed)
        torch.cuda.synchronize()
        assert int(m.sum()) > 50
        return gs_d.points.cpu().numpy()

    st_d = Dz.DefaultStrategy(to_device(pkg, gs_o))
    a, b = one_round(st_d, None), one_round(st_d, None)
    assert st_d.split_rounds == 2
    assert a.shape == b.shape and not np.array_equal(a, b), "two default-seeded rounds must not repeat the noise"
    c, d = one_round(st_d, 7), one_round(st_d, 7)
    assert np.array_equal(c, d), "an explicit seed is reproducible"
    assert st_d.split_rounds == 2, "explicit seeds do not advance the counter"
    # the seed base comes from the trainer's RNG seed and the round counter survives a checkpoint — a resumed run
    # continues the noise sequence instead of replaying it
    s1 = Dz.DefaultStrategy(to_device(pkg, gs_o), seed=1234)
    s2 = Dz.DefaultStrategy(to_device(pkg, gs_o), seed=1235)
    assert s1.next_split_seed() != s2.next_split_seed()
    s3 = Dz.DefaultStrategy(to_device(pkg, gs_o))
    s3.load_state_dict(s1.state_dict())
    assert s3.split_rounds == 1 and s3.next_split_seed() == s1.next_split_seed()


def _morton_ref(points, bits=21):
    """numpy restatement of gsr_morton_codes (float32 arithmetic as the kernel's)."""
    lo, hi = points.min(0), points.max(0)
    ext = (hi - lo).astype(f32)
    inv = np.where(ext > 0, f32(1.0) / np.where(ext > 0, ext, f32(1.0)), f32(0.0)).astype(f32)
    t = np.clip(((points - lo).astype(f32) * inv).astype(f32), f32(0.0), f32(1.0))
    cells = (t * f32(2 ** bits - 1)).astype(f32).astype(np.uint64)
    codes = np.zeros(len(points), np.uint64)
    for b in range(bits):
        for a in range(3):
            codes |= ((cells[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return codes


def test_reorder_spatially_is_a_consistent_permutation_and_leaves_the_render_unchanged(pkg, orc):
    """densification.reorder_spatially (not a reference function; DefaultStrategy(spatial_reorder=True) runs it after each
    densification): every per-Gaussian array — parameters, Adam moments, ids, statistics — is permuted by ONE permutation,
    which sorts the Morton codes of the positions; the rendered image and the gradients are those of the unsorted model."""
    Dz = pkg.densification
    W, H, n, deg = 192, 128, 6000, 1
    s = pkg.synthetic.make_scene(n, W, H, deg, 7300, sigma_px=4.0)
    gs_o = make_model(n, 3, 41, 3)
    gs_o.points[:] = s.means
    gs_d = to_device(pkg, gs_o)
    gs_d.ids = dev(np.arange(n, dtype=np.int32), torch.int32)
    opt_o = dz.new_optimizers(gs_o)
    rng = np.random.default_rng(42)
    for k in dz.PARAMS:
        opt_o[k]["mu"][:] = rng.normal(size=opt_o[k]["mu"].shape); opt_o[k]["nu"][:] = rng.uniform(size=opt_o[k]["nu"].shape)
    opt_d = device_optimizers(pkg, gs_d, opt_o)
    st = Dz.DefaultStrategy(gs_d, spatial_reorder=True)
    st.max_radii = dev(rng.integers(0, 50, n).astype(np.int32), torch.int32)
    before = {k: getattr(gs_d, k).cpu().numpy().copy() for k in dz.PARAMS}
    mu_before = {k: opt_d[k].mu.cpu().numpy().copy() for k in dz.PARAMS}
    radii_before = st.max_radii.cpu().numpy().copy()
    perm = Dz.reorder_spatially(st, gs_d, opt_d).cpu().numpy()
    torch.cuda.synchronize()
    assert np.array_equal(np.sort(perm), np.arange(n))
    codes = _morton_ref(before["points"].reshape(n, 3))
    assert np.array_equal(perm, np.argsort(codes, kind="stable"))
    assert np.array_equal(gs_d.ids.cpu().numpy(), perm)
    for k in dz.PARAMS:
        assert np.array_equal(getattr(gs_d, k).cpu().numpy(), before[k][perm]), k
        rw = int(np.prod(before[k].shape[1:]))
        if rw:
            assert np.array_equal(opt_d[k].mu.cpu().numpy().reshape(n, rw), mu_before[k].reshape(n, rw)[perm]), k
    assert np.array_equal(st.max_radii.cpu().numpy(), radii_before[perm])
    # the render of the scene does not care about the order of its Gaussians (no two of them at exactly the same depth here)
    from hip_helpers import HipRun
    cam = orc.Camera(W, H, s.focal)
    a = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, (0.1, 0.2, 0.3))
    b = HipRun(pkg, s.means[perm], s.shs[perm], s.opacities[perm], s.scales[perm], s.rotations[perm], cam, deg, (0.1, 0.2, 0.3))
    ia, ib = a.forward().clone(), b.forward().clone()
    assert torch.equal(ia, ib)
    vp = np.random.default_rng(5).standard_normal((H, W, 3)).astype(f32)
    ga, gb = a.backward(vp), b.backward(vp)
    p = torch.from_numpy(perm.astype(np.int64)).cuda()
    for x, y in zip(ga[:5], gb[:5]):
        assert torch.equal(x[p], y)


def test_reorder_spatially_survives_non_finite_positions(pkg, orc):
    """a NaN / Inf position (a diverging step's transient) must not abort a densification round after clone /
    split / prune have mutated the model: the bounding box is taken over the finite coordinates, the offending rows sort to
    an end, and an all-non-finite or single-point model skips the re-sort (identity permutation)."""
    Dz = pkg.densification
    n = 500
    gs_o = make_model(n, 3, 43, 3)
    gs_d = to_device(pkg, gs_o)
    opt_d = device_optimizers(pkg, gs_d, dz.new_optimizers(gs_o))
    st = Dz.DefaultStrategy(gs_d, spatial_reorder=True)
    pts = gs_d.points.reshape(n, 3)
    pts[7, 0] = float("nan"); pts[11, 2] = float("inf"); pts[13, 1] = float("-inf")
    finite_before = gs_d.points.reshape(n, 3).cpu().numpy().copy()
    perm = Dz.reorder_spatially(st, gs_d, opt_d).cpu().numpy()
    torch.cuda.synchronize()
    assert np.array_equal(np.sort(perm), np.arange(n))
    after = gs_d.points.reshape(n, 3).cpu().numpy()
    assert np.array_equal(after, finite_before[perm], equal_nan=True)
    assert not np.array_equal(perm, np.arange(n)), "the finite rows were re-sorted"
    # nothing finite at all / one point repeated: identity
    gs_d.points[:] = float("nan")
    assert np.array_equal(Dz.reorder_spatially(st, gs_d, opt_d).cpu().numpy(), np.arange(n))
    gs_d.points[:] = 1.5
    assert np.array_equal(Dz.reorder_spatially(st, gs_d, opt_d).cpu().numpy(), np.arange(n))
