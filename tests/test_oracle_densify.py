"""CPU tests of the densification restatement (oracle/densify.py) against the behaviour the reference's code and
tests pin (src/densification.jl, src/strategy.jl:78-136, src/gaussians.jl:115-137; reference tests
runtests.jl "densification" checks sizes, finiteness and optimizer-state shapes after densify_and_prune!)."""
import numpy as np
import pytest

from oracle import densify as dz

f32 = np.float32


def make_model(n, k_rest, seed, scale_dims=3):
    rng = np.random.default_rng(seed)
    return dz.Model(rng.normal(size=(n, 3)).astype(f32), rng.normal(size=(n, 1, 3)).astype(f32),
                    rng.normal(size=(n, k_rest, 3)).astype(f32), rng.normal(-3.0, 1.0, size=(n, scale_dims)).astype(f32),
                    rng.normal(size=(n, 4)).astype(f32), rng.normal(-1.0, 2.0, size=(n, 1)).astype(f32))


def fill_stats(st, seed):
    rng = np.random.default_rng(seed)
    n = st.denom.shape[0]
    st.denom[:] = rng.integers(0, 4, n).astype(f32)          # zeros -> 0/0 = NaN -> 0 (densification.jl:7-10)
    st.accum_grad_means_2d[:] = (rng.gamma(2.0, 2e-4, n) * st.denom).astype(f32)
    st.max_radii[:] = rng.integers(0, 40, n).astype(np.int32)


@pytest.mark.parametrize("scale_dims,k_rest", [(3, 15), (1, 3), (3, 0)])
def test_densify_and_prune_bookkeeping(scale_dims, k_rest):
    n = 400
    gs = make_model(n, k_rest, 1, scale_dims)
    before = gs.copy()
    st = dz.Strategy.for_model(n)
    fill_stats(st, 2)
    opt = dz.new_optimizers(gs)
    rng = np.random.default_rng(3)
    for k in dz.PARAMS:
        opt[k]["mu"][:] = rng.normal(size=opt[k]["mu"].shape); opt[k]["nu"][:] = rng.uniform(size=opt[k]["nu"].shape)
    mu_before = {k: opt[k]["mu"].reshape(getattr(before, k).shape).copy() for k in dz.PARAMS}
    with np.errstate(invalid="ignore", divide="ignore"):
        grad = np.nan_to_num((st.accum_grad_means_2d / st.denom).astype(f32), nan=0.0)
    extent = 5.0
    masks = dz.densify_and_prune(st, gs, opt, extent, extent, 0, seed=7)
    mc, ms, valid = masks["clone"], masks["split"], masks["valid"]
    gamma = f32(extent) * f32(st.dense_percent)
    assert np.array_equal(mc, (grad > f32(2e-4)) & (dz.max_exp_scale(before.scales) < gamma))
    assert mc.sum() > 0 and ms[:n].sum() > 0 and not ms[n:].any(), "clones have zero gradient: never split"
    assert not (mc & ms[:n]).any()
    n1, m = n + mc.sum(), ms.sum()
    assert valid.shape[0] == n1 - m + 2 * m and len(gs) == valid.sum()
    for k in dz.PARAMS:
        x = getattr(gs, k)
        assert x.shape[0] == len(gs) and np.isfinite(x).all()
        assert opt[k]["mu"].shape == (x.size,) and opt[k]["nu"].shape == (x.size,)
    assert st.max_radii.shape == (len(gs),) and not st.max_radii.any() and not st.denom.any()
    # rows that were neither split nor pruned keep parameters and moments; appended rows have zero moments
    keep_old = np.flatnonzero(~ms[:n])                        # survivors of the split among the originals
    pos = np.cumsum(valid) - 1
    alive = valid[:keep_old.shape[0]]
    assert np.array_equal(gs.points[pos[:keep_old.shape[0]][alive]], before.points[keep_old[alive]])
    mu_pts = opt["points"]["mu"].reshape(-1, 3)
    assert np.array_equal(mu_pts[pos[:keep_old.shape[0]][alive]], mu_before["points"][keep_old[alive]])
    first_new = n - ms[:n].sum()                              # clones + split children start here (before the prune)
    new_alive = valid[first_new:]
    assert not mu_pts[pos[first_new:][new_alive]].any()


def test_split_children_are_sampled_inside_the_parent_and_shrunk():
    n = 3000
    gs = make_model(n, 0, 11)
    gs.scales[:] = np.log(f32(0.5))                           # everything is "big": all split
    before = gs.copy()
    st = dz.Strategy.for_model(n)
    st.denom[:] = 1; st.accum_grad_means_2d[:] = 1.0
    opt = dz.new_optimizers(gs)
    grad = np.ones(n, f32)
    dz.densify_split(st, gs, opt, grad, 2e-4, 5.0, 1e-2, seed=5)
    assert len(gs) == 2 * n
    assert np.allclose(gs.scales, np.log(f32(0.5) / f32(1.6)), rtol=1e-6)
    assert np.array_equal(gs.rotations[:n], before.rotations) and np.array_equal(gs.rotations[n:], before.rotations)
    d = np.concatenate([gs.points[:n] - before.points, gs.points[n:] - before.points])  # block repeat: [A A]
    R = dz.unnorm_quat2rot(np.concatenate([before.rotations, before.rotations]))
    local = np.einsum("nji,nj->ni", R, d) / 0.5               # R' * offset / sigma ~ N(0, I)
    assert abs(local.mean()) < 0.03 and abs(local.std() - 1.0) < 0.03
    assert abs(np.corrcoef(local[:, 0], local[:, 1])[0, 1]) < 0.05
    assert np.abs(local[:n] - local[n:]).mean() > 0.5, "the two children draw different noise"


def test_reset_opacity_and_schedule():
    gs = make_model(50, 0, 21)
    o0 = dz.sigmoid(gs.opacities)
    dz.reset_opacity(gs)
    assert np.allclose(dz.sigmoid(gs.opacities), np.minimum(o0, 0.1), atol=1e-6)
    # post_train_step!: stats always (until densify_until_iter), densify on the interval from densify_from_iter,
    # max_screen_size only after the first opacity reset, reset + NU.reset! on its interval (strategy.jl:78-105)
    st = dz.Strategy.for_model(50, densify_from_iter=4, densify_until_iter=9, densification_interval=2, opacity_reset_interval=6)
    opt = dz.new_optimizers(gs)
    radii = np.full(50, 5, np.int32); g2 = np.full((50, 2), 1e-6, f32)
    log = []
    for step in range(1, 12):
        opt["opacities"]["step"] = 3
        n0 = len(gs)
        radii = np.full(n0, 5, np.int32); g2 = np.full((n0, 2), 1e-9, f32)
        log.append(dz.post_train_step(st, gs, opt, radii, g2, (64, 48), step, 5.0))
        if log[-1][1]:
            assert opt["opacities"]["step"] == 0 and not opt["opacities"]["mu"].any()
    assert [d for d, _ in log] == [False, False, False, True, False, True, False, True, False, False, False]
    assert [r for _, r in log] == [False] * 5 + [True] + [False] * 5
