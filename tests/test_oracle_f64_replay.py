"""The FLOAT64 REPLAY of the oracle's per-Gaussian backward (oracle/gsr_oracle.c compiled with -DORC_REAL_DOUBLE: the same
source text, every `float` a double; oracle.project_bwd_f64) — CPU tests.

Why it exists (round-4 verdict "weak #2"): for needle-shaped splats the reference's fp32 chain
vconic -> ∇inverse -> ∇perspective_projection -> ∇covar_world_to_cam -> ∇quat_scale_to_cov -> ∇unnorm_quat2rot
(render.jl:302-385, projection.jl:289-353) loses the thin eigen-direction of the 2x2 covariance: the last bit of vconic moves
∇rotations by 1e-3 relative, for the reference's float atomics, the oracle's double accumulators and the HIP kernels alike.
The replay is (1) pinned by an independent float64 autograd model, (2) equal to the fp32 restatement where that is well
conditioned, (3) insensitive to the last bits of its inputs where the fp32 restatement is not."""
import numpy as np
import torch

import f64_model as fm

DT = torch.float64


def _scene(orc, n, seed, needle_ratio=None):
    rng = np.random.default_rng(seed)
    W, H = 640, 480
    fx = 0.5 * W / np.tan(np.radians(30.0))
    cam = orc.Camera(W, H, (np.float32(fx), np.float32(fx)))
    z = rng.uniform(2.0, 8.0, n)
    means = np.stack([rng.uniform(-0.8, 0.8, n) * z * W / (2 * fx), rng.uniform(-0.8, 0.8, n) * z * H / (2 * fx), z], 1).astype(np.float32)
    base = (6.0 * z / fx)[:, None] * np.exp(0.3 * rng.standard_normal((n, 3)))
    if needle_ratio is not None:
        base[:, 0] *= needle_ratio ** 0.5
        base[:, 2] /= needle_ratio ** 0.5
    scales = base.astype(np.float32)
    rots = rng.standard_normal((n, 4)).astype(np.float32) * rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    vconics = (1e-3 * rng.standard_normal((n, 3))).astype(np.float32)
    vmeans2d = (1e-3 * rng.standard_normal((n, 2))).astype(np.float32)
    return cam, means, scales, rots, vconics, vmeans2d


def _forward(orc, cam, means, scales, rots):
    n = means.shape[0]
    depths = np.zeros(n, np.float32); radii = np.zeros(n, np.int32)
    m2 = np.zeros((n, 2), np.float32); conics = np.zeros((n, 3), np.float32)
    cs = cam.struct()
    import ctypes as C
    orc.lib().orc_project(C.c_int(n), orc._p(means), orc._p(scales), orc._p(rots), C.byref(cs), orc._p(depths),
                          orc._p(radii, C.c_int32), orc._p(m2), orc._p(conics), None)
    return radii, conics


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _autograd(cam, means, scales, rots, vconics, vmeans2d, vis):
    """Independent float64 model: L = sum_i <G_i, conic_i> + <vmeans2d_i, mean2d_i> with G = [[va, vb], [vb, vc]] (the reference puts
    vconic.y on both off-diagonal entries, projection.jl:170-176), differentiated by torch."""
    W, H = cam.width, cam.height
    q = torch.tensor(rots, dtype=DT, requires_grad=True)
    s = torch.tensor(scales, dtype=DT, requires_grad=True)
    m = torch.tensor(means, dtype=DT, requires_grad=True)
    R = torch.tensor(np.asarray(cam.R), dtype=DT); t = torch.tensor(np.asarray(cam.t), dtype=DT)
    pc = m @ R.T + t
    Sc = R @ fm.quat_scale_to_cov(q, s) @ R.T
    S2, m2 = fm.perspective_projection(pc, Sc, cam.focal, (W, H), cam.principal)
    Cn = torch.linalg.inv(S2 + cam.blur_eps * torch.eye(2, dtype=DT))
    vc = torch.tensor(vconics, dtype=DT); vm = torch.tensor(vmeans2d, dtype=DT)
    mask = torch.tensor(vis)
    L = ((vc[:, 0] * Cn[:, 0, 0] + vc[:, 1] * (Cn[:, 0, 1] + Cn[:, 1, 0]) + vc[:, 2] * Cn[:, 1, 1] + (vm * m2).sum(1)) * mask).sum()
    L.backward()
    return m.grad.numpy(), s.grad.numpy(), q.grad.numpy()


def test_replay_is_pinned_by_float64_autograd_and_equals_fp32_where_that_is_well_conditioned(orc):
    cam, means, scales, rots, vconics, vmeans2d = _scene(orc, 400, 11)
    radii, conics = _forward(orc, cam, means, scales, rots)
    vis = radii > 0
    assert vis.sum() > 300
    vm64, vs64, vr64 = orc.project_bwd_f64(vmeans2d, vconics, None, None, radii, means, scales, rots, cam)
    am, as_, aq = _autograd(cam, means, scales, rots, vconics, vmeans2d, vis)
    # (1e-7: the replay keeps the source's float literals — 0.3f is not 0.3 — and the camera's fp32 blur_eps)
    assert _rel(vs64[vis], as_[vis]) < 1e-7 and _rel(vr64[vis], aq[vis]) < 1e-7 and _rel(vm64[vis], am[vis]) < 1e-7
    assert not vs64[~vis].any() and not vr64[~vis].any()
    vm32, vs32, vr32, _, _ = orc.project_bwd(vmeans2d, vconics, None, None, conics, radii, means, scales, rots, cam)
    assert _rel(vs32[vis], vs64[vis]) < 2e-5 and _rel(vr32[vis], vr64[vis]) < 2e-5 and _rel(vm32[vis], vm64[vis]) < 2e-5


def test_fp32_chain_loses_needles_and_the_replay_does_not(orc):
    """150 : 1 needles: flipping the LAST BIT of every vconic entry moves the fp32 restatement's ∇rotations by orders of magnitude
    more than the float64 replay's — the amplification is the arithmetic's, not the map's."""
    cam, means, scales, rots, vconics, vmeans2d = _scene(orc, 400, 12, needle_ratio=150.0)
    radii, conics = _forward(orc, cam, means, scales, rots)
    vis = radii > 0
    assert vis.sum() > 200
    bumped = np.nextafter(vconics, np.float32(np.inf)).astype(np.float32)   # one ulp
    _, _, r32a, _, _ = orc.project_bwd(vmeans2d, vconics, None, None, conics, radii, means, scales, rots, cam)
    _, _, r32b, _, _ = orc.project_bwd(vmeans2d, bumped, None, None, conics, radii, means, scales, rots, cam)
    _, s64a, r64a = orc.project_bwd_f64(vmeans2d, vconics, None, None, radii, means, scales, rots, cam)
    _, _, r64b = orc.project_bwd_f64(vmeans2d, bumped, None, None, radii, means, scales, rots, cam)
    d32, d64 = _rel(r32b[vis], r32a[vis]), _rel(r64b[vis], r64a[vis])
    e32 = _rel(r32a[vis], r64a[vis])
    print(f"one ulp of vconic: fp32 chain moves ∇rotations by {d32:.1e}, the float64 replay by {d64:.1e}; fp32 chain vs replay {e32:.1e}")
    assert d64 < 1e-6 and e32 > 1e-4 and d32 > 20 * d64
    am, as_, aq = _autograd(cam, means, scales, rots, vconics, vmeans2d, vis)
    assert _rel(r64a[vis], aq[vis]) < 1e-6 and _rel(s64a[vis], as_[vis]) < 1e-6
