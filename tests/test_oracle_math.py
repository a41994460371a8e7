"""Pins the oracle's scalar maths with the reference's own unit tests re-expressed
(reference test/runtests.jl:86-324, 555-611).

The reference differentiates its Float32 primal with 5-point finite differences
in Float64 (atol 1e-3 / rtol 5e-3, 100 draws each).  Here the primal is
restated independently in float64 torch (tests/f64_model.py) and differentiated
exactly by autograd; the oracle's hand-written adjoint must agree within the
reference's tolerances.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import f64_model as fm

DT = torch.float64
RNG = np.random.default_rng(20240607)
ATOL, RTOL = 1e-3, 5e-3


def f32(*shape):
    return RNG.standard_normal(shape).astype(np.float32)


def P(a, ct=C.c_float):
    return a.ctypes.data_as(C.POINTER(ct))


def cm(a):
    """row-major (3,3)/(2,2) numpy -> column-major flat float32 (Julia SMatrix order)"""
    return np.ascontiguousarray(np.asarray(a, np.float32).T).reshape(-1)


def from_cm(v, n):
    return np.asarray(v, np.float32).reshape(n, n).T


def close(a, b, atol=ATOL, rtol=RTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.linalg.norm(a - b) <= max(atol, rtol * max(np.linalg.norm(a), np.linalg.norm(b))), (a, b)


def t64(a, grad=True):
    return torch.tensor(np.asarray(a, np.float64), dtype=DT, requires_grad=grad)


# K3 — runtests.jl:86-93
def test_quat2mat(orc):
    from scipy.spatial.transform import Rotation
    L = orc.lib()
    for _ in range(20):
        r = Rotation.from_euler("xyz", RNG.uniform(0, 1, 3))
        x, y, z, w = r.as_quat()
        q = np.array([w, x, y, z], np.float32)
        out = np.zeros(9, np.float32)
        L.orc_unnorm_quat2rot(P(q), P(out))
        assert np.allclose(from_cm(out, 3), r.as_matrix(), atol=1e-6)
        assert np.allclose(fm.quat2rot(t64(q, False)).numpy(), r.as_matrix(), atol=1e-12)


# K4 — runtests.jl:95-125
def test_grad_unnorm_quat2rot(orc):
    L = orc.lib()
    for _ in range(100):
        q = f32(4) * np.float32(0.3 + 2.0 * RNG.uniform())
        vR = f32(3, 3)
        vq = np.zeros(4, np.float32)
        L.orc_grad_unnorm_quat2rot(P(q), P(cm(vR)), P(vq))
        qt = t64(q)
        (fm.quat2rot(qt) * t64(vR, False)).sum().backward()
        close(vq, qt.grad.numpy())
        # R(c·q) = R(q) ⇒ no radial component
        assert abs(float(vq @ q)) / np.linalg.norm(vq) < 1e-5


# K5 — runtests.jl:127-148
def test_grad_pos_world_to_cam(orc):
    L = orc.lib()
    for _ in range(100):
        R, t, p, v = f32(3, 3), f32(3), f32(3), f32(3)
        vR, vt, vp = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        L.orc_grad_pos_world_to_cam(P(cm(R)), P(p), P(v), P(vR), P(vt), P(vp))
        Rt, tt, pt = t64(R), t64(t), t64(p)
        ((Rt @ pt + tt) * t64(v, False)).sum().backward()
        close(from_cm(vR, 3), Rt.grad.numpy()); close(vt, tt.grad.numpy()); close(vp, pt.grad.numpy())
        out = np.zeros(3, np.float32)
        L.orc_pos_world_to_cam(P(cm(R)), P(t), P(p), P(out))
        assert np.allclose(out, R.astype(np.float64) @ p + t, atol=1e-5)


# K6 — runtests.jl:150-173
def test_grad_covar_world_to_cam(orc):
    L = orc.lib()
    for _ in range(100):
        R, A = f32(3, 3), f32(3, 3)
        S = (A @ A.T).astype(np.float32)
        vSc, vRin = f32(3, 3), f32(3, 3)
        vR, vS = np.zeros(9, np.float32), np.zeros(9, np.float32)
        L.orc_grad_covar_world_to_cam(P(cm(R)), P(cm(S)), P(cm(vSc)), P(cm(vRin)), P(vR), P(vS))
        Rt, St = t64(R), t64(S)
        ((Rt @ St @ Rt.T) * t64(vSc, False)).sum().backward()
        close(from_cm(vR, 3) - vRin, Rt.grad.numpy()); close(from_cm(vS, 3), St.grad.numpy())
        out = np.zeros(9, np.float32)
        L.orc_covar_world_to_cam(P(cm(R)), P(cm(S)), P(out))
        assert np.allclose(from_cm(out, 3), R.astype(np.float64) @ S @ R.T, atol=2e-4, rtol=1e-4)


# K7 — runtests.jl:175-216 (inside and outside the FOV clamp, 1920x1080, f=1000)
@pytest.mark.parametrize("inside", [True, False])
def test_grad_perspective_projection(orc, inside):
    L = orc.lib()
    focal = np.array([1000, 1000], np.float32)
    res = np.array([1920, 1080], np.int32)
    principal = np.array([0.5, 0.5], np.float32)
    tan_fov = 0.5 * res / focal
    lim = (res - principal * res) / focal + 0.3 * tan_fov
    for _ in range(50):
        if inside:
            ratio = (2 * RNG.uniform(size=2) - 1) * 0.5 * lim
        else:
            ratio = np.sign(RNG.standard_normal(2)) * (1.2 + 0.5 * RNG.uniform(size=2)) * lim
        z = 2 + 4 * RNG.uniform()
        mean = np.array([ratio[0] * z, ratio[1] * z, z], np.float32)
        A = 0.1 * f32(3, 3)
        S = (A @ A.T).astype(np.float32)
        vS2, vm2 = f32(2, 2), f32(2)
        vS, vmean = np.zeros(9, np.float32), np.zeros(3, np.float32)
        L.orc_grad_perspective_projection(P(mean), P(cm(S)), P(focal), P(res, C.c_int), P(principal), P(cm(vS2)),
                                          P(vm2), P(vS), P(vmean))
        mt, St = t64(mean), t64(S)
        S2, m2 = fm.perspective_projection(mt, St, focal, res, principal)
        ((S2 * t64(vS2, False)).sum() + (m2 * t64(vm2, False)).sum()).backward()
        close(vmean, mt.grad.numpy()); close(from_cm(vS, 3), St.grad.numpy())
        S2o, m2o = np.zeros(4, np.float32), np.zeros(2, np.float32)
        L.orc_perspective_projection(P(mean), P(cm(S)), P(focal), P(res, C.c_int), P(principal), P(S2o), P(m2o))
        assert np.allclose(from_cm(S2o, 2), S2.detach().numpy(), rtol=1e-3, atol=1e-2)
        assert np.allclose(m2o, m2.detach().numpy(), rtol=1e-5, atol=1e-3)


# K8 — runtests.jl:218-239
def test_grad_quat_scale_to_cov(orc):
    L = orc.lib()
    for _ in range(100):
        q = f32(4) * np.float32(0.3 + 2.0 * RNG.uniform())
        s = np.exp(0.5 * f32(3)).astype(np.float32)
        vS = f32(3, 3)
        vq, vs = np.zeros(4, np.float32), np.zeros(3, np.float32)
        L.orc_grad_quat_scale_to_cov(P(q), P(s), P(cm(vS)), P(np.zeros(9, np.float32)), P(vq), P(vs))
        qt, st = t64(q), t64(s)
        (fm.quat_scale_to_cov(qt, st) * t64(vS, False)).sum().backward()
        close(vq, qt.grad.numpy()); close(vs, st.grad.numpy())


# K9 — runtests.jl:241-266 (symmetric 3-entry parametrisation)
def test_grad_inverse(orc):
    L = orc.lib()
    for _ in range(100):
        A = f32(2, 2)
        X = (A @ A.T + 0.5 * np.eye(2)).astype(np.float32)
        b = f32(3)
        vY = np.array([[b[0], b[1]], [b[1], b[2]]], np.float32)
        Y = np.zeros(4, np.float32)
        L.orc_inverse2(P(cm(X)), P(Y))
        assert np.allclose(from_cm(Y, 2), np.linalg.inv(X.astype(np.float64)), rtol=1e-4, atol=1e-5)
        vX = np.zeros(4, np.float32)
        L.orc_grad_inverse2(P(Y), P(cm(vY)), P(vX))
        vX = from_cm(vX, 2)
        p = t64([X[0, 0], X[1, 0], X[1, 1]])
        Xm = torch.stack([torch.stack([p[0], p[1]]), torch.stack([p[1], p[2]])])
        (torch.linalg.inv(Xm) * t64(vY, False)).sum().backward()
        close([vX[0, 0], vX[0, 1] + vX[1, 0], vX[1, 1]], p.grad.numpy())


# K10 — runtests.jl:268-291
def test_grad_add_blur(orc):
    L = orc.lib()
    eps = 0.3
    for _ in range(100):
        A = f32(2, 2)
        S = (A @ A.T + 0.5 * np.eye(2)).astype(np.float32)
        vcomp = float(f32(1)[0])
        Sb, det, comp = np.zeros(4, np.float32), C.c_float(), C.c_float()
        L.orc_add_blur(P(cm(S)), C.c_float(eps), P(Sb), C.byref(det), C.byref(comp))
        conic = np.zeros(4, np.float32)
        L.orc_inverse2(P(Sb), P(conic))
        vS = np.zeros(4, np.float32)
        L.orc_grad_add_blur(comp, C.c_float(vcomp), P(conic), C.c_float(eps), P(vS))
        vS = from_cm(vS, 2)
        p = t64([S[0, 0], S[1, 0], S[1, 1]])
        d0 = p[0] * p[2] - p[1] * p[1]
        d1 = (p[0] + eps) * (p[2] + eps) - p[1] * p[1]
        (vcomp * torch.sqrt(torch.clamp(d0 / d1, min=0))).backward()
        close([vS[0, 0], vS[0, 1] + vS[1, 0], vS[1, 1]], p.grad.numpy(), atol=1e-4)


# K11 — runtests.jl:293-306
def test_grad_normalize(orc):
    L = orc.lib()
    for _ in range(100):
        d = f32(3) * np.float32(0.3 + 2 * RNG.uniform())
        v = f32(3)
        out = np.zeros(3, np.float32)
        L.orc_grad_normalize(P(d), P(v), P(out))
        dt = t64(d)
        ((dt / dt.norm()) * t64(v, False)).sum().backward()
        close(out, dt.grad.numpy())


# K1 — runtests.jl:308-324
def test_get_rect(orc):
    L = orc.lib()
    grid = np.array([64, 64], np.int32)
    px = np.zeros(2, np.float32)
    rmin, rmax = np.zeros(2, np.int32), np.zeros(2, np.int32)
    L.orc_get_rect(P(px), 1, P(grid, C.c_int), P(rmin, C.c_int), P(rmax, C.c_int))
    assert tuple(rmin) == (0, 0) and tuple(rmax) == (1, 1)
    L.orc_get_rect(P(px), 17, P(grid, C.c_int), P(rmin, C.c_int), P(rmax, C.c_int))
    assert tuple(rmin) == (0, 0) and tuple(rmax) == (2, 2)


# K12 — runtests.jl:555-611
def test_gaussian_normal(orc):
    from scipy.spatial.transform import Rotation
    L = orc.lib()
    for _ in range(100):
        q = f32(4) * np.float32(0.3 + 2 * RNG.uniform())
        s = np.exp(0.5 * f32(3)).astype(np.float32)
        Rw = Rotation.random(random_state=int(RNG.integers(1 << 30))).as_matrix().astype(np.float32)
        mc = np.array([f32(1)[0], f32(1)[0], 1 + 5 * RNG.uniform()], np.float32)
        n, k, sg = np.zeros(3, np.float32), C.c_int(), C.c_float()
        L.orc_gaussian_normal(P(cm(Rw)), P(q), P(s), P(mc), P(n), C.byref(k), C.byref(sg))
        assert abs(np.linalg.norm(n) - 1) < 1e-5
        assert float(n @ mc) <= 0
        assert s[k.value] == s.min()
        assert abs(sg.value) == 1.0
        Rg = fm.quat2rot(t64(q, False)).numpy()
        assert np.allclose(n, sg.value * (Rw @ Rg[:, k.value]), atol=1e-5)


def test_grad_gaussian_normal(orc):
    from scipy.spatial.transform import Rotation
    L = orc.lib()
    done = 0
    for _ in range(100):
        q = f32(4) * np.float32(0.3 + 2 * RNG.uniform())
        s = np.exp(np.array([0, 1, 2]) + 0.1 * f32(3)).astype(np.float32)
        Rw = Rotation.random(random_state=int(RNG.integers(1 << 30))).as_matrix().astype(np.float32)
        mc = np.array([f32(1)[0], f32(1)[0], 2 + 5 * RNG.uniform()], np.float32)
        vn = f32(3)
        n, k, sg = np.zeros(3, np.float32), C.c_int(), C.c_float()
        L.orc_gaussian_normal(P(cm(Rw)), P(q), P(s), P(mc), P(n), C.byref(k), C.byref(sg))
        if abs(float(n @ (mc / np.linalg.norm(mc)))) <= 0.1:
            continue
        vRg = np.zeros((3, 3), np.float32)
        vRg[:, k.value] = sg.value * (Rw.T @ vn)
        vq, vs = np.zeros(4, np.float32), np.zeros(3, np.float32)
        L.orc_grad_quat_scale_to_cov(P(q), P(s), P(np.zeros(9, np.float32)), P(cm(vRg)), P(vq), P(vs))
        assert np.all(vs == 0)
        qt = t64(q)
        ncam = sg.value * (t64(Rw, False) @ fm.quat2rot(qt)[:, k.value])
        (ncam * t64(vn, False)).sum().backward()
        close(vq, qt.grad.numpy())
        done += 1
    assert done > 50


def test_single_compare_blend_test_is_the_references_test():
    """The compositing kernels decide `sigma >= 0 && min(0.99, o·exp(-sigma)) >= 1/255` (render.jl:92-95) with ONE unsigned
    compare, bits(sigma) < X, with X - 1 = the bit pattern of the largest sigma for which that very expression passes, found
    per Gaussian by bisection around ln(255·o) (tile_mask.h `blend_threshold_bits`, evaluated by preprocess).  Restated in
    numpy with numpy's float32 exp in both roles: the two tests agree for every (sigma, opacity) pair — on the boundary, a
    few ulps either side (up to the non-monotonicity of the exp between adjacent floats), negative sigma (always rejected), sigma = +0, opacity around and below 1/255 (never blends), NaN.
    (sigma = -0.0, which the reference accepts and the bit compare rejects, cannot come out of the kernels' sigma: its last
    operation is an fma whose addend ha·dx² is >= +0, and x + (+0) is never -0 in round-to-nearest.)"""
    rng = np.random.default_rng(11)
    n = 200_000
    o = rng.uniform(0.0, 1.0, n).astype(np.float32)
    o[: n // 10] = rng.uniform(0.0, 1.0 / 200.0, n // 10).astype(np.float32)      # around and below 1/255
    o[n // 10: n // 10 + 50] = np.float32(1.0 / 255.0)
    amin = np.float32(1.0 / 255.0)

    def passes(bits, oo):
        with np.errstate(over="ignore", invalid="ignore"):
            s = bits.astype(np.uint32).view(np.float32)
            return np.minimum(np.float32(0.99), oo * np.exp(-s, dtype=np.float32)) >= amin

    # blend_threshold_bits, vectorised
    with np.errstate(divide="ignore", invalid="ignore"):
        tau = np.maximum(np.log(np.float32(255.0) * o, dtype=np.float32), np.float32(0.0))
    tau = np.where(np.isnan(tau), np.float32(0.0), tau)
    w = np.float32(4e-7) * np.maximum(tau, np.float32(1.0))
    lo = np.maximum(tau - w, np.float32(0.0)).astype(np.float32).view(np.uint32).astype(np.int64)
    hi = (tau + w).astype(np.float32).view(np.uint32).astype(np.int64)
    never = ~passes(np.zeros(n, np.int64), o)
    lo = np.where(passes(lo, o), lo, 0)
    assert not passes(hi, o)[~never].any()             # the window holds the boundary (no widening needed with a 1-ulp exp)
    steps = 0
    while ((hi - lo) > 1).any():
        mid = lo + ((hi - lo) >> 1)
        ok = passes(mid, o)
        go = (hi - lo) > 1
        lo = np.where(go & ok, mid, lo)
        hi = np.where(go & ~ok, mid, hi)
        steps += 1
    assert steps <= 31                                 # (31 only for an opacity within 4e-7 of 1/255: the window starts at sigma = +0)
    X = np.where(never, 0, lo + 1).astype(np.uint32)

    # pairs: far from the boundary, exactly on it, every bit pattern within +-6 of it, zero, negative, NaN
    for trial in range(16):
        if trial < 13:
            sig_bits = (lo + (trial - 6)).clip(0, None).astype(np.uint32)
            sigma = sig_bits.view(np.float32)
        elif trial == 13:
            sigma = (rng.uniform(-0.2, 1.3, n) * np.maximum(tau, np.float32(0.5))).astype(np.float32)
        elif trial == 14:
            sigma = np.zeros(n, np.float32)
        else:
            sigma = np.full(n, np.nan, np.float32)
        with np.errstate(invalid="ignore", over="ignore"):
            alpha = np.minimum(np.float32(0.99), o * np.exp(-sigma, dtype=np.float32))
            ref = (sigma >= 0) & (alpha >= amin)
        got = sigma.view(np.uint32) < X
        if trial in (6, 7) or trial >= 13:
            assert np.array_equal(ref, got), (trial, int((ref != got).sum()))   # S, S + 1 ulp, and everything far away
        else:
            # the other near neighbours: equal wherever the exp is monotone between adjacent floats (numpy's SIMD expf is
            # not quite: a handful of 200 000 thresholds have a neighbour out of order)
            assert (ref != got).mean() < 1e-4, (trial, int((ref != got).sum()))
    assert not (X[o < amin * np.float32(0.999)]).any()
