"""CPU pins of the oracle's restatement of the two streaming passes either side of rasterize():
the functor prologue (rasterizer.jl:200-253) with its pullback, and the Adam step of NU.step!
(NerfUtils 0.2 — external, absent from the reference tree: PARITY UNPINNED against a live run;
pinned here against the published algorithm in float64 and against torch.optim.Adam)."""
import numpy as np
import pytest
import torch


def _raw(n, kr, iso, seed):
    r = np.random.default_rng(seed)
    return (r.normal(size=(n, 1, 3)).astype(np.float32), r.normal(size=(n, kr, 3)).astype(np.float32) if kr else None,
            r.normal(size=(n, 1)).astype(np.float32) * 3, r.normal(size=(n, 1 if iso else 3)).astype(np.float32))


@pytest.mark.parametrize("kr,iso", [(15, False), (0, False), (8, True), (0, True)])
def test_prologue_forward_is_hcat_sigmoid_exp(orc, kr, iso):
    dc, rest, o, s = _raw(257, kr, iso, 3)
    shs, oa, sa = orc.prologue_forward(dc, rest, o, s)
    assert shs.shape == (257, 1 + kr, 3) and oa.shape == (257, 1) and sa.shape == (257, 3)
    assert np.array_equal(shs[:, :1], dc)                       # rasterizer.jl:218-228
    if kr:
        assert np.array_equal(shs[:, 1:], rest)
    np.testing.assert_allclose(oa, 1.0 / (1.0 + np.exp(-o.astype(np.float64))), rtol=3e-7)   # :229-234
    ref = np.exp(s.astype(np.float64))
    np.testing.assert_allclose(sa, np.broadcast_to(ref, (257, 3)), rtol=3e-7)                # :235-247


@pytest.mark.parametrize("kr,iso", [(15, False), (3, True), (0, False)])
def test_prologue_pullback_vs_float64_autograd(orc, kr, iso):
    n = 64
    dc, rest, o, s = _raw(n, kr, iso, 11)
    shs, oa, sa = orc.prologue_forward(dc, rest, o, s)
    r = np.random.default_rng(5)
    vshs, voa, vsa = (r.normal(size=a.shape).astype(np.float32) for a in (shs, oa, sa))
    vdc, vrest, vo, vs = orc.prologue_backward(oa, sa, vshs, voa, vsa, 1 if iso else 3)
    t = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (dc, rest if kr else np.zeros((n, 0, 3)), o, s)]
    shs_t = torch.cat([t[0], t[1]], 1)
    oa_t = torch.sigmoid(t[2])
    sa_t = torch.exp(t[3].expand(-1, 3) if iso else t[3])
    L = (shs_t * torch.tensor(vshs, dtype=torch.float64)).sum() + (oa_t * torch.tensor(voa, dtype=torch.float64)).sum() + \
        (sa_t * torch.tensor(vsa, dtype=torch.float64)).sum()
    L.backward()
    np.testing.assert_allclose(vdc, t[0].grad.numpy(), rtol=1e-6, atol=1e-7)
    if kr:
        np.testing.assert_allclose(vrest, t[1].grad.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(vo, t[2].grad.numpy(), rtol=1e-4, atol=1e-6)  # fp32 cancellation in a(1-a) for large logits
    np.testing.assert_allclose(vs, t[3].grad.numpy(), rtol=2e-6, atol=1e-6)


def _adam_f64(theta, grads, lr, b1, b2, eps):
    """Kingma & Ba 2015, Algorithm 1, in the §2 'efficient' ordering, float64."""
    th = theta.astype(np.float64)
    m = np.zeros_like(th); v = np.zeros_like(th)
    for t, g in enumerate(grads, 1):
        g = g.astype(np.float64)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        th = th - lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)
    return th, m, v


def test_adam_first_step_is_lr_times_sign(orc):
    """With zero moments the first update is lr·sign(g) for any gradient scale (ϵ = 1f-15, training.jl:229)."""
    r = np.random.default_rng(0)
    th = r.normal(size=1000).astype(np.float32); th0 = th.copy()
    g = (r.normal(size=1000) * 10.0 ** r.uniform(-6, 3, 1000)).astype(np.float32)
    mu = np.zeros_like(th); nu = np.zeros_like(th)
    orc.adam_step(th, g, mu, nu, 1, 1.6e-4, eps=1e-15)
    np.testing.assert_allclose(th0 - th, 1.6e-4 * np.sign(g), rtol=2e-3, atol=1e-9)
    b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))  # the hyper-parameters are Float32 in the reference
    np.testing.assert_allclose(mu, (1 - b1) * g, rtol=1e-6)
    np.testing.assert_allclose(nu, (1 - b2) * g.astype(np.float64) ** 2, rtol=1e-6)


@pytest.mark.parametrize("lr,eps", [(1.6e-4, 1e-15), (2.5e-3, 1e-15), (0.05, 1e-8)])
def test_adam_vs_float64_algorithm_and_torch(orc, lr, eps):
    r = np.random.default_rng(7)
    n, steps = 513, 25
    th = r.normal(size=n).astype(np.float32)
    grads = [r.normal(size=n).astype(np.float32) * (1.0 + 0.1 * k) for k in range(steps)]
    ref, m_ref, v_ref = _adam_f64(th, grads, lr, float(np.float32(0.9)), float(np.float32(0.999)), eps)
    got = th.copy(); mu = np.zeros_like(th); nu = np.zeros_like(th)
    for k, g in enumerate(grads, 1):
        orc.adam_step(got, g, mu, nu, k, lr, 0.9, 0.999, eps)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6 * max(1.0, lr * steps * 40))
    np.testing.assert_allclose(mu, m_ref, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(nu, v_ref, rtol=1e-5, atol=1e-9)
    # torch.optim.Adam places ϵ inside the debiased denominator (sqrt(v̂)+ϵ); at these ϵ the two
    # forms agree far below fp32 resolution of the parameters
    p = torch.tensor(th, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([p], lr=lr, betas=(0.9, 0.999), eps=eps)
    for g in grads:
        p.grad = torch.tensor(g, dtype=torch.float64)
        opt.step()
    np.testing.assert_allclose(got, p.detach().numpy(), rtol=0, atol=5e-6 * max(1.0, lr * steps * 40))


def test_adam_debias_factor(orc):
    b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))
    for t in (1, 2, 10, 1000, 30000):
        want = float(np.float32(0.01)) * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        assert abs(orc.adam_lr_t(0.01, 0.9, 0.999, t) - want) <= 1e-5 * want  # 1 - β^t cancels in fp32 at small t
