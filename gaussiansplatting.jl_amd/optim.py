"""Host mirror of `NU.Adam` / `NU.step!` / `NU.reset!` (NerfUtils 0.2, external to the reference
tree; call sites src/training.jl:234-239,778, src/strategy.jl:102) on top of gsr_adam_step.

One `Adam` owns the moment pair of ONE parameter array, exactly like the reference's six
optimizers (`optimizers.points`, `.features_dc`, ...); `step_all` updates several of them in
one kernel launch.  μ, ν are flat float32 device vectors (`opt.μ[1]`, `opt.ν[1]` in the
reference's checkpoints, training.jl:396-413).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Adam:
    def __init__(self, theta: torch.Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        if not theta.is_cuda or theta.dtype != torch.float32:
            raise ValueError("Adam parameters must be float32 HIP device tensors (no CPU path)")
        self.mu = torch.zeros(theta.numel(), device=theta.device, dtype=torch.float32)
        self.nu = torch.zeros(theta.numel(), device=theta.device, dtype=torch.float32)
        self.current_step = 0
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)

    def reset(self):
        """NU.reset! (strategy.jl:102): zero the moments and the step counter."""
        self.mu.zero_()
        self.nu.zero_()
        self.current_step = 0

    def _group(self, theta, grad, step: int) -> L.AdamGroup:
        if theta.numel() != self.mu.numel():
            raise ValueError("parameter length does not match the optimizer state")
        if grad.shape != theta.shape:
            raise ValueError("gradient shape does not match the parameter")
        for t, nm in ((theta, "theta"), (grad, "grad")):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError(f"{nm} must be a contiguous float32 HIP device tensor")
        return L.AdamGroup(theta.data_ptr(), grad.data_ptr(), self.mu.data_ptr(), self.nu.data_ptr(), theta.numel(),
                           self.lr, step)

    def step(self, theta: torch.Tensor, grad: torch.Tensor):
        """NU.step!(opt, θ, ∇): in-place update of θ, μ, ν."""
        step_all([self], [theta], [grad])


def step_all(opts: Sequence[Adam], thetas: Sequence[torch.Tensor], grads: Sequence[torch.Tensor]):
    """The loop of training.jl:768-779 as one launch.  Empty parameters are skipped (training.jl:770)."""
    if not (len(opts) == len(thetas) == len(grads)):
        raise ValueError("one optimizer and one gradient per parameter")
    if len(opts) > L.ADAM_MAX_GROUPS:
        raise ValueError(f"at most {L.ADAM_MAX_GROUPS} parameter groups per launch")
    if not opts:
        return
    b1, b2, eps = opts[0].beta1, opts[0].beta2, opts[0].eps
    groups = (L.AdamGroup * len(opts))()
    bump = []
    for i, (o, t, g) in enumerate(zip(opts, thetas, grads)):
        if (o.beta1, o.beta2, o.eps) != (b1, b2, eps):
            raise ValueError("optimizers updated in one launch must share β1, β2, ϵ")
        # the counter the kernel sees is the one AFTER the increment; it is committed only once every
        # group validated and the launch succeeded, so a rejected call leaves all counters untouched
        groups[i] = o._group(t, g, o.current_step + (1 if t.numel() else 0))
        if t.numel():
            bump.append(o)
    L.check(L.load().gsr_adam_step(groups, len(opts), b1, b2, eps, _stream()))
    for o in bump:
        o.current_step += 1


GROUPS = ("points", "features_dc", "features_rest", "opacities", "scales", "rotations")  # training.jl:415-416


def trainer_tail_step(opts, raw, grads, shs, opacities_act, scales_act):
    """Prologue pullback + the six `NU.step!` + the prologue of the next forward in one pass
    (gsr_trainer_tail_step).  `opts`, `raw`: dicts keyed by GROUPS (Adam objects / raw parameter
    tensors); `grads`: dict with vmeans, vshs, vopacities, vscales, vrot (as gsr_backward wrote
    them, w.r.t. the activated values); shs / opacities_act / scales_act are updated in place."""
    n = raw["points"].shape[0]
    k_rest = 0 if raw["features_rest"] is None or raw["features_rest"].numel() == 0 else int(raw["features_rest"].shape[1])
    sd = 1 if raw["scales"].shape[-1] == 1 else 3
    o0 = opts["points"]
    vp = C.c_void_p
    th, mu, nu = (vp * 6)(), (vp * 6)(), (vp * 6)()
    lr, st = (C.c_float * 6)(), (C.c_uint32 * 6)()
    bump = []
    for g, name in enumerate(GROUPS):
        o, t = opts[name], raw[name]
        if (o.beta1, o.beta2, o.eps) != (o0.beta1, o0.beta2, o0.eps):
            raise ValueError("the six optimizers must share β1, β2, ϵ")
        empty = t is None or t.numel() == 0
        if not empty:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) or t.numel() != o.mu.numel():
                raise ValueError(f"{name}: parameter / optimizer state mismatch")
            bump.append(o)
        th[g] = None if empty else t.data_ptr(); mu[g] = None if empty else o.mu.data_ptr()
        nu[g] = None if empty else o.nu.data_ptr()
        lr[g], st[g] = o.lr, o.current_step + (0 if empty else 1)
    for k in ("vmeans", "vshs", "vopacities", "vscales", "vrot"):
        if not (grads[k].is_cuda and grads[k].dtype == torch.float32 and grads[k].is_contiguous()):
            raise ValueError(f"{k} must be a contiguous float32 HIP device tensor")
    tg = L.TailGrads(grads["vmeans"].data_ptr(), grads["vshs"].data_ptr(), grads["vopacities"].data_ptr(),
                     grads["vscales"].data_ptr(), grads["vrot"].data_ptr())
    L.check(L.load().gsr_trainer_tail_step(n, k_rest, sd, C.byref(tg), th, mu, nu, lr, st, o0.beta1, o0.beta2, o0.eps,
                                           shs.data_ptr(), opacities_act.data_ptr(), scales_act.data_ptr(), _stream()))
    for o in bump:  # committed only after validation and a successful launch
        o.current_step += 1


def tail_state(opts, raw, shs, opacities_act, scales_act):
    """gsr_tail_state of a trainer (`opts`, `raw`: dicts keyed by GROUPS) + the optimizers whose counter the
    step will advance.  The counters in the struct are the ones AFTER the increment."""
    o0 = opts["points"]
    st = L.TailState()
    bump = []
    for g, name in enumerate(GROUPS):
        o, t = opts[name], raw[name]
        if (o.beta1, o.beta2, o.eps) != (o0.beta1, o0.beta2, o0.eps):
            raise ValueError("the six optimizers must share β1, β2, ϵ")
        empty = t is None or t.numel() == 0
        if not empty:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) or t.numel() != o.mu.numel():
                raise ValueError(f"{name}: parameter / optimizer state mismatch")
            bump.append(o)
        st.theta[g] = None if empty else t.data_ptr()
        st.mu[g] = None if empty else o.mu.data_ptr()
        st.nu[g] = None if empty else o.nu.data_ptr()
        st.lr[g], st.current_step[g] = o.lr, o.current_step + (0 if empty else 1)
    st.beta1, st.beta2, st.eps = o0.beta1, o0.beta2, o0.eps
    st.scale_dims = 1 if raw["scales"].shape[-1] == 1 else 3
    st.shs, st.opacities_act, st.scales_act = shs.data_ptr(), opacities_act.data_ptr(), scales_act.data_ptr()
    return st, bump


def fused_backward_tail_step(rast, vpixels, opts, raw, shs, opacities_act, scales_act, camera, sh_degree, background,
                             forward_generation: int = 0, color_cotangent: bool = False):
    """The single-GPU `step!` after the loss: ∇rasterize + prologue pullback + the six `NU.step!` + the
    prologue of the next forward, without the gradients ever reaching memory (gsr_backward_trainer_tail).
    Same results, bit for bit, as `rast.backward_raw(...)` followed by `trainer_tail_step(...)`.
    `opts`, `raw`: dicts keyed by GROUPS; raw["points"] / raw["rotations"] are the arrays the forward
    was given as means / rotations, `shs` / `opacities_act` / `scales_act` its other inputs."""
    st, bump = tail_state(opts, raw, shs, opacities_act, scales_act)
    rast.backward_trainer_tail(vpixels, st, raw["points"], shs, opacities_act, scales_act, raw["rotations"], camera,
                               sh_degree, background, forward_generation=forward_generation, color_cotangent=color_cotangent)
    for o in bump:  # committed only after validation and a successful launch
        o.current_step += 1


def sh_views_tail_step(opts, raw, small_grads, vcolors_all, camera_centers, sh_degree: int, shs, opacities_act, scales_act):
    """The multi-GPU `step!` after the gradient exchange, self-contained (gsr_sh_grad_from_views_tail): rebuild
    Σ_views basis x vc from the all-gathered colour cotangents `vcolors_all` (V,N,3) and apply prologue pullback + the six
    `NU.step!` + the next prologue in the same pass — the (N,K,3) ∇shs is never written.  `small_grads`: dict with vmeans,
    vopacities, vscales, vrot of the all-reduced block (distributed.split_factored_arena).  Same results, bit for bit, as
    `rasterizer.sh_grad_from_views(...)` followed by `trainer_tail_step(...)`."""
    n = raw["points"].shape[0]
    K = int(shs.shape[1])
    V = int(vcolors_all.shape[0])
    for t, nm, shape in ((vcolors_all, "vcolors_all", (V, n, 3)), (camera_centers, "camera_centers", (V, 3))):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape):
            raise ValueError(f"{nm} must be a contiguous float32 HIP device tensor of shape {shape}")
    for k in ("vmeans", "vopacities", "vscales", "vrot"):
        if not (small_grads[k].is_cuda and small_grads[k].dtype == torch.float32 and small_grads[k].is_contiguous()):
            raise ValueError(f"{k} must be a contiguous float32 HIP device tensor")
    st, bump = tail_state(opts, raw, shs, opacities_act, scales_act)
    tg = L.TailGrads(small_grads["vmeans"].data_ptr(), None, small_grads["vopacities"].data_ptr(),
                     small_grads["vscales"].data_ptr(), small_grads["vrot"].data_ptr())
    L.check(L.load().gsr_sh_grad_from_views_tail(n, K, int(sh_degree), V, camera_centers.data_ptr(), vcolors_all.data_ptr(),
                                                 C.byref(tg), C.byref(st), _stream()))
    for o in bump:  # committed only after validation and a successful launch
        o.current_step += 1


def nonfinite_gradient_report(names, grads, n: int):
    """The GSP_DEBUG guard of `step!` + the per-parameter part of `nonfinite_gradient_report` (training.jl:534-552,772-777):
    for gradient arrays `grads` (each with the Gaussian index first, n rows; empty ones are skipped) returns
    {name: (n_bad_gaussians, first_bad_index)} for the parameters that hold a NaN / Inf — empty dict = all finite.
    One launch (gsr_count_nonfinite) and one 8-byte-per-parameter read-back."""
    if len(grads) > L.ADAM_MAX_GROUPS:
        raise ValueError(f"at most {L.ADAM_MAX_GROUPS} arrays per launch")
    k = len(grads)
    arr, rw = (C.c_void_p * k)(), (C.c_int32 * k)()
    dev = None
    for i, g in enumerate(grads):
        empty = g is None or g.numel() == 0
        if not empty and not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape[0] == n):
            raise ValueError(f"{names[i]}: gradients must be contiguous float32 HIP device tensors with {n} rows")
        arr[i] = None if empty else g.data_ptr()
        rw[i] = 0 if empty else g.numel() // n
        dev = dev if empty else g.device
    if dev is None:
        return {}
    out = torch.empty((2, k), dtype=torch.int32, device=dev)
    L.check(L.load().gsr_count_nonfinite(arr, rw, k, n, out[0].data_ptr(), out[1].data_ptr(), _stream()))
    host = out.cpu().numpy().astype("uint32")
    return {names[i]: (int(host[0, i]), int(host[1, i])) for i in range(k) if host[0, i] > 0}
