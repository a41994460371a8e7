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

    def _group(self, theta, grad) -> L.AdamGroup:
        if theta.numel() != self.mu.numel():
            raise ValueError("parameter length does not match the optimizer state")
        if grad.shape != theta.shape:
            raise ValueError("gradient shape does not match the parameter")
        for t, nm in ((theta, "theta"), (grad, "grad")):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError(f"{nm} must be a contiguous float32 HIP device tensor")
        return L.AdamGroup(theta.data_ptr(), grad.data_ptr(), self.mu.data_ptr(), self.nu.data_ptr(), theta.numel(),
                           self.lr, self.current_step)

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
    for i, (o, t, g) in enumerate(zip(opts, thetas, grads)):
        if (o.beta1, o.beta2, o.eps) != (b1, b2, eps):
            raise ValueError("optimizers updated in one launch must share β1, β2, ϵ")
        if t.numel():
            o.current_step += 1
        groups[i] = o._group(t, g)
    L.check(L.load().gsr_adam_step(groups, len(opts), b1, b2, eps, _stream()))
