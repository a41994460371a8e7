"""Host mirror of src/fused_ssim.jl:373-424 (`_fused_ssim`, `fused_ssim_bwd`,
`fused_ssim` and its rrule) and of the photometric loss head of `Trainer.step!`
(src/training.jl:656,684-694), on top of gsr_ssim_* / gsr_loss_l1_ssim.

Tensors are (B,CH,H,W) contiguous ≙ the reference's (W,H,CH,B)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

C1_DEFAULT = 0.01 ** 2
C2_DEFAULT = 0.03 ** 2


class exact_arithmetic:
    """`with fused_ssim.exact_arithmetic():` — gsr_ssim_precision(1) inside the block: every fp32 operation of the three SSIM
    entry points as written (no FMA contraction, IEEE divisions), bit-identical to the CPU oracle.  Outside it the library's
    default (contracted multiply-adds, two reciprocals for the formula's six divisions) applies.  This is the PROCESS-WIDE
    DEFAULT (gsr_ssim_forward / gsr_ssim_backward, and the loss head of rasterizers created without `ssim_precision=`); on exit
    the mode found on entry is restored (nesting, `exact_arithmetic(False)` inside an exact block and GSR_SSIM_EXACT=1
    processes all come back to where they were).  Not a per-thread switch: a rasterizer that must not follow it is created
    with `GaussianRasterizer(..., ssim_precision="exact" | "fast")`."""

    def __init__(self, on: bool = True):
        self.on = bool(on)
        self._prev = []

    def __enter__(self):
        lib = L.load()
        self._prev.append(int(lib.gsr_get_ssim_precision()))
        L.check(lib.gsr_ssim_precision(1 if self.on else 0))
        return self

    def __exit__(self, *exc):
        L.check(L.load().gsr_ssim_precision(self._prev.pop()))
        return False


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4):
        raise ValueError(f"{name} must be a contiguous float32 (B,CH,H,W) HIP tensor")
    return t


def _fused_ssim(img, ref, C1=C1_DEFAULT, C2=C2_DEFAULT, train=True):
    """-> (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12) — fused_ssim.jl:373-391"""
    _chk(img, "img"); _chk(ref, "ref")
    if img.shape != ref.shape:
        raise ValueError("img / ref shape mismatch")
    B, CH, H, W = img.shape
    m = torch.empty_like(img)
    d = [torch.empty_like(img) if train else None for _ in range(3)]
    p = [None if t is None else C.c_void_p(t.data_ptr()) for t in d]
    with torch.cuda.device(img.device):
        L.check(L.load().gsr_ssim_forward(W, H, CH, B, img.data_ptr(), ref.data_ptr(), C1, C2, 1 if train else 0,
                                          m.data_ptr(), p[0], p[1], p[2], _stream()))
    return m, d[0], d[1], d[2]


def fused_ssim_bwd(img, ref, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    """fused_ssim.jl:393-408"""
    B, CH, H, W = img.shape
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        L.check(L.load().gsr_ssim_backward(W, H, CH, B, img.data_ptr(), ref.data_ptr(),
                                           _chk(dL_dmap.contiguous(), "dL_dmap").data_ptr(), dm_dmu1.data_ptr(),
                                           dm_dsigma1_sq.data_ptr(), dm_dsigma12.data_ptr(), out.data_ptr(),
                                           _stream()))
    return out


class _FusedSSIM(torch.autograd.Function):
    """CRC.rrule(::typeof(_fused_ssim), ...) — fused_ssim.jl:416-424"""

    @staticmethod
    def forward(ctx, img, ref, C1, C2):
        img_c, ref_c = img.detach().contiguous(), ref.detach().contiguous()
        m, d0, d1, d2 = _fused_ssim(img_c, ref_c, C1, C2, train=True)
        ctx.save_for_backward(img_c, ref_c, d0, d1, d2)
        return m

    @staticmethod
    def backward(ctx, delta):
        img, ref, d0, d1, d2 = ctx.saved_tensors
        return fused_ssim_bwd(img, ref, delta, d0, d1, d2), None, None, None


def fused_ssim(img, ref, C1=C1_DEFAULT, C2=C2_DEFAULT):
    """fused_ssim(img; ref) — fused_ssim.jl:410-414: the SSIM map, differentiable w.r.t. img."""
    if torch.is_grad_enabled() and img.requires_grad:
        return _FusedSSIM.apply(img, ref, C1, C2)
    return _fused_ssim(img.contiguous(), ref.contiguous(), C1, C2, train=False)[0]


def l1_ssim_loss(rast, image, target, lambda_dssim: float = 0.2):
    """Loss head of Trainer.step! (training.jl:656,684-694) fused with its pullback:
    image (H,W,C) as returned by `rasterize`, target (3,H,W).  Returns (loss 0-d tensor,
    vpixels (H,W,C)) with loss = (1-λ)·mean|x-y| + λ·(1-mean(SSIM))."""
    if tuple(target.shape) != (3, rast.height, rast.width):
        raise ValueError("target must be (3,H,W)")
    loss = torch.empty((), device=image.device, dtype=torch.float32)
    vpix = torch.empty_like(image)
    with torch.cuda.device(image.device):
        L.check(L.load().gsr_loss_l1_ssim(rast._h, image.data_ptr(), target.contiguous().data_ptr(),
                                          float(lambda_dssim), loss.data_ptr(), vpix.data_ptr(), _stream()))
    return loss, vpix
