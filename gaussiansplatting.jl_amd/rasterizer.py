"""Host-side mirror of the reference's rasterizer operator
(src/rasterization/rasterizer.jl): `GaussianRasterizer` (lines 5-90), its functor
prologue (200-253), `rasterize` (255-408), `∇rasterize` (416-550) and the
`ChainRulesCore.rrule` (552-573, here a `torch.autograd.Function`).

All compute happens in libgsr_hip.so through the C ABI of include/gsr.h; torch is
used only for device memory, streams and autograd plumbing.  There is no CPU path:
tensors must live on a HIP device and the library must be present.

Tensor shapes are the C-order equivalents of the reference's column-major arrays
(identical memory): means `(N,3)` ≙ `(3,N)`, shs `(N,K,3)` ≙ `(3,K,N)`, rotations
`(N,4)` (w,x,y,z), opacities `(N,1)`, image `(H,W,C)` ≙ `(C,W,H)`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from .camera import Camera


def n_color_features(mode: str) -> int:
    """rasterizer.jl:47-51"""
    if mode not in L.MODES:
        raise ValueError(f"Invalid render mode: `{mode}`.")
    return L.MODES[mode]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str, shape=None):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a HIP device tensor (no CPU path)")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


class GeometryState:
    """The two members of `rast.gstate` (states.jl:2-47) that code outside the rasterizer reads —
    `radii` (Int32, N) and `∇means_2d` (2,N), consumed by the densification strategy right after a
    training step (strategy.jl:85-86).  As in the reference they are arrays the rasterizer OBJECT
    owns (grow-only, rasterizer.jl:275-278); the library writes straight into them through
    gsr_aux.radii / gsr_grads.vmeans2d, so `rast.gstate.radii` is truthful without a copy."""

    def __init__(self, device):
        self.device = device
        self._cap = 0
        self._n = 0
        self._radii = torch.empty(0, dtype=torch.int32, device=device)
        self._grad_means_2d = torch.empty((0, 2), dtype=torch.float32, device=device)

    def __len__(self):  # Base.length(::GeometryState), states.jl
        return self._cap

    def reserve(self, n: int):
        if self._cap < n:
            self._radii = torch.zeros(n, dtype=torch.int32, device=self.device)
            self._grad_means_2d = torch.zeros((n, 2), dtype=torch.float32, device=self.device)
            self._cap = n

    def ensure(self, n: int):
        if self._cap < n:  # rasterizer.jl:275-278: reallocate when the model grew (here: by at least a quarter)
            self.reserve(max(n, self._cap + self._cap // 4))
        self._n = n

    @property
    def radii(self):
        return self._radii[:self._n]

    @property
    def grad_means_2d(self):
        """gstate.∇means_2d"""
        return self._grad_means_2d[:self._n]


class GaussianRasterizer:
    """GaussianRasterizer(kab; width, height, mode=:rgbd, near_plane=0.2, far_plane=1000)
    — rasterizer.jl:60-90.  Owns the grow-only scratch (gstate/bstate/istate) and the
    output image; supports one outstanding forward→backward pair."""

    def __init__(self, width: int, height: int, mode: str = "rgbd", near_plane: float = 0.2,
                 far_plane: float = 1000.0, device="cuda", radius_clip: int = 3, blur_eps: float = 0.3,
                 exact_tile_cull: bool = True, bins_budget_bytes: int = 0, ssim_precision: Optional[str] = None,
                 preprocess_form: Optional[str] = None, form_tuner: Optional[bool] = None, grad_precision: Optional[str] = None):
        self.mode = mode
        self.channels = n_color_features(mode)
        self.width, self.height = int(width), int(height)
        self.grid = ((self.width + 15) // 16, (self.height + 15) // 16)
        self.near_plane, self.far_plane = float(near_plane), float(far_plane)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("GaussianRasterizer needs a HIP device (no CPU path)")
        self._lib = L.load()
        # exact_tile_cull (the library's default, flags = 0): (Gaussian, tile) instances that cannot reach
        # alpha >= 1/255 anywhere in the tile are not emitted; image, gradients, radii and ∇means_2d are
        # unchanged, only the internal lists (n_rendered, ranges, values_sorted, n_contrib positions) shrink.
        # False = GSR_FLAG_REFERENCE_TILE_LISTS: exactly the reference's lists (list-level parity checks).
        self.exact_tile_cull = bool(exact_tile_cull)
        # The two behaviour switches are per handle (ABI 5), like the reference's constructor keywords
        # (rasterizer.jl:60-65): None = follow the process-wide default (gsr_ssim_precision / gsr_preprocess_form).
        #   ssim_precision : None | "fast" | "exact"   — arithmetic of the loss head on this rasterizer (fused_ssim.l1_ssim_loss)
        #   preprocess_form: None | "direct" | "aggregating" — binning form of the forward's first kernel (same outputs)
        #   form_tuner     : None | False | True — on 4K-class grids the handle times both forms once and keeps the faster
        #                    (None: on unless GSR_FORM_TUNER=0; same outputs either way)
        #   grad_precision : None | "accurate" | "fp32_reference" — the backward's arithmetic on needle-shaped splats (gsr.h):
        #                    "accurate" = libm exp + IEEE division per pixel (+12 % of ∇render!), "fp32_reference" = that and the
        #                    reference's own fp32 expression trees for ∇scales / ∇rotations instead of the float64 chain
        #                    (reference-parity runs)
        try:
            sp = {None: L.DEFAULT, "fast": L.SSIM_FAST, "exact": L.SSIM_EXACT}[ssim_precision]
            pf = {None: L.DEFAULT, "direct": L.PREPROCESS_DIRECT, "aggregating": L.PREPROCESS_AGGREGATING}[preprocess_form]
            ft = {None: L.DEFAULT, False: L.TUNER_OFF, True: L.TUNER_ON}[form_tuner]
            gp = {None: L.DEFAULT, "accurate": L.GRAD_ACCURATE, "fp32_reference": L.GRAD_FP32_REFERENCE}[grad_precision]
        except KeyError as e:
            raise ValueError(f"ssim_precision is None / 'fast' / 'exact', preprocess_form None / 'direct' / 'aggregating', "
                             f"form_tuner None / False / True, grad_precision None / 'accurate' / 'fp32_reference': {e}") from None
        self.ssim_precision, self.preprocess_form = ssim_precision, preprocess_form
        self.form_tuner, self.grad_precision = form_tuner, grad_precision
        # a bare `rasterize` outside autograd renders forward-only (no backward state); False restores the reference's
        # always-state-keeping `rasterize` for callers of the manual rasterize / grad_rasterize pair
        self.forward_only_outside_ad = True
        cfg = L.Config(self.width, self.height, self.channels, self.near_plane, self.far_plane, int(radius_clip),
                       float(blur_eps), 0 if exact_tile_cull else L.FLAG_REFERENCE_TILE_LISTS, int(bins_budget_bytes), sp, pf, ft, gp)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self._lib.gsr_create(C.byref(cfg), C.byref(h)))
        self._h = h
        # rast.image — the returned image aliases rasterizer memory (rasterizer.jl:407)
        self.image = torch.zeros(self.height, self.width, self.channels, device=self.device)
        self.stats = L.Stats()
        self.gstate = GeometryState(self.device)
        self._n = 0

    @classmethod
    def for_camera(cls, camera: Camera, **kw):
        """GaussianRasterizer(kab, camera; kwargs...) — rasterizer.jl:37-40"""
        return cls(camera.width, camera.height, **kw)

    def close(self):
        """KA.unsafe_free!(rast) — rasterizer.jl:136-145"""
        if getattr(self, "_h", None):
            self._lib.gsr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release_scene_buffers(self):
        """release_scene_buffers!(rast) — rasterizer.jl:111-123"""
        L.check(self._lib.gsr_release_scene_buffers(self._h))
        self.gstate = GeometryState(self.device)  # rasterizer.jl:116-117
        self._n = 0

    def reserve(self, n_gaussians: int, n_instances: int = 0):
        """gsr_reserve: pre-size the grow-only scratch (and `gstate`) for views of up to that many Gaussians / tile instances, so
        that the forwards that follow allocate nothing (a reallocation inside a forward synchronises the device).  What a trainer
        calls right after a densification round, with headroom — `densification.post_train_step` does."""
        with torch.cuda.device(self.device):
            L.check(self._lib.gsr_reserve(self._h, int(n_gaussians), int(n_instances)))
        self.gstate.reserve(int(n_gaussians))

    def memory_usage(self) -> int:
        """memory_usage(rast) — rasterizer.jl:127-134"""
        return int(self._lib.gsr_memory_usage(self._h)) + self.image.numel() * 4

    def update_stats(self, max_radii, accum_grad_means_2d, denom):
        """update_stats!(strategy, rast.gstate.radii, rast.gstate.∇means_2d, resolution) —
        strategy.jl:107-136: in-place update of the densification statistics (int32 max_radii,
        float32 accum, float32 denom; N elements each) from the last forward/backward pair."""
        for t, dt in ((max_radii, torch.int32), (accum_grad_means_2d, torch.float32), (denom, torch.float32)):
            if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and t.numel() == self._n):
                raise ValueError("statistics must be contiguous HIP tensors of N elements (int32 / float32 / float32)")
        with torch.cuda.device(self.device):
            L.check(self._lib.gsr_update_stats(self._h, max_radii.data_ptr(), accum_grad_means_2d.data_ptr(),
                                               denom.data_ptr(), _stream()))

    # ---- per-stage kernel timing (HIP events on the launch stream) ----
    def profile(self, on: bool = True, stages=None):
        """Per-stage HIP-event timing on/off; `stages`: iterable of stage names to restrict the event pairs to (None = all)."""
        mask = 0xFFFFFFFF
        if stages is not None:
            names = [self._lib.gsr_profile_stage_name(i).decode() for i in range(self._lib.gsr_profile_stage_count())]
            mask = 0
            for st in stages:
                mask |= 1 << names.index(st)
        L.check(self._lib.gsr_profile_stages(self._h, mask))
        L.check(self._lib.gsr_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, reset: bool = True) -> dict:
        """{stage: (total_ms, launches)} since the last reset; waits for the recorded events."""
        ns = self._lib.gsr_profile_stage_count()
        ms, cnt = (C.c_double * ns)(), (C.c_int * ns)()
        L.check(self._lib.gsr_profile_read(self._h, ms, cnt, 1 if reset else 0))
        return {self._lib.gsr_profile_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(ns)}

    def profile_intervals(self, stage: str) -> list:
        """Milliseconds from each recorded launch of `stage` to the next (gsr_profile_read_intervals): the per-step
        times of a run with one launch of the stage per step.  Call before profile_read(reset=True)."""
        ns = self._lib.gsr_profile_stage_count()
        names = [self._lib.gsr_profile_stage_name(i).decode() for i in range(ns)]
        n = C.c_int()
        L.check(self._lib.gsr_profile_read_intervals(self._h, names.index(stage), None, 0, C.byref(n)))
        out = (C.c_double * max(n.value, 1))()
        L.check(self._lib.gsr_profile_read_intervals(self._h, names.index(stage), out, n.value, C.byref(n)))
        return [out[i] for i in range(n.value)]

    # ---- views into gstate / bstate / istate ----
    def _buffer(self, which: int, dtype, shape):
        p, sz = C.c_void_p(), C.c_size_t()
        L.check(self._lib.gsr_buffer(self._h, which, C.byref(p), C.byref(sz)))
        n = int(np.prod(shape))
        itemsize = torch.empty((), dtype=dtype).element_size()
        if sz.value < n * itemsize or not p.value:
            raise RuntimeError("buffer not produced yet")
        # copied out by the library itself on the current stream (torch owns no view of library memory,
        # and a second HIP runtime must not be opened next to torch's)
        out = torch.empty(shape, dtype=dtype, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self._lib.gsr_copy_buffer(self._h, which, out.data_ptr(), n * itemsize, _stream()))
        return out

    @property
    def radii(self):
        """gstate.radii (Int32, N) — read by densification (strategy.jl:85-86)"""
        return self.gstate.radii

    @property
    def grad_means_2d(self):
        """gstate.∇means_2d (2,N) — valid after the backward"""
        return self.gstate.grad_means_2d

    @property
    def n_contrib(self):
        return self._buffer(L.BUF_N_CONTRIB, torch.int32, (self.height, self.width))

    @property
    def accum_alpha(self):
        """istate.accum_α: the final transmittance T"""
        return self._buffer(L.BUF_FINAL_T, torch.float32, (self.height, self.width))

    @property
    def ranges(self):
        return self._buffer(L.BUF_TILE_RANGES, torch.int32, (self.grid[0] * self.grid[1], 2))

    @property
    def values_sorted(self):
        return self._buffer(L.BUF_VALUES_SORTED, torch.int32, (int(self.stats.n_rendered),))

    @property
    def instance_masks(self):
        """Footprint masks of the sorted instances (int32, D): bits 0..15 tile rows, 16..19 quadrants."""
        d = int(self.stats.n_rendered)
        return self._buffer(L.BUF_INSTANCE_AUX, torch.int32, (d, 4))[:, 3]

    def geometry(self):
        """means_2d, conics, depths, rgbs, clamped, tile rects of the last forward."""
        n = self._n
        g = self._buffer(L.BUF_GEOM, torch.float32, (n, 16))
        gi = g.view(torch.int32)
        lo, hi = gi[:, 12], gi[:, 13]
        rect = torch.stack([lo & 0xFFFF, lo >> 16, hi & 0xFFFF, hi >> 16], 1)
        out = dict(means2d=g[:, 0:2], conics=g[:, 2:5], opacities=g[:, 5], rgbs=g[:, 6:9], clamped_bits=gi[:, 9],
                   depths=g[:, 10], rect=rect)
        if self.channels > 5:
            out["normals"] = self._buffer(L.BUF_NORMALS, torch.float32, (n, 4))[:, :3]
        return out

    # ---- functor prologue: rasterizer.jl:200-253 ----
    def __call__(self, means_3d, opacities, scales, rotations, sh_color, sh_remainder, R_w2c=None, t_w2c=None, *,
                 camera: Camera, sh_degree: int, background=(0.0, 0.0, 0.0), covisibilities=None,
                 uncertainties=None):
        shs, opacities_act, scales_act = _Prologue.apply(sh_color, sh_remainder, opacities, scales)
        return rasterize(means_3d, shs, opacities_act, scales_act, rotations, R_w2c, t_w2c,
                         rast=self, camera=camera, sh_degree=sh_degree, background=background,
                         covisibilities=covisibilities, uncertainties=uncertainties)

    # ---- raw entry points used by rasterize / ∇rasterize ----
    def _inputs(self, means_3d, shs, opacities, scales, rotations, sh_degree, background):
        n = means_3d.shape[0]
        _chk(means_3d, "means_3d", (n, 3)); _chk(scales, "scales", (n, 3)); _chk(rotations, "rotations", (n, 4))
        _chk(shs, "shs"); _chk(opacities, "opacities")
        if shs.dim() != 3 or shs.shape[0] != n or shs.shape[2] != 3:
            raise ValueError(f"shs has shape {tuple(shs.shape)}, expected (N,K,3)")
        if opacities.numel() != n:
            raise ValueError("opacities must have N elements")
        inp = L.Inputs(n, int(shs.shape[1]), int(sh_degree), means_3d.data_ptr(), shs.data_ptr(),
                       opacities.data_ptr(), scales.data_ptr(), rotations.data_ptr(),
                       (C.c_float * 3)(*[float(b) for b in background]))
        return inp

    def _camera(self, camera: Camera, R_w2c, t_w2c):
        if (camera.width, camera.height) != (self.width, self.height):
            raise ValueError("camera resolution does not match the rasterizer")
        cs = L.CameraS()
        R = np.asarray(camera.R, np.float32)
        for c in range(3):
            for r in range(3):
                cs.R[c * 3 + r] = float(R[r, c])
        cc = camera.camera_center
        for k in range(3):
            cs.t[k] = float(camera.t[k])
            cs.camera_center[k] = float(cc[k])
        for k in range(2):
            cs.focal[k] = float(camera.focal[k])
            cs.principal[k] = float(camera.principal[k])
        if R_w2c is not None:
            # pose-optimisation variant (examples/pose_opt.jl): device (3,3) column-major == torch R.T.contiguous()
            _chk(R_w2c, "R_w2c", (3, 3)); _chk(t_w2c, "t_w2c", (3,))
            cs.R_dev, cs.t_dev = R_w2c.data_ptr(), t_w2c.data_ptr()
        return cs

    def forward_raw(self, means_3d, shs, opacities, scales, rotations, camera, sh_degree, background, R_w2c=None,
                    t_w2c=None, covisibilities=None, uncertainties=None, image_out=None, forward_only: bool = False):
        """`rasterize` (rasterizer.jl:255-408).  forward_only=True (GSR_FORWARD_ONLY): the render of the reference's non-AD
        branch (rasterizer.jl:214-248: `validate`, GUI, render-views) — same image bit for bit, no backward state kept
        (a third of the forward's HBM traffic); `backward_raw` after it raises GSR_E_STATE."""
        inp = self._inputs(means_3d, shs, opacities, scales, rotations, sh_degree, background)
        cs = self._camera(camera, R_w2c, t_w2c)
        img = self.image if image_out is None else _chk(image_out, "image_out", (self.height, self.width, self.channels))
        self.gstate.ensure(inp.n)  # rasterizer.jl:275-278
        aux = L.Aux(None if covisibilities is None else covisibilities.data_ptr(),
                    None if uncertainties is None else uncertainties.data_ptr(),
                    self.gstate._radii.data_ptr() if inp.n else None, L.FORWARD_ONLY if forward_only else 0, 0)
        with torch.cuda.device(self.device):
            L.check(self._lib.gsr_forward(self._h, C.byref(inp), C.byref(cs), _ptr(img), C.byref(aux), _stream(),
                                          C.byref(self.stats)))
        self._n = inp.n
        return img

    def backward_raw(self, vpixels, means_3d, shs, opacities, scales, rotations, camera, sh_degree, background,
                     R_w2c=None, t_w2c=None, arena: Optional[torch.Tensor] = None, factored_sh: bool = False,
                     forward_generation: int = 0, color_cotangent: bool = False):
        """∇rasterize.  Returns (vmeans, vshs, vopacities, vscales, vrot, vR, vt); when `arena`
        (a flat float32 tensor of (11+3K)·N elements, 59·N at K=16) is given the five gradients
        are views into it, laid out [vrot | vmeans | vshs | vopacities | vscales] for one
        collective (vrot first keeps its 16-byte alignment for any N).

        factored_sh=True (multi-view exchange): the arena is
        [vrot | vmeans | vopacities | vscales | vcolors] (14·N floats instead of (11+3K)·N) with
        (N,3) `vcolors` — the view's colour cotangent after the clamp mask — in place of vshs; the
        first 11·N floats are summed over ranks, `vcolors` is all-gathered and
        `sh_grad_from_views` rebuilds ∇shs.  The second return value is then `vcolors`.

        color_cotangent=True (GSR_GRADS_COLOR_COTANGENT): the caller's promise that channels >= 3 of `vpixels` are exact zeros —
        the cotangent `fused_ssim.l1_ssim_loss` returns, untouched; in :rgbd / :rgbdn mode the backward then runs the :rgb
        arithmetic (:rgbdn 0.885 -> 0.702 ms).  Not checked: a depth / normal loss added to vpixels must not set it."""
        inp = self._inputs(means_3d, shs, opacities, scales, rotations, sh_degree, background)
        cs = self._camera(camera, R_w2c, t_w2c)
        _chk(vpixels, "vpixels", (self.height, self.width, self.channels))
        n, K = inp.n, inp.n_coeffs
        sizes = [4 * n, 3 * n, n, 3 * n, 3 * n] if factored_sh else [4 * n, 3 * n, 3 * K * n, n, 3 * n]
        if arena is None:
            arena = torch.empty(sum(sizes), device=self.device, dtype=torch.float32)
        elif arena.numel() != sum(sizes):
            raise ValueError("arena has the wrong size")
        offs = np.cumsum([0] + sizes)
        vrot = arena[offs[0]:offs[1]].view(n, 4)
        vmeans = arena[offs[1]:offs[2]].view(n, 3)
        if factored_sh:
            vopac = arena[offs[2]:offs[3]].view(*opacities.shape)
            vscales = arena[offs[3]:offs[4]].view(n, 3)
            vshs = arena[offs[4]:offs[5]].view(n, 3)  # vcolors
        else:
            vshs = arena[offs[2]:offs[3]].view(n, K, 3)
            vopac = arena[offs[3]:offs[4]].view(*opacities.shape)
            vscales = arena[offs[4]:offs[5]].view(n, 3)
        if vrot.data_ptr() % 16:
            raise ValueError("arena must be 16-byte aligned")
        vR = vt = None
        if R_w2c is not None:
            vR = torch.empty(3, 3, device=self.device)
            vt = torch.empty(3, device=self.device)
        g = L.Grads(vmeans.data_ptr(), None if factored_sh else vshs.data_ptr(), vopac.data_ptr(), vscales.data_ptr(),
                    vrot.data_ptr(), None if vR is None else vR.data_ptr(), None if vt is None else vt.data_ptr(),
                    vshs.data_ptr() if factored_sh else None,
                    self.gstate._grad_means_2d.data_ptr() if n else None, int(forward_generation),
                    L.GRADS_COLOR_COTANGENT if color_cotangent else 0, 0)
        with torch.cuda.device(self.device):
            L.check(self._lib.gsr_backward(self._h, C.byref(inp), C.byref(cs), _ptr(vpixels), C.byref(g), _stream()))
        return vmeans, vshs, vopac, vscales, vrot, vR, vt


    def backward_trainer_tail(self, vpixels, tail_state: "L.TailState", means_3d, shs, opacities, scales, rotations,
                              camera, sh_degree, background, forward_generation: int = 0, color_cotangent: bool = False):
        """∇rasterize with the trainer tail applied in its epilogue (gsr_backward_trainer_tail): no gradient
        arrays; the raw parameters, Adam moments and activated copies named by `tail_state` are updated in
        place.  `means_3d` / `rotations` must be the raw points / rotations, `shs` / `opacities` / `scales` the
        activated copies of `tail_state` (the library checks the pointers).  gstate.∇means_2d is written as usual."""
        inp = self._inputs(means_3d, shs, opacities, scales, rotations, sh_degree, background)
        cs = self._camera(camera, None, None)
        _chk(vpixels, "vpixels", (self.height, self.width, self.channels))
        tail_state.vmeans2d = self.gstate._grad_means_2d.data_ptr() if inp.n else None
        tail_state.forward_generation = int(forward_generation)
        tail_state.flags = L.GRADS_COLOR_COTANGENT if color_cotangent else 0   # (see backward_raw)
        with torch.cuda.device(self.device):
            L.check(self._lib.gsr_backward_trainer_tail(self._h, C.byref(inp), C.byref(cs), _ptr(vpixels),
                                                        C.byref(tail_state), _stream()))


def sh_grad_from_views(means_3d, vcolors_all, camera_centers, n_coeffs: int, sh_degree: int, out=None):
    """∇shs (N,K,3) of a batch of views from their factored colour cotangents `vcolors_all` (V,N,3)
    and camera centres (V,3) (gsr_sh_grad_from_views); V = 1 reproduces backward_raw's vshs bit for bit."""
    V, n = vcolors_all.shape[0], vcolors_all.shape[1]
    _chk(means_3d, "means_3d", (n, 3)); _chk(vcolors_all, "vcolors_all", (V, n, 3))
    _chk(camera_centers, "camera_centers", (V, 3))
    if out is None:
        out = torch.empty((n, n_coeffs, 3), device=means_3d.device, dtype=torch.float32)
    else:
        _chk(out, "out", (n, n_coeffs, 3))
    L.check(L.load().gsr_sh_grad_from_views(n, int(n_coeffs), int(sh_degree), V, _ptr(camera_centers), _ptr(means_3d),
                                            _ptr(vcolors_all), _ptr(out), _stream()))
    return out


def prologue_forward(sh_color, sh_remainder, opacities, scales):
    """The functor prologue (rasterizer.jl:218-247) on the device: shs = hcat(sh_color, sh_remainder),
    σ(opacities), exp(scales) with an isotropic (N,1) scale tiled x3.  Returns (shs, opacities_act, scales_act)."""
    n = sh_color.shape[0]
    _chk(sh_color, "sh_color", (n, 1, 3)); _chk(opacities, "opacities"); _chk(scales, "scales")
    k_rest = 0 if sh_remainder is None or sh_remainder.numel() == 0 else int(sh_remainder.shape[1])
    if k_rest:
        _chk(sh_remainder, "sh_remainder", (n, k_rest, 3))
    if opacities.numel() != n or scales.numel() not in (n, 3 * n):
        raise ValueError("opacities must be (N,1) and scales (N,3) or (N,1)")
    sd = 1 if scales.numel() == n and n > 0 and scales.shape[-1] == 1 else 3
    shs = torch.empty((n, 1 + k_rest, 3), device=sh_color.device, dtype=torch.float32)
    oa = torch.empty((n, 1), device=sh_color.device, dtype=torch.float32)
    sa = torch.empty((n, 3), device=sh_color.device, dtype=torch.float32)
    L.check(L.load().gsr_prologue_forward(n, k_rest, sd, _ptr(sh_color), _ptr(sh_remainder) if k_rest else None,
                                          _ptr(opacities), _ptr(scales), _ptr(shs), _ptr(oa), _ptr(sa), _stream()))
    return shs, oa, sa


def prologue_backward(opacities_act, scales_act, vshs, vopacities_act, vscales_act, scale_dims=3):
    """Pullback of the prologue to the raw parameters: (v_sh_color, v_sh_remainder, v_opacities, v_scales)."""
    n, K = vshs.shape[0], vshs.shape[1]
    dev = vshs.device
    vdc = torch.empty((n, 1, 3), device=dev, dtype=torch.float32)
    vrest = torch.empty((n, K - 1, 3), device=dev, dtype=torch.float32)
    vo = torch.empty((n, 1), device=dev, dtype=torch.float32)
    vs = torch.empty((n, scale_dims), device=dev, dtype=torch.float32)
    for t, nm in ((opacities_act, "opacities_act"), (scales_act, "scales_act"), (vshs, "vshs"),
                  (vopacities_act, "vopacities_act"), (vscales_act, "vscales_act")):
        _chk(t, nm)
    L.check(L.load().gsr_prologue_backward(n, K - 1, scale_dims, _ptr(opacities_act), _ptr(scales_act), _ptr(vshs),
                                           _ptr(vopacities_act), _ptr(vscales_act), _ptr(vdc),
                                           _ptr(vrest) if K > 1 else None, _ptr(vo), _ptr(vs), _stream()))
    return vdc, vrest, vo, vs


class _Prologue(torch.autograd.Function):
    """rasterizer.jl:218-247 under AD: forward and pullback are the library's kernels."""

    @staticmethod
    def forward(ctx, sh_color, sh_remainder, opacities, scales):
        shs, oa, sa = prologue_forward(sh_color, sh_remainder, opacities, scales)
        ctx.save_for_backward(oa, sa)
        ctx.scale_dims = 1 if scales.shape[-1] == 1 else 3
        ctx.K = shs.shape[1]
        return shs, oa, sa

    @staticmethod
    def backward(ctx, vshs, voa, vsa):
        oa, sa = ctx.saved_tensors
        n = oa.shape[0]
        vshs = torch.zeros((n, ctx.K, 3), device=oa.device) if vshs is None else vshs.contiguous()
        voa = torch.zeros_like(oa) if voa is None else voa.contiguous()
        vsa = torch.zeros_like(sa) if vsa is None else vsa.contiguous()
        vdc, vrest, vo, vs = prologue_backward(oa, sa, vshs, voa, vsa, ctx.scale_dims)
        return vdc, (vrest if ctx.K > 1 else None), vo, vs


class _Rasterize(torch.autograd.Function):
    """ChainRulesCore.rrule(::typeof(rasterize), ...) — rasterizer.jl:552-573."""

    @staticmethod
    def forward(ctx, means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c, rast, camera, sh_degree, background,
                covisibilities, uncertainties):
        args = [a.detach().contiguous() for a in (means_3d, shs, opacities, scales, rotations)]
        Rd = None if R_w2c is None else R_w2c.detach().contiguous()
        td = None if t_w2c is None else t_w2c.detach().contiguous()
        img = rast.forward_raw(*args, camera, sh_degree, background, Rd, td, covisibilities, uncertainties)
        ctx.generation = int(rast.stats.generation)  # an eval render in between is reported, not silently used
        ctx.rast, ctx.camera, ctx.sh_degree, ctx.background = rast, camera, sh_degree, background
        ctx.pose = (Rd, td)
        ctx.save_for_backward(*args)
        return img

    @staticmethod
    def backward(ctx, vpixels):
        means_3d, shs, opacities, scales, rotations = ctx.saved_tensors
        Rd, td = ctx.pose
        vm, vs, vo, vsc, vr, vR, vt = ctx.rast.backward_raw(
            vpixels.contiguous(), means_3d, shs, opacities, scales, rotations, ctx.camera, ctx.sh_degree,
            ctx.background, Rd, td, forward_generation=ctx.generation)
        return vm, vs, vo, vsc, vr, vR, vt, None, None, None, None, None, None


def rasterize(means_3d, shs, opacities, scales, rotations, R_w2c=None, t_w2c=None, *, rast: GaussianRasterizer,
              camera: Camera, sh_degree: int, background=(0.0, 0.0, 0.0), covisibilities=None, uncertainties=None,
              forward_only: Optional[bool] = None):
    """rasterize(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c; rast, camera,
    sh_degree, background, covisibilities, uncertainties) — rasterizer.jl:255-408.
    opacities / scales are the activated values.  Returns `rast.image` (aliased, overwritten by
    the next call), shape (H,W,C).  R_w2c is the device (3,3) array in the reference's
    column-major order, i.e. the transpose of a row-major torch matrix.
    Outside AD — `torch.no_grad()`, or no argument requires a gradient; the reference's `within_gradient` test,
    rasterizer.jl:214-215 — the render keeps no backward state (GSR_FORWARD_ONLY) and `grad_rasterize` / `backward_raw`
    after it raise GSR_E_STATE.  The manual pair the reference also offers — a bare `rasterize` followed by `∇rasterize` on
    plain arrays, no autograd (rasterizer.jl:255,416) — asks for a state-keeping forward explicitly: `forward_only=False`
    here, or `rast.forward_only_outside_ad = False` once (the Julia binding's `enable_hip_native!(rast;
    forward_only_outside_ad=false)`); `forward_only=True` forces the inference render.  None (default): decided as above."""
    diff = (means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c)
    outside_ad = not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in diff))
    if forward_only is None:
        forward_only = outside_ad and getattr(rast, "forward_only_outside_ad", True)
    elif forward_only and not outside_ad:
        raise ValueError("forward_only=True on a differentiated rasterize: the pullback would find no backward state")
    if outside_ad and not forward_only:
        args = [a.detach().contiguous() for a in diff[:5]]
        return rast.forward_raw(*args, camera, sh_degree, tuple(float(b) for b in background),
                                None if R_w2c is None else R_w2c.detach().contiguous(),
                                None if t_w2c is None else t_w2c.detach().contiguous(), covisibilities, uncertainties)
    if forward_only:
        args = [a.detach().contiguous() for a in diff[:5]]
        return rast.forward_raw(*args, camera, sh_degree, tuple(float(b) for b in background),
                                None if R_w2c is None else R_w2c.detach().contiguous(),
                                None if t_w2c is None else t_w2c.detach().contiguous(), covisibilities, uncertainties,
                                forward_only=True)
    return _Rasterize.apply(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c, rast, camera, sh_degree,
                            tuple(float(b) for b in background), covisibilities, uncertainties)


def grad_rasterize(vpixels, means_3d, shs, scales, rotations, opacities, radii=None, R_w2c=None, t_w2c=None, *,
                   rast: GaussianRasterizer, camera: Camera, sh_degree: int, background=(0.0, 0.0, 0.0),
                   color_cotangent: bool = False):
    """∇rasterize(vpixels, means_3d, shs, scales, rotations, opacities, radii, R_w2c, t_w2c; ...)
    — rasterizer.jl:416-550 (argument order as in the reference; `radii` is taken from the
    rasterizer state).  Needs a state-keeping forward on `rast` just before it: `rasterize(..., forward_only=False)` (or
    `rast.forward_only_outside_ad = False`), `forward_raw(...)`, or a differentiated `rasterize`; after a forward-only
    render it raises GsrError(GSR_E_STATE).  `color_cotangent`: see `GaussianRasterizer.backward_raw`."""
    return rast.backward_raw(vpixels, means_3d, shs, opacities, scales, rotations, camera, sh_degree, background,
                             R_w2c, t_w2c, color_cotangent=color_cotangent)
