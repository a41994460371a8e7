"""Host mirror of the boolean-mask compaction that src/densification.jl applies to every
per-Gaussian array (`x[:, mask]`, `x[:, :, mask]`, `x[mask]`: prune_points! :138-191,
densify_clone! :29-62, densify_split! :64-121, _prune_optimizer! :279-288), on top of
gsr_mask_findall / gsr_gather_rows.  Tensors are the C-order equivalents of the Julia arrays,
so the Gaussian index is the FIRST axis here ((N,3) ≙ (3,N), (N,K,3) ≙ (3,K,N))."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def findall(mask: torch.Tensor) -> torch.Tensor:
    """`findall(mask)`: ascending 0-based indices of the true entries (device int32 vector).
    One host read-back (the count sizes the result), as the reference's logical indexing has."""
    if not mask.is_cuda or mask.dim() != 1:
        raise ValueError("mask must be a 1-D HIP device tensor (no CPU path)")
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    if mask.dtype != torch.uint8 or not mask.is_contiguous():
        raise ValueError("mask must be contiguous bool / uint8")
    n = mask.numel()
    lib = L.load()
    idx = torch.empty(max(n, 1), device=mask.device, dtype=torch.int32)
    cnt = torch.zeros(1, device=mask.device, dtype=torch.int32)
    scratch = torch.empty(lib.gsr_mask_findall_scratch_bytes(n), device=mask.device, dtype=torch.uint8)
    L.check(lib.gsr_mask_findall(mask.data_ptr(), n, idx.data_ptr(), cnt.data_ptr(), scratch.data_ptr(), _stream()))
    return idx[: int(cnt.item())]


def select(arrays: Sequence[torch.Tensor], idx: torch.Tensor):
    """[x[idx] for x in arrays] along the Gaussian (first) axis — `x[:, idxs]` of the reference —
    for float32 / int32 arrays of any trailing shape, up to 8 per launch."""
    if not (idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and idx.dim() == 1):
        raise ValueError("idx must be a contiguous 1-D int32 HIP device tensor")
    count = idx.numel()
    out = []
    lib = L.load()
    for b in range(0, len(arrays), L.ADAM_MAX_GROUPS):
        chunk = arrays[b:b + L.ADAM_MAX_GROUPS]
        groups = (L.GatherGroup * len(chunk))()
        for i, x in enumerate(chunk):
            if not (x.is_cuda and x.is_contiguous() and x.element_size() == 4):
                raise ValueError("arrays must be contiguous 4-byte-element HIP device tensors")
            rw = 1
            for d in x.shape[1:]:
                rw *= int(d)
            y = torch.empty((count,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
            groups[i] = L.GatherGroup(x.data_ptr(), y.data_ptr(), rw)
            out.append(y)
        L.check(lib.gsr_gather_rows(groups, len(chunk), idx.data_ptr(), count, _stream()))
    return out


def prune(arrays: Sequence[torch.Tensor], valid_mask: torch.Tensor):
    """`x[:, valid_mask]` for every array (prune_points!, densification.jl:138-191)."""
    return select(arrays, findall(valid_mask))


# ======================================================================================================
# DefaultStrategy: the reference's adaptive density control on the device (src/strategy.jl:28-136,
# src/densification.jl:1-297, src/gaussians.jl:115-126).  The control flow below is the reference's, function
# for function; every per-Gaussian pass is one library launch (gsr_densify_* / gsr_compose_rows /
# gsr_split_transform / gsr_reset_opacity / gsr_update_stats).  As in the reference, each logical-indexing step
# reads one count back to size its result.
# ======================================================================================================
PARAMS = ("points", "features_dc", "features_rest", "scales", "rotations", "opacities")  # training.jl:415-416


class GaussianModel:
    """The six parameter arrays of `GaussianModel` (gaussians.jl:2-20) as HIP device tensors, Gaussian index first:
    points (N,3), features_dc (N,1,3), features_rest (N,K-1,3) (may be empty), scales (N,3) or (N,1), rotations (N,4),
    opacities (N,1) — all raw (pre-activation).  `ids`: the optional Int32 (N) label array of the reference's model
    (`use_ids`, gaussians.jl:14,25,46): not a parameter, no optimizer — carried through clone / split / prune like one
    (densification.jl:47,90,185-188,253-257)."""

    def __init__(self, points, features_dc, features_rest, scales, rotations, opacities, ids=None):
        self.points, self.features_dc, self.features_rest = points, features_dc, features_rest
        self.scales, self.rotations, self.opacities = scales, rotations, opacities
        self.ids = ids
        for k in PARAMS:
            t = getattr(self, k)
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError(f"{k} must be a contiguous float32 HIP device tensor (no CPU path)")
        if ids is not None and not (ids.is_cuda and ids.dtype == torch.int32 and ids.is_contiguous()
                                    and tuple(ids.shape) == (points.shape[0],)):
            raise ValueError("ids must be a contiguous int32 HIP device tensor of N elements")

    def __len__(self):
        return int(self.points.shape[0])


class DefaultStrategy:
    """DefaultStrategy (strategy.jl:28-66): per-Gaussian statistics + hyper-parameters."""

    def __init__(self, gs: GaussianModel, dense_percent=1e-2, densify_from_iter=500, densify_until_iter=15_000,
                 densification_interval=100, densify_grad_threshold=2e-4, opacity_reset_interval=3_000, min_opacity=0.005,
                 seed: int = 0, spatial_reorder: bool = False):
        n, dev = len(gs), gs.points.device
        # Not in the reference: re-sort the Gaussians along a Morton curve at the end of every densification round (the
        # arrays are re-composed there anyway; `gs.ids` keeps the identities).  Off by default — it changes the ORDER of the
        # model's rows (and of an exported .ply), nothing else.
        self.spatial_reorder = bool(spatial_reorder)
        self.max_radii = torch.zeros(n, dtype=torch.int32, device=dev)
        self.accum_grad_means_2d = torch.zeros(n, dtype=torch.float32, device=dev)
        self.denom = torch.zeros(n, dtype=torch.float32, device=dev)
        self.dense_percent, self.densify_from_iter, self.densify_until_iter = float(dense_percent), int(densify_from_iter), int(densify_until_iter)
        self.densification_interval, self.densify_grad_threshold = int(densification_interval), float(densify_grad_threshold)
        self.opacity_reset_interval, self.min_opacity = int(opacity_reset_interval), float(min_opacity)
        # Split noise is a pure function of (seed, appended row, draw): the seed MUST differ between densification rounds
        # or appended row i would get the same normal triple every round (the reference draws fresh randn each time,
        # densification.jl:121-135).  `split_seed_base` identifies the run — the trainer's RNG seed (`seed`), the same on
        # every rank of a multi-GPU job so that the replicas split identically —, `split_rounds` advances on every split;
        # both travel in checkpoints (state_dict / checkpoint.save_state(strategy=...)), so a resumed run continues the
        # noise sequence instead of replaying it from round 1.
        self.split_seed_base = int(seed) & 0xFFFFFFFF
        self.split_rounds = 0

    def state_dict(self) -> dict:
        """What a checkpoint must hold for `next_split_seed` to continue where it stopped (scalars only: the per-Gaussian
        statistics restart from zero at every densification anyway, densification.jl:203-209)."""
        return {"split_seed_base": int(self.split_seed_base), "split_rounds": int(self.split_rounds)}

    def load_state_dict(self, d: dict):
        self.split_seed_base = int(d["split_seed_base"]) & 0xFFFFFFFF
        self.split_rounds = int(d["split_rounds"])

    def next_split_seed(self) -> int:
        """Seed for the next split when the caller gives none: distinct for every round of this strategy object."""
        self.split_rounds += 1
        return (self.split_seed_base * 0x9E3779B1 + self.split_rounds * 0x85EBCA6B) & 0xFFFFFFFF


def _ptr(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


def _mask(kind, gs, n_grad=0, grad=None, max_radii=None, thr=0.0, gamma=0.0, min_opacity=0.0, max_screen_size=0):
    n = len(gs)
    mask = torch.empty(n, dtype=torch.uint8, device=gs.points.device)
    L.check(L.load().gsr_densify_mask(kind, n, n_grad, _ptr(grad), _ptr(gs.scales), int(gs.scales.shape[1]), _ptr(gs.opacities),
                                      _ptr(max_radii), float(thr), float(gamma), float(min_opacity), int(max_screen_size),
                                      _ptr(mask), _stream()))
    return mask


def _compose(gs: GaussianModel, optimizers, keep_idx, n_keep, sel_idx, n_sel, reps):
    """One launch for the six parameters and their twelve moment vectors (append_gaussians! + _append_optimizer!, or
    prune_points! + _prune_optimizer!, or both at once for a split).  Replaces the arrays in `gs` / `optimizers`."""
    rows = n_keep + n_sel * reps
    groups, outs = [], []
    for k in PARAMS:
        x = getattr(gs, k)
        rw = int(np.prod(x.shape[1:]))
        if rw == 0:  # empty features_rest (densification.jl:40-41,196-201): passed through, tracking N
            setattr(gs, k, torch.empty((rows,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype))
            continue
        opt = optimizers[k]
        y = torch.empty((rows,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        mu = torch.empty(rows * rw, device=x.device, dtype=torch.float32)
        nu = torch.empty(rows * rw, device=x.device, dtype=torch.float32)
        groups += [L.ComposeGroup(x.data_ptr(), y.data_ptr(), rw, 0), L.ComposeGroup(opt.mu.data_ptr(), mu.data_ptr(), rw, 1),
                   L.ComposeGroup(opt.nu.data_ptr(), nu.data_ptr(), rw, 1)]
        outs.append((k, y, mu, nu))
    new_ids = None
    if gs.ids is not None:  # gs.ids[mask] / repeat(gs.ids[mask], 2) / gs.ids[valid_mask] (densification.jl:47,90,185-188)
        new_ids = torch.empty(rows, device=gs.ids.device, dtype=torch.int32)
        groups.append(L.ComposeGroup(gs.ids.data_ptr(), new_ids.data_ptr(), 1, 0))
    arr = (L.ComposeGroup * len(groups))(*groups)
    L.check(L.load().gsr_compose_rows(arr, len(groups), _ptr(keep_idx), n_keep, _ptr(sel_idx), n_sel, reps, _stream()))
    for k, y, mu, nu in outs:
        setattr(gs, k, y)
        optimizers[k].mu, optimizers[k].nu = mu, nu
    if new_ids is not None:
        gs.ids = new_ids


def _reset_stats(strategy: DefaultStrategy, n, dev):
    """densification_postfix! (densification.jl:203-209): the statistics restart from zero for the whole model"""
    strategy.max_radii = torch.zeros(n, dtype=torch.int32, device=dev)
    strategy.accum_grad_means_2d = torch.zeros(n, dtype=torch.float32, device=dev)
    strategy.denom = torch.zeros(n, dtype=torch.float32, device=dev)


def densify_clone(strategy, gs, optimizers, grad, grad_threshold, extent, dense_percent):
    """densify_clone! (densification.jl:29-62): clone Gaussians with a high gradient and a small size."""
    mask = _mask(L.DENSIFY_CLONE, gs, grad.numel(), grad, thr=grad_threshold, gamma=np.float32(extent) * np.float32(dense_percent))
    sel = findall(mask)
    _compose(gs, optimizers, None, len(gs), sel, sel.numel(), 1)
    _reset_stats(strategy, len(gs), gs.points.device)
    return mask


def densify_split(strategy, gs, optimizers, grad, grad_threshold, extent, dense_percent, seed=None):
    """densify_split! (densification.jl:64-119): replace big Gaussians with a high gradient by two smaller ones
    sampled inside them; the originals are pruned in the same composition.  `seed` selects the noise stream of THIS
    round and must differ from round to round; None = the strategy's own advancing counter (never a constant)."""
    if seed is None:
        seed = strategy.next_split_seed()
    n = len(gs)
    mask = _mask(L.DENSIFY_SPLIT, gs, grad.numel(), grad, thr=grad_threshold, gamma=np.float32(extent) * np.float32(dense_percent))
    sel = findall(mask)
    keep = findall(mask ^ 1)
    m2 = 2 * sel.numel()
    _compose(gs, optimizers, keep, keep.numel(), sel, sel.numel(), 2)
    nk = keep.numel()
    if m2 > 0:  # densification.jl:94
        L.check(L.load().gsr_split_transform(m2, int(gs.scales.shape[1]), gs.points[nk:].data_ptr(), gs.rotations[nk:].data_ptr(),
                                             gs.scales[nk:].data_ptr(), int(seed) & 0xFFFFFFFF, _stream()))
    _reset_stats(strategy, len(gs), gs.points.device)
    assert len(gs) == n - sel.numel() + m2
    return mask


def prune_points(strategy, gs, optimizers, valid_mask):
    """prune_points! (densification.jl:138-191)"""
    keep = findall(valid_mask)
    _compose(gs, optimizers, keep, keep.numel(), None, 0, 1)
    strategy.max_radii, strategy.accum_grad_means_2d, strategy.denom = select(
        [strategy.max_radii, strategy.accum_grad_means_2d, strategy.denom], keep)


def densify_and_prune(strategy: DefaultStrategy, gs: GaussianModel, optimizers, extent, pruning_extent, max_screen_size, seed=None):
    """densify_and_prune! (densification.jl:1-27).  Returns the three masks (clone, split, valid) for inspection."""
    n = len(gs)
    grad = torch.empty(n, dtype=torch.float32, device=gs.points.device)
    L.check(L.load().gsr_densify_grad_mean(n, _ptr(strategy.accum_grad_means_2d), _ptr(strategy.denom), _ptr(grad), _stream()))
    m_clone = densify_clone(strategy, gs, optimizers, grad, strategy.densify_grad_threshold, extent, strategy.dense_percent)
    m_split = densify_split(strategy, gs, optimizers, grad, strategy.densify_grad_threshold, extent, strategy.dense_percent, seed)
    valid = _mask(L.DENSIFY_PRUNE, gs, max_radii=strategy.max_radii, gamma=np.float32(0.1) * np.float32(pruning_extent),
                  min_opacity=strategy.min_opacity, max_screen_size=max_screen_size)
    prune_points(strategy, gs, optimizers, valid)
    if strategy.spatial_reorder:
        reorder_spatially(strategy, gs, optimizers)
    return m_clone, m_split, valid


def reorder_spatially(strategy: DefaultStrategy, gs: GaussianModel, optimizers) -> torch.Tensor:
    """Sort the model's rows (parameters, Adam moments, ids, the strategy's statistics) along a 3-D Morton curve of the
    positions; returns the permutation (new row r = old row perm[r]).  Not a reference function: spatially ordered
    Gaussians make the binning's counter traffic and the tile sort's record gathers coherent (config 3: 1.41 ms per
    step instead of 1.43).  Results of a render are unchanged up to the order of exactly equal depths."""
    n = len(gs)
    if n == 0:
        return torch.empty(0, dtype=torch.int32, device=gs.points.device)
    # bounding box over the FINITE coordinates only: one NaN / Inf position — a transient of a diverging step —
    # used to make the box NaN (gsr_morton_codes: GSR_E_INVALID_ARG, i.e. an exception AFTER clone / split / prune had
    # already mutated model and optimizers) or infinite (every code 0).  A degenerate box skips the re-sort instead: it is an
    # optimisation, never a reason to fail a densification round.  Non-finite rows get clamped codes and sort to an end.
    pts = gs.points.reshape(n, 3)
    finite = torch.isfinite(pts)
    big = torch.finfo(torch.float32).max
    lo = torch.where(finite, pts, torch.full_like(pts, big)).amin(0)
    hi = torch.where(finite, pts, torch.full_like(pts, -big)).amax(0)
    box = torch.cat([lo, hi]).tolist()   # (the one host read of this optional pass)
    if not all(np.isfinite(box)) or any(box[3 + k] < box[k] for k in range(3)) or all(box[3 + k] == box[k] for k in range(3)):
        return torch.arange(n, dtype=torch.int32, device=gs.points.device)
    lo_h = (C.c_float * 3)(*box[:3])
    hi_h = (C.c_float * 3)(*box[3:])
    codes = torch.empty(n, dtype=torch.int64, device=gs.points.device)   # 63-bit codes: order as signed = as unsigned
    L.check(L.load().gsr_morton_codes(n, _ptr(gs.points), lo_h, hi_h, _ptr(codes), _stream()))
    perm = torch.argsort(codes, stable=True).to(torch.int32)
    _compose(gs, optimizers, perm, n, None, 0, 1)
    strategy.max_radii, strategy.accum_grad_means_2d, strategy.denom = select(
        [strategy.max_radii, strategy.accum_grad_means_2d, strategy.denom], perm)
    return perm


def reset_opacity(gs: GaussianModel):
    """reset_opacity! (gaussians.jl:115-126)"""
    L.check(L.load().gsr_reset_opacity(gs.opacities.numel(), _ptr(gs.opacities), _stream()))


def post_train_step(strategy: DefaultStrategy, gs: GaussianModel, optimizers, rast, step: int, extent: float, seed=None):
    """post_train_step! (strategy.jl:78-105), called once per train step after the optimizer update: statistics from
    `rast.gstate.radii` / `rast.gstate.∇means_2d`, densification on its schedule, periodic opacity reset.
    `seed` = noise stream of this step's split (None: derived from the strategy's round counter — distinct per round).
    Returns (densified, reset)."""
    if step > strategy.densify_until_iter:
        return False, False
    rast.update_stats(strategy.max_radii, strategy.accum_grad_means_2d, strategy.denom)
    densified = step >= strategy.densify_from_iter and step % strategy.densification_interval == 0
    if densified:
        mss = 20 if step > strategy.opacity_reset_interval else 0
        n_before = max(len(gs), 1)
        densify_and_prune(strategy, gs, optimizers, extent, extent, mss, seed)
        # The model changed size: let the rasterizer re-size its scratch HERE, with headroom for the next rounds — the reference
        # empties its allocation cache at this very place (strategy.jl:92: GPUArrays.unsafe_free!(cache)) — instead of inside the
        # next training step's forward, where a hipFree + hipMalloc pair synchronises the device (round 6: the first step after
        # a densification took 1.6 x a plain one whenever four per-Gaussian buffers had to grow).
        if hasattr(rast, "reserve"):
            grow = len(gs) / n_before
            rast.reserve(int(1.5 * len(gs)), int(1.5 * max(grow, 1.0) * int(rast.stats.n_rendered)))
    reset = step % strategy.opacity_reset_interval == 0
    if reset:
        reset_opacity(gs)
        optimizers["opacities"].reset()  # NU.reset!(optimizers.opacities)
    return densified, reset
