"""Host mirror of the boolean-mask compaction that src/densification.jl applies to every
per-Gaussian array (`x[:, mask]`, `x[:, :, mask]`, `x[mask]`: prune_points! :138-191,
densify_clone! :29-62, densify_split! :64-121, _prune_optimizer! :279-288), on top of
gsr_mask_findall / gsr_gather_rows.  Tensors are the C-order equivalents of the Julia arrays,
so the Gaussian index is the FIRST axis here ((N,3) ≙ (3,N), (N,K,3) ≙ (3,K,N))."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

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
