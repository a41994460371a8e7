"""Multi-view data parallelism (no counterpart in the reference, which
renders one camera per step, src/training.jl:587-591).

One process per GPU.  The Gaussian parameters are replicated; rank r renders view
`r` of the batch with its own rasterizer handle; the per-view gradients live in one
contiguous arena `[vrot 4 | vmeans 3 | vshs 3K | vopacity 1 | vscales 3]·N` floats and are
summed across ranks by ONE all-reduce (RCCL over xGMI when the backend is "nccl").
Batch semantics: g = Σ_views ∇L_view(θ) at identical θ.  Per-view side outputs (radii,
∇means_2d) stay local, as they feed per-view densification statistics.

Factored exchange (default for world > 1): 81 % of the arena is ∇shs, and a view's ∇shs is the
outer product basis(dir_view) x vc_view (spherical_harmonics.jl:32-37).  So a rank ships its
(N,3) colour cotangent instead: the arena becomes [vrot 4 | vmeans 3 | vopacity 1 | vscales 3 |
vcolors 3]·N, the first 11·N floats are all-reduced, `vcolors` is all-gathered (12 B per Gaussian
and view) and every rank rebuilds Σ_views basis x vc with one kernel — 56 + 12·V MB over xGMI
per step at N = 1 M instead of a 236 MB all-reduce (the links are point-to-point, 7 x 153 GB/s:
bytes are what an 8-GPU step pays for).
"""
from __future__ import annotations

import os
from datetime import timedelta

import numpy as np
import torch
import torch.distributed as dist


def arena_sizes(n: int, K: int):
    """Element counts of the arena sections, in order (matches rasterizer.backward_raw)."""
    return [4 * n, 3 * n, 3 * K * n, n, 3 * n]


def arena_numel(n: int, K: int) -> int:
    return sum(arena_sizes(n, K))


def split_arena(arena, n: int, K: int):
    """-> dict of views: vrot (N,4), vmeans (N,3), vshs (N,K,3), vopacities (N,), vscales (N,3)"""
    offs = np.cumsum([0] + arena_sizes(n, K))
    return dict(vrot=arena[offs[0]:offs[1]].view(n, 4), vmeans=arena[offs[1]:offs[2]].view(n, 3),
                vshs=arena[offs[2]:offs[3]].view(n, K, 3), vopacities=arena[offs[3]:offs[4]],
                vscales=arena[offs[4]:offs[5]].view(n, 3))


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (as set by torch.distributed.run).  Returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GSR_DIST_FORCE=1: run the collectives even with ONE rank (a 1-rank RCCL communicator): the only way to push the
    # real "nccl" code path — communicator creation, the two extra groups, async work handles — through a 1-GPU box
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # GSR_DIST_BACKEND=gloo lets several ranks share one GPU (logic tests on a 1-GPU box)
            backend = os.environ.get("GSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            ndev = torch.cuda.device_count()
            if local >= ndev:
                # several ranks on one GPU is a logic-test configuration (gloo only); with RCCL it is a
                # mis-launch and must not be hidden by wrapping the device index
                if backend != "gloo":
                    raise RuntimeError(f"LOCAL_RANK={local} but only {ndev} HIP device(s) are visible "
                                       f"(set GSR_DIST_BACKEND=gloo to share a device in logic tests)")
                local = local % max(ndev, 1)
            torch.cuda.set_device(local)
        kw = {}
        if backend == "nccl" and torch.cuda.is_available():
            # bind the communicator to this rank's device explicitly (RCCL otherwise guesses it from the global rank at the
            # first collective — "can cause a hang if rank to GPU mapping is heterogeneous")
            kw["device_id"] = torch.device("cuda", local)
        # a collective that never completes (a rank died, a link hung) must end as an error after a bounded wait, not hang
        # the job: rendezvous, gloo operations and the RCCL watchdog all take this limit (GSR_DIST_TIMEOUT_S, default 120 s)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=collective_timeout(), **kw)
    return rank, world, local


def collective_timeout() -> timedelta:
    return timedelta(seconds=float(os.environ.get("GSR_DIST_TIMEOUT_S", "120")))


def forced() -> bool:
    return os.environ.get("GSR_DIST_FORCE", "0") == "1"


def active() -> bool:
    """Collectives are issued: more than one rank, or a forced 1-rank communicator."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def views_of_rank(rank: int, world: int, n_views: int):
    """Views rendered by `rank`: round-robin, one view per GPU when n_views == world."""
    return list(range(rank, n_views, world))


def allreduce_arena(arena: torch.Tensor, op=dist.ReduceOp.SUM):
    """The path's single collective: sum the gradient arena over all ranks, in place."""
    if active():
        dist.all_reduce(arena, op=op)
    return arena


# ---- factored exchange: [vrot 4 | vmeans 3 | vopacity 1 | vscales 3 | vcolors 3]·N ----
def factored_arena_numel(n: int) -> int:
    return 14 * n


def split_factored_arena(arena, n: int):
    """-> dict of views: vrot (N,4), vmeans (N,3), vopacities (N,), vscales (N,3), vcolors (N,3)"""
    o = np.cumsum([0, 4 * n, 3 * n, n, 3 * n, 3 * n])
    return dict(vrot=arena[o[0]:o[1]].view(n, 4), vmeans=arena[o[1]:o[2]].view(n, 3), vopacities=arena[o[2]:o[3]],
                vscales=arena[o[3]:o[4]].view(n, 3), vcolors=arena[o[4]:o[5]].view(n, 3))


_overlap_groups = None
_last_overlapped = False


def _can_overlap(arena_is_cuda: bool) -> bool:
    """Two communicators in flight at once: RCCL ("nccl") with device tensors, or gloo with HOST tensors.  gloo moving
    DEVICE tensors through the host (several ranks sharing one GPU in logic tests) deadlocks after a few steps with two
    communicators in flight, so that combination runs the sequential form.  Decided from the backend and the tensor,
    not from torch.cuda.is_available() (a GPU host running the gloo/CPU-tensor tests must still overlap)."""
    if not active():
        return False
    return dist.get_backend() == "nccl" or not arena_is_cuda


def overlap_groups():
    """Two extra process groups (= two RCCL communicators with their own streams) so that the two collectives
    of the factored exchange can be in flight at the same time.  Created once, collectively, by every rank,
    whenever collectives are active (cheap; whether they are USED is decided per exchange by `_can_overlap`)."""
    global _overlap_groups
    if _overlap_groups is None and active():
        _overlap_groups = (dist.new_group(timeout=collective_timeout()), dist.new_group(timeout=collective_timeout()))
    return _overlap_groups


def last_exchange_overlapped() -> bool:
    """Whether the most recent exchange_factored_overlapped really had its two collectives in flight together
    (False: it fell back to the sequential form) — so that callers report the form that actually ran."""
    return _last_overlapped


def exchange_factored_overlapped(arena: torch.Tensor, n: int, gathered: torch.Tensor, rebuild):
    """The factored exchange with its two collectives overlapped (SURVEY.md §8e "overlap"): the all-gather of the
    colour cotangents and the all-reduce of the 11·N small gradients are issued together on two communicators;
    `rebuild(vcolors_all)` — the ∇shs reconstruction kernel, which needs only the gathered cotangents — runs on the
    compute stream as soon as the all-gather lands, WHILE the all-reduce is still crossing the links; the compute
    stream joins the all-reduce last.  Same results as exchange_factored + rebuild (the collectives are
    independent); world == 1: no collective.  Falls back to the sequential form when the backend cannot keep two
    communicators in flight for this tensor (`_can_overlap`) or the groups were never created."""
    global _last_overlapped
    world = dist.get_world_size() if dist.is_initialized() else 1
    vc = arena[11 * n:]
    _last_overlapped = False
    if not active():
        return rebuild(vc.view(1, n, 3))
    groups = overlap_groups() if _can_overlap(arena.is_cuda) else None
    if groups is None:
        return rebuild(exchange_factored(arena, n, gathered))
    ga, gb = groups
    h_gather = dist.all_gather_into_tensor(gathered.view(-1), vc, group=ga, async_op=True)
    h_reduce = dist.all_reduce(arena[:11 * n], op=dist.ReduceOp.SUM, group=gb, async_op=True)
    h_gather.wait()  # nccl: the compute stream waits for the gather (no host block); gloo: host wait
    out = rebuild(gathered.view(world, n, 3))
    h_reduce.wait()
    _last_overlapped = True
    return out


def exchange_factored(arena: torch.Tensor, n: int, gathered: torch.Tensor | None = None):
    """The multi-view exchange on a factored arena, in place: all-reduce (sum) of the first 11·N
    floats + all-gather of the (N,3) colour cotangents.  Returns `vcolors_all` (V,N,3), rank-major
    (= view-major when rank r renders view r).  world == 1: no collective, V = 1."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    vc = arena[11 * n:]
    if not active():
        return vc.view(1, n, 3)
    if gathered is None:
        gathered = torch.empty(world * 3 * n, device=arena.device, dtype=arena.dtype)
    dist.all_reduce(arena[:11 * n], op=dist.ReduceOp.SUM)
    try:
        dist.all_gather_into_tensor(gathered.view(-1), vc.contiguous())
    except (RuntimeError, NotImplementedError):  # backends without the flat form
        dist.all_gather(list(gathered.view(world, 3 * n).unbind(0)), vc.contiguous())
    return gathered.view(world, n, 3)
