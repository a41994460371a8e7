"""Multi-view data parallelism (no counterpart in the reference, which
renders one camera per step, src/training.jl:587-591).

One process per GPU.  The Gaussian parameters are replicated; rank r renders view
`r` of the batch with its own rasterizer handle; the per-view gradients live in one
contiguous arena `[vrot 4 | vmeans 3 | vshs 3K | vopacity 1 | vscales 3]·N` floats and are
summed across ranks by ONE all-reduce (RCCL over xGMI when the backend is "nccl").
Batch semantics: g = Σ_views ∇L_view(θ) at identical θ.  Per-view side outputs (radii,
∇means_2d) stay local, as they feed per-view densification statistics.
"""
from __future__ import annotations

import os

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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # GSR_DIST_BACKEND=gloo lets several ranks share one GPU (logic tests on a 1-GPU box)
            backend = os.environ.get("GSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def views_of_rank(rank: int, world: int, n_views: int):
    """Views rendered by `rank`: round-robin, one view per GPU when n_views == world."""
    return list(range(rank, n_views, world))


def allreduce_arena(arena: torch.Tensor, op=dist.ReduceOp.SUM):
    """The path's single collective: sum the gradient arena over all ranks, in place."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(arena, op=op)
    return arena
