"""Checkpoints: host mirror of src/checkpoint.jl + the `write_state!` / `read_state!` pairs of
`GaussianModel` (src/gaussians.jl:91-116) and `NU.Adam` (src/training.jl:396-413) — SURVEY.md
§8f rank 4.  A checkpoint is ONE safetensors file: a flat `name -> tensor` table under dotted
prefixes (`gaussians.points`, `optimizers.scales.mu.1`, ...) and every scalar as a string in
the `__metadata__` map, tagged `format = GaussianSplatting.jl-checkpoint-1`.  Pure host I/O.

On-disk layout.  safetensors stores, per tensor, a shape and the data in C (row-major) order OF THAT SHAPE.
The reference hands SafeTensors.jl its column-major Julia arrays and notes that tensors "are stored in C order,
so [reading] also un-permutes back to the column-major array that was written" (checkpoint.jl:30-33): the header
carries the JULIA shape — `points` is [3, N], `features_rest` [3, K-1, N], `rotations` [4, N] — and the bytes are
that logical array in row-major order.  This package keeps the C-order equivalents in memory ((N,3) ≙ Julia (3,N),
identical bytes to the Julia array), so every tensor is written with its axes reversed and reversed back on read;
a numpy/torch reader of a reference-written file therefore sees (3, N) and gets (N, 3) from `read_gaussians`.
Adam moments are written with the parameter's Julia shape (what `NU.Adam` holds before the first densification)
and accepted either shaped or flat (after `_append_optimizer!` / `_prune_optimizer!` they are flat vectors,
densification.jl:255-288).  Shapes are validated on read.  SafeTensors.jl 1.2.1 itself is external to the reference
tree, so this layout is pinned by the reference's comment and the safetensors specification (tests build a
byte-level file by hand), not by a file the reference wrote: PARITY UNPINNED in that sense.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

CHECKPOINT_FORMAT = "GaussianSplatting.jl-checkpoint-1"
OPTIMIZER_NAMES = ("points", "features_dc", "features_rest", "opacities", "scales", "rotations")  # training.jl:415-416
STRATEGY_STATS = ("max_radii", "accum_grad_means_2d", "denom")  # DefaultStrategy's per-Gaussian statistics (strategy.jl:30-34)


class Checkpoint:
    """A checkpoint opened for reading (checkpoint.jl:17-35): `tensor(key)`, `meta[key]`, `key in ckpt`."""

    def __init__(self, filename: str):
        from safetensors import safe_open
        self._f = safe_open(filename, framework="numpy")
        meta = self._f.metadata()
        if meta is None or meta.get("format") != CHECKPOINT_FORMAT:
            raise ValueError(f"`{filename}` is not a GaussianSplatting.jl checkpoint "
                             f"(no `{CHECKPOINT_FORMAT}` in its metadata).")
        self.meta: Dict[str, str] = dict(meta)
        self._keys = set(self._f.keys())

    def __contains__(self, key: str) -> bool:
        return key in self._keys

    def tensor(self, key: str) -> np.ndarray:
        """The array at `key` in this package's in-memory convention: file (Julia logical) axes reversed."""
        return _from_file(self._f.get_tensor(key))

    def raw_tensor(self, key: str) -> np.ndarray:
        """The array exactly as the file describes it (Julia logical shape)."""
        return np.ascontiguousarray(self._f.get_tensor(key))

    def read_scalar(self, key: str, typ=int):
        return typ(self.meta[key])


def _to_file(x: np.ndarray) -> np.ndarray:
    """In-memory (Gaussian index first) -> file layout (Julia logical shape, row-major bytes)."""
    x = np.asarray(x)
    return np.ascontiguousarray(x.transpose(tuple(reversed(range(x.ndim)))))


def _from_file(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a)
    return np.ascontiguousarray(a.transpose(tuple(reversed(range(a.ndim)))))


def save_checkpoint(filename: str, tensors: Dict[str, np.ndarray], meta: Dict[str, str]) -> None:
    """checkpoint.jl:44-55.  `tensors` are in-memory arrays (Gaussian index first); they are written with the
    reference's (Julia) shapes."""
    from safetensors.numpy import save_file
    meta = dict(meta)
    meta["format"] = CHECKPOINT_FORMAT
    save_file({k: _to_file(v) for k, v in tensors.items()}, filename, metadata=meta)


def load_checkpoint(filename: str) -> Checkpoint:
    """checkpoint.jl:57-70."""
    return Checkpoint(filename)


def _host(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x)


# ---- GaussianModel (gaussians.jl:91-116); `m` is a ply.GaussianModel (numpy or torch arrays) ----
def write_gaussians(tensors, meta, prefix: str, m) -> None:
    for name in ("points", "features_dc", "features_rest", "scales", "rotations", "opacities"):
        tensors[f"{prefix}.{name}"] = _host(getattr(m, name))
    meta[f"{prefix}.sh_degree"] = str(int(m.sh_degree))
    meta[f"{prefix}.max_sh_degree"] = str(int(m.max_sh_degree))


def read_gaussians(ckpt: Checkpoint, prefix: str):
    from .ply import GaussianModel
    t = {name: ckpt.tensor(f"{prefix}.{name}") for name in
         ("points", "features_dc", "features_rest", "scales", "rotations", "opacities")}
    n = t["points"].shape[0]
    want = {"points": lambda a: a.ndim == 2 and a.shape[1] == 3, "features_dc": lambda a: a.ndim == 3 and a.shape[1:] == (1, 3),
            "features_rest": lambda a: a.ndim == 3 and a.shape[2] == 3, "scales": lambda a: a.ndim == 2 and a.shape[1] in (1, 3),
            "rotations": lambda a: a.ndim == 2 and a.shape[1] == 4, "opacities": lambda a: a.ndim == 2 and a.shape[1] == 1}
    for name, ok in want.items():
        if not ok(t[name]) or t[name].shape[0] != n:
            raise ValueError(f"{prefix}.{name}: file shape {ckpt.raw_tensor(f'{prefix}.{name}').shape} is not the reference's "
                             f"layout for {n} Gaussians (expected Julia shapes (3,N), (3,1,N), (3,K-1,N), (3|1,N), (4,N), (1,N))")
    return GaussianModel(t["points"], t["features_dc"], t["features_rest"], t["scales"], t["rotations"],
                         t["opacities"], ckpt.read_scalar(f"{prefix}.sh_degree"),
                         ckpt.read_scalar(f"{prefix}.max_sh_degree"))


# ---- NU.Adam (training.jl:396-413): one numbered moment pair per parameter array ----
def write_adam(tensors, meta, prefix: str, opt, shape=None) -> None:
    """`shape`: the parameter's in-memory shape, e.g. (N,3) — the moments are then written with the parameter's Julia
    shape, as `NU.Adam` holds them; None writes the flat vector (what they are after a densification)."""
    mu, nu = _host(opt.mu), _host(opt.nu)
    if shape is not None:
        if int(np.prod(shape)) != mu.size:
            raise ValueError(f"{prefix}: moments of {mu.size} elements do not match a parameter of shape {tuple(shape)}")
        mu, nu = mu.reshape(shape), nu.reshape(shape)
    tensors[f"{prefix}.mu.1"] = mu
    tensors[f"{prefix}.nu.1"] = nu
    meta[f"{prefix}.n_moments"] = "1"
    meta[f"{prefix}.current_step"] = str(int(opt.current_step))


def read_adam(opt, ckpt: Checkpoint, prefix: str, numel=None) -> None:
    """In place on an optim.Adam (device moments).  Shaped (parameter-shaped) and flat moments are both accepted: the
    flat vector of the in-memory (Gaussian-first, C-order) array IS the column-major flattening of the Julia array."""
    import torch
    n = ckpt.read_scalar(f"{prefix}.n_moments")
    if n != 1:
        raise ValueError(f"{prefix}: {n} moment pairs, expected one per parameter array")
    for attr, key in (("mu", "mu.1"), ("nu", "nu.1")):
        host = torch.from_numpy(ckpt.tensor(f"{prefix}.{key}").reshape(-1).astype(np.float32))
        if numel is not None and host.numel() != numel:
            raise ValueError(f"{prefix}.{key}: {host.numel()} elements, the parameter has {numel}")
        cur = getattr(opt, attr)
        setattr(opt, attr, host.to(cur.device) if cur is not None else host)
    opt.current_step = ckpt.read_scalar(f"{prefix}.current_step")


def save_state(filename: str, gaussians, optimizers: Dict[str, object], step: int, strategy=None,
               extra_meta: Optional[Dict[str, str]] = None) -> None:
    """The Gaussian + optimizer part of `save_state` (training.jl:418-445).  `strategy` (a densification.DefaultStrategy):
    its split-noise position is added as two metadata scalars (`strategy.split_seed_base`, `strategy.split_rounds`) —
    keys the reference's reader never asks for, so the file stays a valid reference checkpoint; the reference itself draws
    split noise from the backend's RNG and has nothing to save there (densification.jl:128).
    `extra_meta`: more string scalars of the same kind — e.g. {"gsr.ssim_precision": "fast"}: the arithmetic of the loss head the
    run trained with (the library default is the contracted build, not bit-identical to the exact twin)."""
    tensors, meta = {}, {}
    write_gaussians(tensors, meta, "gaussians", gaussians)
    for name in OPTIMIZER_NAMES:
        write_adam(tensors, meta, f"optimizers.{name}", optimizers[name], shape=tuple(_host(getattr(gaussians, name)).shape))
    meta["step"] = str(int(step))
    if strategy is not None:
        for k, v in strategy.state_dict().items():
            meta[f"strategy.{k}"] = str(int(v))
        # ... and its running statistics (strategy.jl:30-34: max_radii, accum_∇means_2d, denom).  The reference does not
        # save them — a resumed reference run densifies from half an interval's statistics —; with them a resume in the
        # middle of a densification interval continues BIT-IDENTICALLY (tests/test_gpu_train_protocol.py).  Extra tensors the
        # reference's reader never asks for.
        for k in STRATEGY_STATS:
            if getattr(strategy, k, None) is not None:
                tensors[f"strategy.{k}"] = _host(getattr(strategy, k))
    for k, v in (extra_meta or {}).items():
        meta[str(k)] = str(v)
    save_checkpoint(filename, tensors, meta)


def load_state(filename: str, optimizers: Dict[str, object], strategy=None):
    """Counterpart (training.jl:447-470): returns (GaussianModel, step); optimizers are filled in place, and so is
    `strategy`'s split-noise position when the file carries one."""
    ckpt = load_checkpoint(filename)
    g = read_gaussians(ckpt, "gaussians")
    for name in OPTIMIZER_NAMES:
        read_adam(optimizers[name], ckpt, f"optimizers.{name}", numel=int(np.asarray(getattr(g, name)).size))
    if strategy is not None and "strategy.split_rounds" in ckpt.meta:
        strategy.load_state_dict({k: ckpt.read_scalar(f"strategy.{k}") for k in ("split_seed_base", "split_rounds")})
    if strategy is not None and all(f"strategy.{k}" in ckpt for k in STRATEGY_STATS):
        import torch
        n = int(np.asarray(g.points).shape[0])
        for k in STRATEGY_STATS:
            host = ckpt.tensor(f"strategy.{k}").reshape(-1)
            if host.size != n:
                raise ValueError(f"strategy.{k}: {host.size} entries for {n} Gaussians")
            cur = getattr(strategy, k)
            setattr(strategy, k, torch.from_numpy(np.ascontiguousarray(host)).to(cur.device if cur is not None else "cpu"))
    return g, ckpt.read_scalar("step")
