"""Checkpoints: host mirror of src/checkpoint.jl + the `write_state!` / `read_state!` pairs of
`GaussianModel` (src/gaussians.jl:91-116) and `NU.Adam` (src/training.jl:396-413) — SURVEY.md
§8f rank 4.  A checkpoint is ONE safetensors file: a flat `name -> tensor` table under dotted
prefixes (`gaussians.points`, `optimizers.scales.mu.1`, ...) and every scalar as a string in
the `__metadata__` map, tagged `format = GaussianSplatting.jl-checkpoint-1`.  Pure host I/O.

Tensors are written in C order with the C-order shapes used throughout this package
((N,3) ≙ Julia (3,N)), which is what SafeTensors.jl 1.2.1 (external, absent from the reference
tree) stores for the column-major originals according to checkpoint.jl:30-33 — the exact
dimension order of that package is PARITY UNPINNED here.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

CHECKPOINT_FORMAT = "GaussianSplatting.jl-checkpoint-1"
OPTIMIZER_NAMES = ("points", "features_dc", "features_rest", "opacities", "scales", "rotations")  # training.jl:415-416


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
        return np.ascontiguousarray(self._f.get_tensor(key))

    def read_scalar(self, key: str, typ=int):
        return typ(self.meta[key])


def save_checkpoint(filename: str, tensors: Dict[str, np.ndarray], meta: Dict[str, str]) -> None:
    """checkpoint.jl:44-55."""
    from safetensors.numpy import save_file
    meta = dict(meta)
    meta["format"] = CHECKPOINT_FORMAT
    save_file({k: np.ascontiguousarray(v) for k, v in tensors.items()}, filename, metadata=meta)


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
    return GaussianModel(t["points"], t["features_dc"], t["features_rest"], t["scales"], t["rotations"],
                         t["opacities"], ckpt.read_scalar(f"{prefix}.sh_degree"),
                         ckpt.read_scalar(f"{prefix}.max_sh_degree"))


# ---- NU.Adam (training.jl:396-413): one numbered moment pair per parameter array ----
def write_adam(tensors, meta, prefix: str, opt) -> None:
    tensors[f"{prefix}.mu.1"] = _host(opt.mu)
    tensors[f"{prefix}.nu.1"] = _host(opt.nu)
    meta[f"{prefix}.n_moments"] = "1"
    meta[f"{prefix}.current_step"] = str(int(opt.current_step))


def read_adam(opt, ckpt: Checkpoint, prefix: str) -> None:
    """In place on an optim.Adam (device moments)."""
    import torch
    n = ckpt.read_scalar(f"{prefix}.n_moments")
    if n != 1:
        raise ValueError(f"{prefix}: {n} moment pairs, expected one per parameter array")
    for attr, key in (("mu", "mu.1"), ("nu", "nu.1")):
        host = torch.from_numpy(ckpt.tensor(f"{prefix}.{key}").reshape(-1).astype(np.float32))
        cur = getattr(opt, attr)
        setattr(opt, attr, host.to(cur.device) if cur is not None else host)
    opt.current_step = ckpt.read_scalar(f"{prefix}.current_step")


def save_state(filename: str, gaussians, optimizers: Dict[str, object], step: int) -> None:
    """The Gaussian + optimizer part of `save_state` (training.jl:418-445)."""
    tensors, meta = {}, {}
    write_gaussians(tensors, meta, "gaussians", gaussians)
    for name in OPTIMIZER_NAMES:
        write_adam(tensors, meta, f"optimizers.{name}", optimizers[name])
    meta["step"] = str(int(step))
    save_checkpoint(filename, tensors, meta)


def load_state(filename: str, optimizers: Dict[str, object]):
    """Counterpart (training.jl:447-470): returns (GaussianModel, step); optimizers are filled in place."""
    ckpt = load_checkpoint(filename)
    g = read_gaussians(ckpt, "gaussians")
    for name in OPTIMIZER_NAMES:
        read_adam(optimizers[name], ckpt, f"optimizers.{name}")
    return g, ckpt.read_scalar("step")
