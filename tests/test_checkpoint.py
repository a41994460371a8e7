"""Checkpoint mirror (src/checkpoint.jl, gaussians.jl:91-116, training.jl:396-470): flat dotted
names, scalars as metadata strings, the format tag (reference test/runtests.jl:904-980)."""
import numpy as np
import pytest


class _Opt:  # stand-in with the fields of optim.Adam that a checkpoint carries (no GPU needed here)
    def __init__(self, n, seed):
        import torch
        r = np.random.default_rng(seed)
        self.mu = torch.from_numpy(r.normal(size=n).astype(np.float32))
        self.nu = torch.from_numpy(r.uniform(size=n).astype(np.float32))
        self.current_step = 7 + seed


def _model(pkg, n=11, kr=15):
    r = np.random.default_rng(1)
    f = lambda *s: r.normal(size=s).astype(np.float32)  # noqa: E731
    return pkg.ply.GaussianModel(f(n, 3), f(n, 1, 3), f(n, kr, 3), f(n, 3), f(n, 4), f(n, 1), 2, 3)


def test_state_roundtrip_names_and_metadata(pkg, tmp_path):
    ck = pkg.checkpoint
    g = _model(pkg)
    sizes = dict(points=33, features_dc=33, features_rest=11 * 45, opacities=11, scales=33, rotations=44)
    opts = {k: _Opt(v, i) for i, (k, v) in enumerate(sizes.items())}
    path = str(tmp_path / "state.safetensors")
    ck.save_state(path, g, opts, step=1234)
    c = ck.load_checkpoint(path)
    assert c.meta["format"] == ck.CHECKPOINT_FORMAT and c.meta["step"] == "1234"
    assert c.meta["gaussians.sh_degree"] == "2" and c.meta["gaussians.max_sh_degree"] == "3"
    for name in ("gaussians.points", "gaussians.features_rest", "optimizers.scales.mu.1", "optimizers.points.nu.1"):
        assert name in c
    assert "sky.gaussians.points" not in c            # optional groups are simply absent
    assert c.meta["optimizers.rotations.n_moments"] == "1"
    fresh = {k: _Opt(v, 99) for k, v in sizes.items()}
    g2, step = ck.load_state(path, fresh)
    assert step == 1234 and g2.sh_degree == 2 and g2.max_sh_degree == 3
    for a in ("points", "features_dc", "features_rest", "scales", "rotations", "opacities"):
        assert np.array_equal(getattr(g2, a), getattr(g, a)), a
    for k in sizes:
        assert np.array_equal(fresh[k].mu.numpy(), opts[k].mu.numpy())
        assert np.array_equal(fresh[k].nu.numpy(), opts[k].nu.numpy())
        assert fresh[k].current_step == opts[k].current_step


def test_foreign_safetensors_is_rejected(pkg, tmp_path):
    from safetensors.numpy import save_file
    path = str(tmp_path / "other.safetensors")
    save_file({"x": np.zeros(3, np.float32)}, path, metadata={"format": "pt"})
    with pytest.raises(ValueError, match="not a GaussianSplatting.jl checkpoint"):
        pkg.checkpoint.load_checkpoint(path)
