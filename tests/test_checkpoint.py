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


def test_strategy_split_noise_position_travels_in_the_metadata(pkg, tmp_path):
    """`split_rounds` / `split_seed_base` of the DefaultStrategy are saved as two extra metadata scalars (the
    reference's reader never asks for them) and restored, so a resumed run does not replay its split noise from round 1."""
    ck = pkg.checkpoint

    class _Strat:
        def __init__(self): self.split_seed_base, self.split_rounds = 0, 0
        def state_dict(self): return {"split_seed_base": self.split_seed_base, "split_rounds": self.split_rounds}
        def load_state_dict(self, d): self.split_seed_base, self.split_rounds = int(d["split_seed_base"]), int(d["split_rounds"])

    g = _model(pkg)
    sizes = dict(points=33, features_dc=33, features_rest=11 * 45, opacities=11, scales=33, rotations=44)
    opts = {k: _Opt(v, i) for i, (k, v) in enumerate(sizes.items())}
    st = _Strat(); st.split_seed_base, st.split_rounds = 4242, 17
    path = str(tmp_path / "state.safetensors")
    ck.save_state(path, g, opts, step=5, strategy=st)
    c = ck.load_checkpoint(path)
    assert c.meta["strategy.split_rounds"] == "17" and c.meta["strategy.split_seed_base"] == "4242"
    st2 = _Strat()
    ck.load_state(path, {k: _Opt(v, 9) for k, v in sizes.items()}, strategy=st2)
    assert (st2.split_seed_base, st2.split_rounds) == (4242, 17)
    # a file without the keys (a reference-written checkpoint) leaves the strategy alone
    ck.save_state(path, g, opts, step=5)
    st3 = _Strat(); st3.split_rounds = 3
    ck.load_state(path, {k: _Opt(v, 9) for k, v in sizes.items()}, strategy=st3)
    assert st3.split_rounds == 3


def test_foreign_safetensors_is_rejected(pkg, tmp_path):
    from safetensors.numpy import save_file
    path = str(tmp_path / "other.safetensors")
    save_file({"x": np.zeros(3, np.float32)}, path, metadata={"format": "pt"})
    with pytest.raises(ValueError, match="not a GaussianSplatting.jl checkpoint"):
        pkg.checkpoint.load_checkpoint(path)


def test_on_disk_layout_is_the_references(pkg, tmp_path):
    """checkpoint.jl:30-33 + the safetensors spec: a Julia (3,N) array is stored with header shape [3,N] and row-major
    bytes of that logical array.  (a) what we write has those shapes; (b) a file built BY HAND the way the reference
    writes it — no code of this package involved — loads into the (N,3)-style in-memory arrays; (c) moments are accepted
    shaped or flat and mis-shaped files are rejected instead of being silently reinterpreted."""
    import json
    import struct
    from safetensors import safe_open
    ck = pkg.checkpoint
    g = _model(pkg, n=5, kr=3)
    sizes = dict(points=15, features_dc=15, features_rest=45, opacities=5, scales=15, rotations=20)
    opts = {k: _Opt(v, i) for i, (k, v) in enumerate(sizes.items())}
    path = str(tmp_path / "ours.safetensors")
    ck.save_state(path, g, opts, step=3)
    with safe_open(path, framework="numpy") as f:
        assert f.get_tensor("gaussians.points").shape == (3, 5)
        assert f.get_tensor("gaussians.features_rest").shape == (3, 3, 5)
        assert f.get_tensor("gaussians.rotations").shape == (4, 5) and f.get_tensor("gaussians.opacities").shape == (1, 5)
        assert f.get_tensor("optimizers.features_rest.mu.1").shape == (3, 3, 5)
        assert np.array_equal(f.get_tensor("gaussians.points")[1], g.points[:, 1])       # A[d, i] = points[i, d]

    # (b) a reference-style file, byte by byte
    n, kr = 4, 2
    julia = dict(points=(3, n), features_dc=(3, 1, n), features_rest=(3, kr, n), scales=(3, n), rotations=(4, n), opacities=(1, n))
    rng = np.random.default_rng(9)
    logical = {k: rng.normal(size=shp).astype(np.float32) for k, shp in julia.items()}      # A[d, (k,) i] as Julia indexes it
    tensors = {f"gaussians.{k}": v for k, v in logical.items()}
    flat_moment = rng.normal(size=3 * n).astype(np.float32)                                 # after a densification: flat, vec(A)
    for name in ck.OPTIMIZER_NAMES:
        shp = julia[name]
        col_major_vec = lambda a: np.ascontiguousarray(a.transpose(tuple(reversed(range(a.ndim))))).reshape(-1)  # noqa: E731
        if name == "points":
            tensors[f"optimizers.{name}.mu.1"] = flat_moment
            tensors[f"optimizers.{name}.nu.1"] = flat_moment * 2
        else:
            tensors[f"optimizers.{name}.mu.1"] = rng.normal(size=shp).astype(np.float32)
            tensors[f"optimizers.{name}.nu.1"] = rng.uniform(size=shp).astype(np.float32)
    meta = {"format": ck.CHECKPOINT_FORMAT, "step": "42", "gaussians.sh_degree": "1", "gaussians.max_sh_degree": "1"}
    for name in ck.OPTIMIZER_NAMES:
        meta[f"optimizers.{name}.n_moments"] = "1"; meta[f"optimizers.{name}.current_step"] = "17"
    header, blob = {"__metadata__": meta}, b""
    for k, v in tensors.items():
        data = np.ascontiguousarray(v).tobytes()                                              # row-major bytes of the logical shape
        header[k] = {"dtype": "F32", "shape": list(v.shape), "data_offsets": [len(blob), len(blob) + len(data)]}
        blob += data
    hj = json.dumps(header).encode()
    hj += b" " * (-len(hj) % 8)
    ref_path = str(tmp_path / "reference_style.safetensors")
    with open(ref_path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)) + hj + blob)
    fresh = {k: _Opt(1, 0) for k in ck.OPTIMIZER_NAMES}
    g2, step = ck.load_state(ref_path, fresh)
    assert step == 42 and g2.points.shape == (n, 3) and g2.features_rest.shape == (n, kr, 3) and g2.rotations.shape == (n, 4)
    for i in range(n):
        assert np.array_equal(g2.points[i], logical["points"][:, i])
        assert np.array_equal(g2.features_rest[i], logical["features_rest"][:, :, i].T)
    # moments: vec() of the Julia array == flat C-order of the in-memory (N,...) array, shaped or already flat
    assert np.array_equal(fresh["points"].mu.numpy(), flat_moment)
    want = np.ascontiguousarray(tensors["optimizers.rotations.mu.1"].T).reshape(-1)
    assert np.array_equal(fresh["rotations"].mu.numpy(), want) and fresh["rotations"].current_step == 17

    # (c) a file in the old C-order-shape convention ((N,3) in the header) is rejected, not reinterpreted
    from safetensors.numpy import save_file
    bad = {k: np.ascontiguousarray(np.asarray(v).transpose(tuple(reversed(range(np.asarray(v).ndim))))) for k, v in tensors.items()}
    bad_path = str(tmp_path / "bad.safetensors")
    save_file(bad, bad_path, metadata=meta)
    with pytest.raises(ValueError, match="not the reference's layout"):
        ck.load_state(bad_path, {k: _Opt(1, 0) for k in ck.OPTIMIZER_NAMES})
