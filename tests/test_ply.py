"""`export_ply` / `import_ply` (src/gaussians.jl:157-247): the reference's own test
(test/runtests.jl:982-1048) re-expressed, plus the reader's stated freedoms (property order,
storage precision, ascii)."""
import numpy as np
import pytest


def _model(ply, n, kr, seed=0):
    r = np.random.default_rng(seed)
    rest = np.arange(1, 3 * kr * n + 1, dtype=np.float32).reshape(n, kr, 3)  # distinct: a transposed f_rest cannot pass
    return ply.GaussianModel(r.uniform(size=(n, 3)).astype(np.float32), r.uniform(size=(n, 1, 3)).astype(np.float32),
                             rest, r.uniform(size=(n, 3)).astype(np.float32), r.uniform(size=(n, 4)).astype(np.float32),
                             r.uniform(size=(n, 1)).astype(np.float32), 3 if kr else 0, 3 if kr else 0)


def test_export_header_layout_and_roundtrip(pkg, tmp_path):
    ply = pkg.ply
    n = 8
    gs = _model(ply, n, 15)
    path = tmp_path / "splat.ply"
    ply.export_ply(gs, str(path))
    raw = path.read_bytes()
    header = raw[:raw.index(b"end_header")].decode().split("\n")
    assert not any("float32" in l for l in header)          # canonical `float`, as external viewers expect
    assert header[0] == "ply" and f"element vertex {n}" in header
    for name in ("x", "nx", "f_dc_0", "f_rest_0", "f_rest_44", "opacity", "scale_0", "rot_3"):
        assert f"property float {name}" in header
    assert sum(l.startswith("property") for l in header) == 62
    v = ply._read_vertex(str(path))
    # f_rest is channel-major in the file: R's coefficients, then G's, then B's
    # (Julia features_rest[c, k, 1] == here features_rest[0, k-1, c-1])
    assert v["f_rest_0"][0] == gs.features_rest[0, 0, 0]
    assert v["f_rest_14"][0] == gs.features_rest[0, 14, 0]
    assert v["f_rest_15"][0] == gs.features_rest[0, 0, 1]
    assert v["f_rest_30"][0] == gs.features_rest[0, 0, 2]
    assert not v["nx"].any() and not v["nz"].any()
    g = ply.import_ply(str(path))
    for a in ("points", "features_dc", "features_rest", "scales", "rotations", "opacities"):
        assert np.array_equal(getattr(g, a), getattr(gs, a)), a
    assert g.max_sh_degree == 3


def test_degree_zero_has_no_f_rest(pkg, tmp_path):
    ply = pkg.ply
    gs0 = _model(ply, 8, 0)
    path = tmp_path / "splat0.ply"
    ply.export_ply(gs0, str(path))
    g = ply.import_ply(str(path))
    assert g.max_sh_degree == 0 and g.features_rest.shape == (8, 0, 3)
    assert np.array_equal(g.features_dc, gs0.features_dc)


def test_import_any_property_order_precision_and_ascii(pkg, tmp_path):
    """gaussians.jl:205-211: 'both the property order in the header & the storage precision are free'."""
    ply = pkg.ply
    gs = _model(ply, 5, 3, seed=3)          # degree 1
    names = ply.property_names(3)
    cols = {}
    for i, k in enumerate(("x", "y", "z")): cols[k] = gs.points[:, i]
    for k in ("nx", "ny", "nz"): cols[k] = np.zeros(5, np.float32)
    for i in range(3): cols[f"f_dc_{i}"] = gs.features_dc[:, 0, i]
    rest = gs.features_rest.transpose(0, 2, 1).reshape(5, 9)
    for i in range(9): cols[f"f_rest_{i}"] = rest[:, i]
    cols["opacity"] = gs.opacities[:, 0]
    for i in range(3): cols[f"scale_{i}"] = gs.scales[:, i]
    for i in range(4): cols[f"rot_{i}"] = gs.rotations[:, i]
    order = list(reversed(names))           # shuffled order, doubles, big endian, an extra property
    path = tmp_path / "weird.ply"
    with open(path, "wb") as io:
        hdr = ["ply", "format binary_big_endian 1.0", "comment made by a test", "element vertex 5"] + \
              [f"property double {k}" for k in order] + ["property uchar flag", "end_header"]
        io.write(("\n".join(hdr) + "\n").encode())
        dt = np.dtype([(k, ">f8") for k in order] + [("flag", "u1")])
        rec = np.zeros(5, dt)
        for k in order: rec[k] = cols[k]
        io.write(rec.tobytes())
    g = ply.import_ply(str(path))
    for a in ("points", "features_dc", "features_rest", "scales", "rotations", "opacities"):
        assert np.array_equal(getattr(g, a), getattr(gs, a)), a
    assert g.max_sh_degree == 1
    # ascii
    apath = tmp_path / "ascii.ply"
    with open(apath, "w") as io:
        io.write("\n".join(["ply", "format ascii 1.0", "element vertex 5"] + [f"property float {k}" for k in names] +
                           ["end_header"]) + "\n")
        for r in range(5):
            io.write(" ".join(repr(float(cols[k][r])) for k in names) + "\n")
    g = ply.import_ply(str(apath))
    assert np.array_equal(g.features_rest, gs.features_rest) and np.array_equal(g.rotations, gs.rotations)


def test_import_rejects_partial_sh_bands(pkg, tmp_path):
    ply = pkg.ply
    path = tmp_path / "bad.ply"
    names = [k for k in ply.property_names(3) if k != "f_rest_8"]   # 8 f_rest properties: not a multiple of 3
    with open(path, "wb") as io:
        io.write(("\n".join(["ply", "format binary_little_endian 1.0", "element vertex 1"] +
                            [f"property float {k}" for k in names] + ["end_header"]) + "\n").encode())
        io.write(np.zeros(len(names), np.float32).tobytes())
    with pytest.raises(ValueError, match="whole number of SH"):
        ply.import_ply(str(path))
