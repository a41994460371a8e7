"""3DGS `.ply` scenes: host mirror of `export_ply` / `import_ply`
(src/gaussians.jl:157-247) so that trained scenes can be fed to the
rasterizer and the benchmark.  `export_ply` / `import_ply` are the host (numpy) forms; `export_ply_device` /
`import_ply_device` keep the model in HBM and build / take apart the AoS vertex rows with one library launch
(gsr_ply_pack_rows / gsr_ply_unpack_rows) — only the header and the byte stream touch the host.

Layout (gaussians.jl:140-156): one `vertex` element, every property `float`, in the order
x y z | nx ny nz (zeros) | f_dc_0..2 | f_rest_0..3(K-1)-1 | opacity | scale_0..2 | rot_0..3.
`f_rest` is CHANNEL-major in the file (all coefficients of R, then G, then B); the model keeps
(channel, coefficient, gaussian), i.e. C-order (N, K-1, 3) here.  Values are the RAW parameters
(opacity logits, log-scales, un-normalised quaternions w,x,y,z).

The reader accepts any property order and any scalar type (only the names matter), ascii or
binary of either endianness, as PlyIO does for the reference.
"""
from __future__ import annotations

import sys
from dataclasses import dataclass

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


@dataclass
class GaussianModel:
    """Arrays of `GaussianModel` (gaussians.jl:1-20) in the C-order equivalents of the Julia layouts."""
    points: np.ndarray         # (N,3)
    features_dc: np.ndarray    # (N,1,3)
    features_rest: np.ndarray  # (N,K-1,3)
    scales: np.ndarray         # (N,3) log-scales
    rotations: np.ndarray      # (N,4) w,x,y,z
    opacities: np.ndarray      # (N,1) logits
    sh_degree: int = 0
    max_sh_degree: int = 0

    @property
    def n(self) -> int:
        return self.points.shape[0]


def property_names(n_rest_coeffs: int):
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] +
            [f"f_rest_{i}" for i in range(3 * n_rest_coeffs)] + ["opacity"] +
            [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])


def export_ply(g: GaussianModel, filename: str) -> None:
    """gaussians.jl:157-203."""
    n = g.n
    kr = g.features_rest.shape[1] if g.features_rest.size else 0
    f32 = lambda a: np.asarray(a, np.float32)  # noqa: E731
    # (gaussian, coefficient, channel) -> per gaussian: channel-major flattening
    rest = f32(g.features_rest).reshape(n, kr, 3).transpose(0, 2, 1).reshape(n, 3 * kr)
    props = np.concatenate([f32(g.points).reshape(n, 3), np.zeros((n, 3), np.float32),
                            f32(g.features_dc).reshape(n, 3), rest, f32(g.opacities).reshape(n, 1),
                            f32(g.scales).reshape(n, 3), f32(g.rotations).reshape(n, 4)], axis=1)
    names = property_names(kr)
    assert props.shape[1] == len(names)
    fmt = "binary_little_endian" if sys.byteorder == "little" else "binary_big_endian"
    with open(filename, "wb") as io:
        header = ["ply", f"format {fmt} 1.0", f"element vertex {n}"] + [f"property float {nm}" for nm in names] + \
                 ["end_header"]
        io.write(("\n".join(header) + "\n").encode("ascii"))
        io.write(np.ascontiguousarray(props, np.float32).tobytes())


def _read_vertex(filename: str):
    with open(filename, "rb") as io:
        if io.readline().strip() != b"ply":
            raise ValueError(f"`{filename}` is not a PLY file")
        fmt, elements, cur = None, [], None
        while True:
            line = io.readline()
            if not line:
                raise ValueError(f"`{filename}`: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = {"name": tok[1], "count": int(tok[2]), "props": []}
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    cur["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    if tok[1] not in _PLY_TYPES:
                        raise ValueError(f"unknown PLY type `{tok[1]}`")
                    cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"`{filename}`: unsupported PLY format `{fmt}`")
        for el in elements:
            if any(p[0] == "list" for p in el["props"]):
                if el["name"] == "vertex":
                    raise ValueError("list properties on the vertex element are not supported")
                if el is not elements[-1] and elements.index(el) < [e["name"] for e in elements].index("vertex"):
                    raise ValueError("a list element precedes `vertex`: not supported")
        out = None
        for el in elements:
            if any(p[0] == "list" for p in el["props"]):
                break  # only trailing list elements (faces) are tolerated; nothing after them is needed
            names = [p[0] for p in el["props"]]
            if fmt == "ascii":
                rows = [io.readline().split() for _ in range(el["count"])]
                data = {nm: np.array([r[j] for r in rows], dtype=np.float64).astype(p[1])
                        for j, (nm, p) in enumerate(zip(names, el["props"]))}
            else:
                end = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(nm, end + p[1]) for nm, p in zip(names, el["props"])])
                rec = np.frombuffer(io.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
                data = {nm: rec[nm] for nm in names}
            if el["name"] == "vertex":
                out = data
                break
        if out is None:
            raise ValueError(f"`{filename}` has no vertex element")
        return out


def import_ply(filename: str) -> GaussianModel:
    """gaussians.jl:205-247: property order and storage precision are free, only the names matter."""
    v = _read_vertex(filename)
    n_frest = sum(1 for k in v if k.startswith("f_rest_"))
    if n_frest % 3 != 0:
        raise ValueError(f"`{filename}` has {n_frest} `f_rest_*` properties, which is not a whole number of SH "
                         "coefficients per color channel.")
    col = lambda name: np.asarray(v[name], np.float32)  # noqa: E731
    n = col("x").shape[0]
    xyz = np.stack([col(k) for k in ("x", "y", "z")], 1)
    scales = np.stack([col(f"scale_{i}") for i in range(3)], 1)
    rots = np.stack([col(f"rot_{i}") for i in range(4)], 1)
    opac = col("opacity").reshape(n, 1)
    dc = np.stack([col(f"f_dc_{i}") for i in range(3)], 1).reshape(n, 1, 3)
    kr = n_frest // 3
    if kr:
        rest = np.stack([col(f"f_rest_{i}") for i in range(n_frest)], 1).reshape(n, 3, kr).transpose(0, 2, 1)
        rest = np.ascontiguousarray(rest)
    else:
        rest = np.empty((n, 0, 3), np.float32)
    deg = int(round(np.sqrt(kr + 1))) - 1
    if (deg + 1) ** 2 != kr + 1:
        raise ValueError(f"`{filename}`: {kr + 1} SH coefficients per channel is not a square number")
    return GaussianModel(np.ascontiguousarray(xyz), dc, rest, np.ascontiguousarray(scales),
                         np.ascontiguousarray(rots), opac, deg, deg)


# ---- device forms: the model lives in HBM (densification.GaussianModel-style objects with torch tensors) ----
def _header(n: int, kr: int) -> bytes:
    fmt = "binary_little_endian" if sys.byteorder == "little" else "binary_big_endian"
    lines = ["ply", f"format {fmt} 1.0", f"element vertex {n}"] + [f"property float {nm}" for nm in property_names(kr)] + ["end_header"]
    return ("\n".join(lines) + "\n").encode("ascii")


def export_ply_device(gs, filename: str) -> None:
    """export_ply (gaussians.jl:157-203) of a device-resident model: the N x (17+3kr) row matrix is packed on the device
    (gsr_ply_pack_rows) and streamed to the file; byte-identical to `export_ply` of the same arrays."""
    import ctypes as C

    import torch

    from . import _lib as L
    n = int(gs.points.shape[0])
    kr = int(gs.features_rest.shape[1]) if gs.features_rest.numel() else 0
    rows = torch.empty((n, 17 + 3 * kr), device=gs.points.device, dtype=torch.float32)
    ptr = lambda t: None if t.numel() == 0 else C.c_void_p(t.data_ptr())  # noqa: E731
    for t in (gs.points, gs.features_dc, gs.features_rest, gs.opacities, gs.scales, gs.rotations):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("export_ply_device needs contiguous float32 HIP device tensors (use export_ply for host arrays)")
    if gs.scales.shape[1] != 3:
        raise ValueError("the .ply layout holds three scales per Gaussian")
    L.check(L.load().gsr_ply_pack_rows(n, kr, ptr(gs.points), ptr(gs.features_dc), ptr(gs.features_rest), ptr(gs.opacities),
                                       ptr(gs.scales), ptr(gs.rotations), ptr(rows), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    with open(filename, "wb") as io:
        io.write(_header(n, kr))
        io.write(rows.cpu().numpy().tobytes())


def import_ply_device(filename: str, device="cuda"):
    """import_ply (gaussians.jl:205-247) into HBM.  A file in the canonical export layout (all-float little-endian rows in
    `property_names` order — what the reference and this package write) is uploaded as one block and taken apart on the
    device (gsr_ply_unpack_rows); any other property order / precision / ascii goes through the host parser first.
    Returns (points, features_dc, features_rest, scales, rotations, opacities, sh_degree) as device tensors."""
    import ctypes as C

    import torch

    from . import _lib as L
    with open(filename, "rb") as io:
        head = b""
        while not head.endswith(b"end_header\n"):
            line = io.readline()
            if not line:
                raise ValueError(f"`{filename}`: unterminated PLY header")
            head += line
        text = head.decode("ascii", "replace").split("\n")
        props = [ln.split() for ln in text if ln.startswith("property")]
        names = [p[2] for p in props if len(p) == 3]
        nverts = [int(ln.split()[2]) for ln in text if ln.startswith("element vertex")]
        n_frest = sum(1 for nm in names if nm.startswith("f_rest_"))
        canonical = (len(nverts) == 1 and sum(ln.startswith("element") for ln in text) == 1 and n_frest % 3 == 0 and
                     f"format binary_{sys.byteorder}_endian 1.0" in [ln.strip() for ln in text] and
                     all(len(p) == 3 and p[1] in ("float", "float32") for p in props) and names == property_names(n_frest // 3))
        if canonical:
            n, kr = nverts[0], n_frest // 3
            raw = np.frombuffer(io.read(4 * n * (17 + 3 * kr)), dtype=np.float32)
            if raw.size != n * (17 + 3 * kr):
                raise ValueError(f"`{filename}`: truncated vertex data")
    if not canonical:
        g = import_ply(filename)
        to = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)  # noqa: E731
        return (to(g.points), to(g.features_dc), to(g.features_rest), to(g.scales), to(g.rotations), to(g.opacities), g.sh_degree)
    deg = int(round(np.sqrt(kr + 1))) - 1
    if (deg + 1) ** 2 != kr + 1:
        raise ValueError(f"`{filename}`: {kr + 1} SH coefficients per channel is not a square number")
    dev = torch.device(device)
    rows = torch.from_numpy(raw.copy()).to(dev)
    mk = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
    pts, dc, rest, sc, rot, op = mk(n, 3), mk(n, 1, 3), mk(n, kr, 3), mk(n, 3), mk(n, 4), mk(n, 1)
    ptr = lambda t: None if t.numel() == 0 else C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(dev):
        L.check(L.load().gsr_ply_unpack_rows(n, kr, ptr(rows), ptr(pts), ptr(dc), ptr(rest), ptr(op), ptr(sc), ptr(rot),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return pts, dc, rest, sc, rot, op, deg
