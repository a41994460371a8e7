"""ctypes/numpy front-end of the CPU oracle (oracle/gsr_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of gsr_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

`forward()` / `backward()` restate the orchestration of
src/rasterization/rasterizer.jl:255-408 (`rasterize`) and :416-550 (`∇rasterize`);
every intermediate the reference keeps in `gstate/bstate/istate`
(states.jl:2-111) is returned so the HIP kernels can be compared stage by stage.
Array layouts are the reference's (Julia column-major `(3,N)` == numpy `(N,3)`
C-order, `(C,W,H)` == numpy `(H,W,C)`).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgsr_oracle.so")
_SO64 = os.path.join(_HERE, "_build", "libgsr_oracle_f64.so")  # the float64 replay of the same source (ORC_REAL_DOUBLE)
_lib = None
_lib64 = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class OrcCamera(C.Structure):
    _fields_ = [
        ("R", C.c_float * 9), ("t", C.c_float * 3), ("focal", C.c_float * 2),
        ("principal", C.c_float * 2), ("camera_center", C.c_float * 3),
        ("width", C.c_int), ("height", C.c_int),
        ("near_plane", C.c_float), ("far_plane", C.c_float),
        ("radius_clip", C.c_int), ("blur_eps", C.c_float),
    ]


class OrcCamera64(C.Structure):
    """orc_camera as the float64 replay lays it out (every float field a double)."""
    _fields_ = [
        ("R", C.c_double * 9), ("t", C.c_double * 3), ("focal", C.c_double * 2),
        ("principal", C.c_double * 2), ("camera_center", C.c_double * 3),
        ("width", C.c_int), ("height", C.c_int),
        ("near_plane", C.c_double), ("far_plane", C.c_double),
        ("radius_clip", C.c_int), ("blur_eps", C.c_double),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gsr_oracle.c")
    if (force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src)
            or not os.path.exists(_SO64) or os.path.getmtime(_SO64) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _SO


def lib64():
    """The float64 replay (libgsr_oracle_f64.so): gsr_oracle.c compiled with every `float` a double.  Only the per-Gaussian
    functions are meaningful there; arrays cross as float64 and the camera as OrcCamera64."""
    global _lib64
    if _lib64 is None:
        build()
        _lib64 = C.CDLL(_SO64)
    return _lib64


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_cumsum.restype = C.c_int64
        _lib.orc_inverse2.restype = C.c_float
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


def _p(a, ct=C.c_float):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


@dataclass
class Camera:
    """Plain mirror of the fields `rasterize` reads from `Camera`
    (camera.jl:2-16): R,t of w2c, intrinsics (focal px, principal normalised,
    resolution), camera_center."""
    width: int
    height: int
    focal: tuple
    principal: tuple = (0.5, 0.5)
    R: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))  # row-major R[r][c]
    t: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=np.float32))
    near_plane: float = 0.2
    far_plane: float = 1000.0
    radius_clip: int = 3
    blur_eps: float = 0.3

    @property
    def camera_center(self):
        # c2w[1:3,4] = -R' t  (camera.jl get_w2c)
        R = np.asarray(self.R, np.float64)
        return (-R.T @ np.asarray(self.t, np.float64)).astype(np.float32)

    def struct(self) -> OrcCamera:
        s = OrcCamera()
        Rm = np.asarray(self.R, np.float32)
        for c in range(3):
            for r in range(3):
                s.R[c * 3 + r] = float(Rm[r, c])
        for k in range(3):
            s.t[k] = float(self.t[k])
            s.camera_center[k] = float(self.camera_center[k])
        for k in range(2):
            s.focal[k] = float(self.focal[k])
            s.principal[k] = float(self.principal[k])
        s.width, s.height = int(self.width), int(self.height)
        s.near_plane, s.far_plane = float(self.near_plane), float(self.far_plane)
        s.radius_clip, s.blur_eps = int(self.radius_clip), float(self.blur_eps)
        return s

    @property
    def grid(self):
        return ((self.width + 15) // 16, (self.height + 15) // 16)


def n_color_features(mode: str) -> int:
    """rasterizer.jl:47-51"""
    return {"rgb": 3, "rgbd": 5, "rgbdn": 8}[mode]


def feature_background(background, channels):
    """rasterizer.jl:411-414"""
    bg = np.zeros(channels, np.float32)
    bg[:3] = background
    return bg


# --------------------------------------------------------------------------
# stages
# --------------------------------------------------------------------------
def project(means, scales, rots, cam: Camera, with_normals=False):
    n = means.shape[0]
    depths = np.zeros(n, np.float32)
    radii = np.zeros(n, np.int32)
    means2d = np.zeros((n, 2), np.float32)
    conics = np.zeros((n, 3), np.float32)
    normals = np.zeros((n, 3), np.float32) if with_normals else None
    cs = cam.struct()
    lib().orc_project(C.c_int(n), _p(_f(means)), _p(_f(scales)), _p(_f(rots)), C.byref(cs),
                      _p(depths), _p(radii, C.c_int32), _p(means2d), _p(conics), _p(normals))
    return depths, radii, means2d, conics, normals


def sh_forward(radii, means, cam_center, shs, degree):
    n, K = shs.shape[0], shs.shape[1]
    rgbs = np.zeros((n, 3), np.float32)
    clamped = np.zeros((n, 3), np.uint8)
    lib().orc_sh_forward(C.c_int(n), C.c_int(K), C.c_int(degree), _p(radii, C.c_int32), _p(_f(means)),
                         _p(_f(cam_center)), _p(_f(shs)), _p(rgbs), _p(clamped, C.c_uint8))
    return rgbs, clamped


def sh_backward(means, cam_center, shs, clamped, vrgbs, degree, vmeans):
    n, K = shs.shape[0], shs.shape[1]
    vshs = np.zeros_like(shs, dtype=np.float32)
    lib().orc_sh_backward(C.c_int(n), C.c_int(K), C.c_int(degree), _p(_f(means)), _p(_f(cam_center)),
                          _p(_f(shs)), _p(np.ascontiguousarray(clamped, np.uint8), C.c_uint8), _p(_f(vrgbs)),
                          _p(vshs), _p(vmeans))
    return vshs


def count_tiles(means2d, radii, grid):
    n = radii.shape[0]
    out = np.zeros(n, np.int32)
    g = np.asarray(grid, np.int32)
    lib().orc_count_tiles(C.c_int(n), _p(means2d), _p(radii, C.c_int32), _p(g, C.c_int), _p(out, C.c_int32))
    return out


def cumsum(x):
    out = np.zeros_like(x)
    d = lib().orc_cumsum(C.c_int(x.shape[0]), _p(x, C.c_int32), _p(out, C.c_int32))
    return out, int(d)


def duplicate_with_keys(means2d, depths, offsets, radii, grid, d):
    keys = np.zeros(d, np.uint64)
    values = np.zeros(d, np.uint32)
    g = np.asarray(grid, np.int32)
    lib().orc_duplicate_with_keys(C.c_int(radii.shape[0]), _p(means2d), _p(depths), _p(offsets, C.c_int32),
                                  _p(radii, C.c_int32), _p(g, C.c_int), _p(keys, C.c_uint64), _p(values, C.c_uint32))
    return keys, values


def sort_pairs(keys, values):
    ko, vo = np.zeros_like(keys), np.zeros_like(values)
    lib().orc_sort_pairs(C.c_int64(keys.shape[0]), _p(keys, C.c_uint64), _p(values, C.c_uint32),
                         _p(ko, C.c_uint64), _p(vo, C.c_uint32))
    return ko, vo


def identify_tile_range(keys_sorted, n_tiles):
    ranges = np.zeros((n_tiles, 2), np.uint32)
    lib().orc_identify_tile_range(C.c_int64(keys_sorted.shape[0]), _p(keys_sorted, C.c_uint64),
                                  _p(ranges, C.c_uint32))
    return ranges


def render(W, H, Cn, values, means2d, opac, conics, features, ranges, background,
           want_covis=False, want_uncert=False, n=None):
    image = np.zeros((H, W, Cn), np.float32)
    n_contrib = np.zeros((H, W), np.uint32)
    accum = np.zeros((H, W), np.float32)
    covis = np.zeros(n if n is not None else means2d.shape[0], np.uint8) if want_covis else None
    unc = np.zeros((H, W), np.float32) if want_uncert else None
    lib().orc_render(C.c_int(W), C.c_int(H), C.c_int(Cn), _p(values, C.c_uint32), _p(means2d), _p(_f(opac)),
                     _p(conics), _p(_f(features)), _p(ranges, C.c_uint32), _p(_f(background)), _p(image),
                     _p(n_contrib, C.c_uint32), _p(accum), _p(covis, C.c_uint8), _p(unc))
    return image, n_contrib, accum, covis, unc


def render_bwd(W, H, Cn, n, vpixels, n_contrib, accum, values, means2d, opac, conics, features, ranges,
               background, deterministic=True):
    vfeat = np.zeros((n, Cn), np.float32)
    vopac = np.zeros(n, np.float32)
    vconics = np.zeros((n, 3), np.float32)
    vmeans2d = np.zeros((n, 2), np.float32)
    lib().orc_render_bwd(C.c_int(W), C.c_int(H), C.c_int(Cn), C.c_int(n), _p(_f(vpixels)),
                         _p(n_contrib, C.c_uint32), _p(accum), _p(values, C.c_uint32), _p(means2d), _p(_f(opac)),
                         _p(conics), _p(_f(features)), _p(ranges, C.c_uint32), _p(_f(background)),
                         _p(vfeat), _p(vopac), _p(vconics), _p(vmeans2d),
                         C.c_int(2 if deterministic == "parallel" else 1 if deterministic else 0))
    return vfeat, vopac, vconics, vmeans2d


def project_bwd(vmeans2d, vconics, vdepths, vnormals, conics, radii, means, scales, rots, cam: Camera,
                pose_grad=False):
    n = means.shape[0]
    vmeans = np.zeros((n, 3), np.float32)
    vscales = np.zeros((n, 3), np.float32)
    vrots = np.zeros((n, 4), np.float32)
    vR = np.zeros(9, np.float32) if pose_grad else None
    vt = np.zeros(3, np.float32) if pose_grad else None
    cs = cam.struct()
    lib().orc_project_bwd(C.c_int(n), _p(vmeans2d), _p(vconics),
                          _p(None if vdepths is None else _f(vdepths)),
                          _p(None if vnormals is None else _f(vnormals)),
                          _p(conics), _p(radii, C.c_int32), _p(_f(means)), _p(_f(scales)), _p(_f(rots)),
                          C.byref(cs), _p(vmeans), _p(vscales), _p(vrots), _p(vR), _p(vt))
    return vmeans, vscales, vrots, vR, vt


def project_bwd_f64(vmeans2d, vconics, vdepths, vnormals, radii, means, scales, rots, cam: Camera):
    """The per-Gaussian backward (projection.jl:132-257) REPLAYED IN FLOAT64 on the same inputs: the same C source with
    every `float` a double (gsr_oracle.c, ORC_REAL_DOUBLE), the conic re-derived from the raw inputs.  Truth for
    ∇scales / ∇rotations of needle-shaped splats, whose fp32 evaluation — the reference's, the oracle's, anybody's — loses the
    thin eigen-direction of the 2x2 covariance.  Inputs are the float32 arrays the fp32 path takes (values unchanged);
    returns float64 (vmeans, vscales, vrots)."""
    d = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    n = means.shape[0]
    vmeans = np.zeros((n, 3), np.float64)
    vscales = np.zeros((n, 3), np.float64)
    vrots = np.zeros((n, 4), np.float64)
    s32 = cam.struct()
    cs = OrcCamera64()
    for name, _ in OrcCamera64._fields_:
        v = getattr(s32, name)
        if hasattr(v, "__len__"):
            for k in range(len(v)):
                getattr(cs, name)[k] = float(v[k])
        else:
            setattr(cs, name, v)
    vm2, vc, vd, vn, me, sc, ro = d(vmeans2d), d(vconics), d(vdepths), d(vnormals), d(means), d(scales), d(rots)
    dummy_conics = np.zeros((n, 3), np.float64)  # ignored by the float64 build (it re-derives the conic)
    dp = lambda a: _p(a, C.c_double)  # noqa: E731
    lib64().orc_project_bwd(C.c_int(n), dp(vm2), dp(vc), dp(vd), dp(vn), dp(dummy_conics),
                            _p(np.ascontiguousarray(radii, np.int32), C.c_int32), dp(me), dp(sc), dp(ro), C.byref(cs),
                            dp(vmeans), dp(vscales), dp(vrots), None, None)
    return vmeans, vscales, vrots


# --------------------------------------------------------------------------
# rasterize / ∇rasterize
# --------------------------------------------------------------------------
@dataclass
class FwdState:
    """Everything `rasterize` leaves behind in the rasterizer object."""
    mode: str
    depths: np.ndarray
    radii: np.ndarray
    means2d: np.ndarray
    conics: np.ndarray
    normals: Optional[np.ndarray]
    rgbs: np.ndarray
    clamped: np.ndarray
    tiles_touched: np.ndarray
    points_offset: np.ndarray
    n_rendered: int
    keys_unsorted: Optional[np.ndarray] = None
    values_unsorted: Optional[np.ndarray] = None
    keys_sorted: Optional[np.ndarray] = None
    values_sorted: Optional[np.ndarray] = None
    ranges: Optional[np.ndarray] = None
    features: Optional[np.ndarray] = None
    image: Optional[np.ndarray] = None
    n_contrib: Optional[np.ndarray] = None
    accum_alpha: Optional[np.ndarray] = None
    covisibilities: Optional[np.ndarray] = None
    uncertainties: Optional[np.ndarray] = None


def forward(means, shs, opacities, scales, rots, cam: Camera, sh_degree: int, background=(0, 0, 0),
            mode="rgb", want_covis=False, want_uncert=False) -> FwdState:
    """rasterizer.jl:255-408.  Inputs are the *activated* opacities/scales."""
    Cn = n_color_features(mode)
    means, shs, scales, rots = _f(means), _f(shs), _f(scales), _f(rots)
    opacities = _f(opacities).reshape(-1)
    n = means.shape[0]
    W, H = cam.width, cam.height
    grid = cam.grid
    depths, radii, means2d, conics, normals = project(means, scales, rots, cam, with_normals=Cn > 5)
    rgbs, clamped = sh_forward(radii, means, cam.camera_center, shs, sh_degree)
    tiles = count_tiles(means2d, radii, grid)
    offsets, d = cumsum(tiles)
    st = FwdState(mode, depths, radii, means2d, conics, normals, rgbs, clamped, tiles, offsets, d)
    st.image = np.zeros((H, W, Cn), np.float32)  # rasterizer.jl:283 fill!(image, 0)
    st.n_contrib = np.zeros((H, W), np.uint32)
    st.accum_alpha = np.zeros((H, W), np.float32)
    st.ranges = np.zeros((grid[0] * grid[1], 2), np.uint32)
    st.values_sorted = np.zeros(0, np.uint32)
    st.features = _features(st, Cn)
    if d == 0:
        return st  # rasterizer.jl:338: all-zero image, no background
    ku, vu = duplicate_with_keys(means2d, depths, offsets, radii, grid, d)
    ks, vs = sort_pairs(ku, vu)
    st.keys_unsorted, st.values_unsorted, st.keys_sorted, st.values_sorted = ku, vu, ks, vs
    st.ranges = identify_tile_range(ks, grid[0] * grid[1])
    bg = feature_background(np.asarray(background, np.float32), Cn)
    st.image, st.n_contrib, st.accum_alpha, st.covisibilities, st.uncertainties = render(
        W, H, Cn, vs, means2d, opacities, conics, st.features, st.ranges, bg,
        want_covis=want_covis, want_uncert=want_uncert, n=n)
    return st


def _features(st: FwdState, Cn: int):
    """rasterizer.jl:380-391: rgb | depth | 1 | normal"""
    n = st.rgbs.shape[0]
    if Cn == 3:
        return st.rgbs
    f = np.zeros((n, Cn), np.float32)
    f[:, :3] = st.rgbs
    f[:, 3] = st.depths
    f[:, 4] = 1.0
    if Cn == 8:
        f[:, 5:8] = st.normals
    return f


@dataclass
class Grads:
    vmeans: np.ndarray
    vshs: np.ndarray
    vopacities: np.ndarray
    vscales: np.ndarray
    vrots: np.ndarray
    vR: Optional[np.ndarray]
    vt: Optional[np.ndarray]
    vmeans2d: np.ndarray
    vconics: np.ndarray
    vfeatures: np.ndarray


def backward(st: FwdState, vpixels, means, shs, opacities, scales, rots, cam: Camera, sh_degree: int,
             background=(0, 0, 0), pose_grad=False, deterministic=True, truth_project=False) -> Grads:
    """rasterizer.jl:416-550.  `vpixels` is (H,W,C).  `deterministic`: True = serial tile loop with double accumulators
    (truth gradients, bit-reproducible); "parallel" = the same double accumulators updated atomically from an OpenMP tile
    loop (truth gradients at large sizes; equal to True up to the order of double additions); False = the reference's
    float-atomic form (render.jl:242,275-282).
    `truth_project`: ∇scales / ∇rotations from the FLOAT64 REPLAY of the per-Gaussian backward (project_bwd_f64) on the
    double-accumulated per-Gaussian cotangents — what to compare against on scenes with needle-shaped splats, where the
    reference's fp32 ∇project is itself only good to ~1e-3 (returned as float64 in `vscales` / `vrots`; `vmeans` and
    everything else stay the fp32 restatement's)."""
    Cn = n_color_features(st.mode)
    means, shs, scales, rots = _f(means), _f(shs), _f(scales), _f(rots)
    opacities = _f(opacities).reshape(-1)
    n = means.shape[0]
    W, H = cam.width, cam.height
    bg = feature_background(np.asarray(background, np.float32), Cn)
    if st.n_rendered > 0:
        vfeat, vopac, vconics, vmeans2d = render_bwd(
            W, H, Cn, n, vpixels, st.n_contrib, st.accum_alpha, st.values_sorted, st.means2d, opacities,
            st.conics, st.features, st.ranges, bg, deterministic=deterministic)
    else:
        vfeat = np.zeros((n, Cn), np.float32); vopac = np.zeros(n, np.float32)
        vconics = np.zeros((n, 3), np.float32); vmeans2d = np.zeros((n, 2), np.float32)
    vrgbs = np.ascontiguousarray(vfeat[:, :3])
    vdepths = np.ascontiguousarray(vfeat[:, 3]) if Cn > 3 else None
    vnormals = np.ascontiguousarray(vfeat[:, 5:8]) if Cn > 5 else None
    vmeans, vscales, vrots, vR, vt = project_bwd(vmeans2d, vconics, vdepths, vnormals, st.conics, st.radii,
                                                 means, scales, rots, cam, pose_grad=pose_grad)
    if truth_project:
        _, vscales, vrots = project_bwd_f64(vmeans2d, vconics, vdepths, vnormals, st.radii, means, scales, rots, cam)
    vshs = sh_backward(means, cam.camera_center, shs, st.clamped, vrgbs, sh_degree, vmeans)
    return Grads(vmeans, vshs, vopac, vscales, vrots, vR, vt, vmeans2d, vconics, vfeat)


# --------------------------------------------------------------------------
# fused SSIM + loss head (fused_ssim.jl, training.jl:684-694)
# --------------------------------------------------------------------------
def ssim_forward(img, ref, train=True, C1=np.float32(0.01) ** 2, C2=np.float32(0.03) ** 2):
    """img/ref: numpy (B,CH,H,W) C-order == Julia (W,H,CH,B)."""
    img, ref = _f(img), _f(ref)
    B, CH, H, W = img.shape
    m = np.zeros_like(img)
    d0 = np.zeros_like(img); d1 = np.zeros_like(img); d2 = np.zeros_like(img)
    lib().orc_ssim_forward(C.c_int(W), C.c_int(H), C.c_int(CH), C.c_int(B), _p(img), _p(ref),
                           C.c_float(C1), C.c_float(C2), C.c_int(1 if train else 0), _p(m), _p(d0), _p(d1), _p(d2))
    return m, d0, d1, d2


def ssim_backward(img, ref, dL_dmap, d0, d1, d2):
    img, ref = _f(img), _f(ref)
    B, CH, H, W = img.shape
    out = np.zeros_like(img)
    lib().orc_ssim_backward(C.c_int(W), C.c_int(H), C.c_int(CH), C.c_int(B), _p(img), _p(ref), _p(_f(dL_dmap)),
                            _p(d0), _p(d1), _p(d2), _p(out))
    return out


def loss_head(image_hwc, target_chw, lambda_dssim=np.float32(0.2)):
    """training.jl:656,684-694: L = (1-λ)·mean|x-y| + λ·(1-mean(SSIM)); returns
    (loss, vpixels (H,W,3)).  image_hwc: (H,W,C>=3); target: (3,H,W)."""
    x = np.ascontiguousarray(np.transpose(image_hwc[:, :, :3], (2, 0, 1)))[None]  # (1,3,H,W)
    y = _f(target_chw)[None]
    npx = np.float32(x.size)
    l1 = np.abs(x - y).mean(dtype=np.float32)
    m, d0, d1, d2 = ssim_forward(x, y, train=True)
    s = np.float32(1.0) - m.mean(dtype=np.float32)
    lam = np.float32(lambda_dssim)
    loss = (np.float32(1.0) - lam) * l1 + lam * s
    dL_dmap = np.full_like(m, -lam / npx)
    g = ssim_backward(x, y, dL_dmap, d0, d1, d2)
    g = g + (np.float32(1.0) - lam) * np.sign(x - y).astype(np.float32) / npx
    vpix = np.zeros_like(image_hwc)
    vpix[:, :, :3] = np.transpose(g[0], (1, 2, 0))
    return np.float32(loss), vpix


def update_stats(max_radii, accum, denom, radii, vmeans2d, width, height):
    """_update_stats! (src/strategy.jl:118-136), in place on numpy arrays."""
    vis = radii > 0
    max_radii[vis] = np.maximum(max_radii[vis], radii[vis])
    g = _f(vmeans2d)
    gx = g[:, 0] * np.float32(width) * np.float32(0.5)
    gy = g[:, 1] * np.float32(height) * np.float32(0.5)
    nrm = np.sqrt(gx * gx + gy * gy, dtype=np.float32)
    accum[vis] += nrm[vis]
    denom[vis] += np.float32(1.0)


# ---- functor prologue (rasterizer.jl:200-253) and the optimizer step (training.jl:234-239,778) ----
def prologue_forward(sh_color, sh_remainder, opacities, scales):
    """(N,1,3), (N,KR,3) or None, (N,1), (N,3)|(N,1) raw -> shs (N,1+KR,3), σ(opacities) (N,1), exp(scales) (N,3)."""
    sh_color = _f(sh_color)
    n = sh_color.shape[0]
    kr = 0 if sh_remainder is None else int(_f(sh_remainder).shape[1])
    rem = _f(sh_remainder) if kr else np.zeros(1, np.float32)
    opacities, scales = _f(opacities), _f(scales)
    sd = 1 if scales.size == n else 3
    shs = np.empty((n, 1 + kr, 3), np.float32)
    oa = np.empty((n, 1), np.float32)
    sa = np.empty((n, 3), np.float32)
    lib().orc_prologue_forward(n, kr, sd, _p(sh_color), _p(rem), _p(opacities), _p(scales), _p(shs), _p(oa), _p(sa))
    return shs, oa, sa


def prologue_backward(opacities_act, scales_act, vshs, vopacities_act, vscales_act, scale_dims=3):
    vshs = _f(vshs)
    n, K = vshs.shape[0], vshs.shape[1]
    kr = K - 1
    vdc = np.empty((n, 1, 3), np.float32)
    vrest = np.empty((n, max(kr, 1), 3), np.float32)
    vo = np.empty((n, 1), np.float32)
    vs = np.empty((n, scale_dims), np.float32)
    lib().orc_prologue_backward(n, kr, scale_dims, _p(_f(opacities_act)), _p(_f(scales_act)), _p(vshs),
                                _p(_f(vopacities_act)), _p(_f(vscales_act)), _p(vdc), _p(vrest), _p(vo), _p(vs))
    return vdc, (vrest if kr else vrest[:, :0]), vo, vs


def adam_step(theta, grad, mu, nu, step, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """In place on float32 numpy arrays (theta, mu, nu); `step` is the counter AFTER increment."""
    for a in (theta, mu, nu):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    L = lib()
    L.orc_adam_step.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_adam_step(theta.size, theta.ctypes.data, _f(grad).ctypes.data, mu.ctypes.data, nu.ctypes.data,
                    int(step), float(lr), float(beta1), float(beta2), float(eps))


def adam_lr_t(lr, beta1, beta2, step):
    L = lib()
    L.orc_adam_lr_t.restype = C.c_float
    L.orc_adam_lr_t.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32]
    return L.orc_adam_lr_t(float(lr), float(beta1), float(beta2), int(step))


# ---- boolean-mask compaction (src/densification.jl:138-191,279-288): x[:, mask] is plain logical indexing ----
def findall(mask):
    """`findall(mask)`, 0-based."""
    return np.flatnonzero(np.asarray(mask)).astype(np.int32)


def select_rows(x, idx):
    """`x[:, idxs]` / `x[:, :, idxs]` for the C-order equivalent array (Gaussian index first)."""
    return np.ascontiguousarray(np.asarray(x)[np.asarray(idx)])
