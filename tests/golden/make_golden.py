#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

These fixtures are produced by the CPU ORACLE (oracle/gsr_oracle.c), not by the
reference: the reference is GPU-only Julia and cannot run in the build image
(SURVEY.md §0, §8c), and it ships no golden vectors of its own.  They freeze inputs +
every intermediate + all gradients of three tiny scenes (one per render mode), one
SSIM case and one trainer-tail case (prologue + Adam), so that (a) the oracle cannot drift silently and (b) the HIP path is checked
against committed numbers on the GPU box, where /root/reference does not exist.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gsr_pkg  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("rgb", 3, 201, 64, 48, 150), ("rgbd", 1, 202, 64, 48, 150), ("rgbdn", 2, 203, 48, 40, 120)]


def scene_case(pkg, mode, deg, seed, W, H, n):
    s = pkg.synthetic.make_scene(n, W, H, deg, seed, sigma_px=4.0)
    R, t = pkg.synthetic.view_pose(2)
    cam = orc.Camera(W, H, s.focal, R=R, t=t)
    bg = np.array([0.25, 0.5, 0.125], np.float32)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, background=bg, mode=mode)
    C = st.image.shape[2]
    vp = np.random.default_rng(seed).standard_normal((H, W, C)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, background=bg,
                     pose_grad=True)
    return dict(
        mode=mode, sh_degree=deg, width=W, height=H, focal=np.asarray(s.focal, np.float32), R=R, t=t, background=bg,
        means=s.means, shs=s.shs, opacities=s.opacities, scales=s.scales, rotations=s.rotations, vpixels=vp,
        radii=st.radii, means2d=st.means2d, conics=st.conics, depths=st.depths, rgbs=st.rgbs, clamped=st.clamped,
        tiles_touched=st.tiles_touched, n_rendered=st.n_rendered, ranges=st.ranges, values_sorted=st.values_sorted,
        image=st.image, n_contrib=st.n_contrib, accum_alpha=st.accum_alpha,
        normals=st.normals if st.normals is not None else np.zeros((0, 3), np.float32),
        vmeans=g.vmeans, vshs=g.vshs, vopacities=g.vopacities, vscales=g.vscales, vrots=g.vrots, vR=g.vR, vt=g.vt,
        vmeans2d=g.vmeans2d)


def ssim_case():
    rng = np.random.default_rng(204)
    x = rng.uniform(size=(1, 3, 37, 53)).astype(np.float32)
    y = rng.uniform(size=(1, 3, 37, 53)).astype(np.float32)
    m, d0, d1, d2 = orc.ssim_forward(x, y, train=True)
    dl = rng.standard_normal(x.shape).astype(np.float32)
    g = orc.ssim_backward(x, y, dl, d0, d1, d2)
    img = np.ascontiguousarray(np.transpose(x[0], (1, 2, 0)))
    loss, vp = orc.loss_head(img, y[0])
    return dict(img=x, ref=y, ssim_map=m, dm_dmu1=d0, dm_dsigma1_sq=d1, dm_dsigma12=d2, dL_dmap=dl, dL_dimg=g,
                loss=loss, vpixels=vp)


def trainer_case():
    """Functor prologue + pullback and three Adam steps (trainer tail)."""
    rng = np.random.default_rng(205)
    n, kr = 97, 15
    dc = rng.normal(size=(n, 1, 3)).astype(np.float32)
    rest = rng.normal(size=(n, kr, 3)).astype(np.float32)
    o = (rng.normal(size=(n, 1)) * 3).astype(np.float32)
    sc = rng.normal(size=(n, 3)).astype(np.float32)
    shs, oa, sa = orc.prologue_forward(dc, rest, o, sc)
    vshs, voa, vsa = (rng.normal(size=a.shape).astype(np.float32) for a in (shs, oa, sa))
    vdc, vrest, vo, vs = orc.prologue_backward(oa, sa, vshs, voa, vsa, 3)
    theta = rng.normal(size=n * kr * 3).astype(np.float32)
    th, mu, nu = theta.copy(), np.zeros_like(theta), np.zeros_like(theta)
    grads = [(rng.normal(size=theta.size) * 10.0 ** rng.uniform(-3, 1)).astype(np.float32) for _ in range(3)]
    for k, g in enumerate(grads, 1):
        orc.adam_step(th, g, mu, nu, k, 2.5e-3 / 20, 0.9, 0.999, 1e-15)
    return dict(sh_color=dc, sh_remainder=rest, opacities=o, scales=sc, shs=shs, opacities_act=oa, scales_act=sa,
                vshs=vshs, vopacities_act=voa, vscales_act=vsa, v_sh_color=vdc, v_sh_remainder=vrest, v_opacities=vo,
                v_scales=vs, theta0=theta, grads=np.stack(grads), lr=np.float32(2.5e-3 / 20), theta3=th, mu3=mu, nu3=nu)


def config1_case(pkg):
    """BASELINE.json configs[0] ("CPU restatement only ... image hash + timings"):
    10 k Gaussians, SH degree 0, 640x480, forward only, seed 1001.  Frozen: the integer outputs in
    full (as checksums), every 4th pixel of the image, and the sha256 of the full image bytes (the hash
    is informational across machines: glibc's expf is IFUNC-dispatched per CPU)."""
    import hashlib
    import time
    W, H, n, deg, seed = 640, 480, 10_000, 0, 1001
    s = pkg.synthetic.make_scene(n, W, H, deg, seed)
    cam = orc.Camera(W, H, s.focal)
    t0 = time.perf_counter()
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    dt = time.perf_counter() - t0
    print(f"config 1: oracle forward {dt * 1e3:.1f} ms on {orc.num_threads()} threads, "
          f"{W * H / dt / 1e6:.2f} Mpixels/s, D = {st.n_rendered}")
    return dict(image_sub=st.image[::4, ::4].copy(), n_rendered=st.n_rendered, radii_sum=int(st.radii.sum()),
                n_visible=int((st.radii > 0).sum()), values_sorted_sum=int(st.values_sorted.astype(np.int64).sum()),
                n_contrib_sum=int(st.n_contrib.astype(np.int64).sum()),
                image_sha256=hashlib.sha256(st.image.tobytes()).hexdigest())


def main():
    """`make_golden.py [name ...]` regenerates only the named fixtures (config1, scene_rgb, ..., ssim, trainer)."""
    pkg = gsr_pkg.load()
    only = set(sys.argv[1:])
    want = lambda name: not only or name in only  # noqa: E731
    if want("config1"):
        np.savez_compressed(os.path.join(HERE, "config1.npz"), **config1_case(pkg))
    for mode, deg, seed, W, H, n in CASES:
        if want(f"scene_{mode}"):
            np.savez_compressed(os.path.join(HERE, f"scene_{mode}.npz"), **scene_case(pkg, mode, deg, seed, W, H, n))
    if want("ssim"):
        np.savez_compressed(os.path.join(HERE, "ssim.npz"), **ssim_case())
    if want("trainer"):
        np.savez_compressed(os.path.join(HERE, "trainer.npz"), **trainer_case())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
