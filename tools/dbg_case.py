import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gsr_pkg; pkg = gsr_pkg.load()
from oracle import oracle as orc
from hip_helpers import HipRun, rel_l2, blend_boundary_pixels
case = int(sys.argv[1])
rng = np.random.default_rng(9000 + case)
mode = ["rgb", "rgbd", "rgbdn"][case % 3]
deg = int(rng.integers(0, 4))
W, H = int(rng.integers(20, 140)), int(rng.integers(20, 110))
n = int(rng.integers(1, 1500))
s = pkg.synthetic.make_scene(n, W, H, deg, 9100 + case, sigma_px=float(rng.uniform(1.5, 9.0)), K=16 if case % 4 == 0 else None)
opac = (s.opacities * rng.uniform(0.05, 1.0)).astype(np.float32) if case % 2 else s.opacities
Rm, t = pkg.synthetic.view_pose(int(rng.integers(0, 8)))
cam = orc.Camera(W, H, s.focal, R=Rm, t=t, principal=(float(rng.uniform(0.4, 0.6)), float(rng.uniform(0.4, 0.6))))
bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
print("case", case, mode, "deg", deg, W, H, "n", n)
st = orc.forward(s.means, s.shs, opac, s.scales, s.rotations, cam, deg, background=bg, mode=mode)
print("n_rendered", st.n_rendered, "visible", int((st.radii > 0).sum()))
for cull in (False, True):
    run = HipRun(pkg, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, bg, mode, exact_tile_cull=cull)
    img = run.forward().clone().cpu().numpy()
    Tt = run.rast.accum_alpha.cpu().numpy()
    d = np.abs(Tt - st.accum_alpha)
    print("cull", cull, "max dT", d.max(), "n bad px", int((d > 1e-4).sum()), "img max d", np.abs(img - st.image).max(),
          "boundary px", int(blend_boundary_pixels(st, opac, W, H).sum()), "n_contrib diff", int((run.rast.n_contrib.cpu().numpy().astype(np.uint32) != st.n_contrib).sum()))
    C = st.image.shape[2]
    rng2 = np.random.default_rng(9000 + case)
    vp = np.random.default_rng(5).standard_normal((H, W, C)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, background=bg)
    out = [None if o is None else o.cpu().numpy() for o in run.backward(vp)]
    names = ["vmeans", "vshs", "vopacities", "vscales", "vrots"]
    refs = [g.vmeans, g.vshs, g.vopacities, g.vscales, g.vrots]
    for nm, o, r in zip(names, out, refs):
        o = o.reshape(r.shape)
        e = np.abs(o - r)
        idx = np.unravel_index(np.argmax(e), e.shape)
        print("   ", nm, "rel_l2 %.3e" % rel_l2(o, r), "max abs err %.3e at %s (ref %.4e got %.4e) norm %.3e" % (e.max(), idx, r[idx], o[idx], np.linalg.norm(r)))
