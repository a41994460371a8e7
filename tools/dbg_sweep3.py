"""On the GPU box: failing cases of tools/fuzz_parity.py's randomised sweep — oracle and kernels against the float64
autograd model (tests/f64_model.py), and where the difference sits.   python tools/dbg_sweep3.py CASE ..."""
import os, sys, numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import gsr_pkg; pkg = gsr_pkg.load()
from oracle import oracle as orc
from hip_helpers import HipRun, blend_boundary_pixels
import f64_model as fm
DT = torch.float64
rel = lambda a, b: np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)
for case in [int(a) for a in sys.argv[1:]]:
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
    st = orc.forward(s.means, s.shs, opac, s.scales, s.rotations, cam, deg, background=bg, mode=mode)
    C = st.image.shape[2]
    vp = rng.standard_normal((H, W, C)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, background=bg)
    run = HipRun(pkg, s.means, s.shs, opac, s.scales, s.rotations, cam, deg, bg, mode)
    run.forward()
    out = [None if o is None else o.cpu().numpy() for o in run.backward(vp)]
    tt = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=DT, requires_grad=True)
    means, shs, op, sc, rots = tt(s.means), tt(s.shs), tt(opac), tt(s.scales), tt(s.rotations)
    img = fm.render_dense(means, shs, op, sc, rots, cam, deg, np.asarray(bg, np.float32), mode, st.values_sorted, st.ranges, st.radii)
    (img * torch.tensor(vp, dtype=DT)).sum().backward()
    vis = st.radii > 0
    print("case", case, mode, "deg", deg, W, H, "n", n, "visible", int(vis.sum()), "boundary px", int(blend_boundary_pixels(st, opac, W, H).sum()))
    for nm, o, r, tr in (("vmeans", out[0], g.vmeans, means.grad.numpy()), ("vshs", out[1], g.vshs, shs.grad.numpy()), ("vopac", out[2], g.vopacities, op.grad.numpy()),
                         ("vscales", out[3], g.vscales, sc.grad.numpy()), ("vrots", out[4], g.vrots, rots.grad.numpy())):
        o = o.reshape(r.shape); tr = tr.reshape(r.shape)
        e = np.abs(o - r).reshape(r.shape[0], -1).max(1); worst = int(np.argmax(e))
        share = float((np.abs(o - r).reshape(r.shape[0], -1)[worst] ** 2).sum() / max((np.abs(o - r) ** 2).sum(), 1e-300))
        print("   %-8s oracle vs f64 %.2e  HIP vs f64 %.2e  HIP vs oracle %.2e | %.0f %% of the squared difference sits on Gaussian %d (opacity %.3f, scales %s)" % (
            nm, rel(r[vis], tr[vis]), rel(o[vis], tr[vis]), rel(o[vis], r[vis]), 100 * share, worst, opac[worst], np.round(s.scales[worst], 3)))
