"""On the GPU box: are the gradient differences on needle-shaped Gaussians (600 : 1) a kernel bug or fp32 conditioning?
Oracle and HIP kernels against the float64 autograd model of tests/f64_model.py on the same scenes.   python tools/dbg_needle.py"""
import os, sys, numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import gsr_pkg; pkg = gsr_pkg.load()
from oracle import oracle as orc
from hip_helpers import HipRun
import f64_model as fm
DT = torch.float64
for seed in (() if len(sys.argv) > 1 else (1, 2, 3)):
    rng = np.random.default_rng(seed)
    W, H, n, deg, mode = 64, 48, 150, 1, "rgb"
    s = pkg.synthetic.make_scene(n, W, H, deg, 500 + seed, sigma_px=4.0)
    scales = s.scales.copy()
    m = rng.random(n) < 0.15
    scales[m, 0] *= 12.0; scales[m, 1] *= 0.02
    R, t = pkg.synthetic.view_pose(2)
    cam = orc.Camera(W, H, s.focal, R=R, t=t)
    bg = np.array([0.3, 0.1, 0.6], np.float32)
    st = orc.forward(s.means, s.shs, s.opacities, scales, s.rotations, cam, deg, background=bg, mode=mode)
    vp = rng.standard_normal((H, W, 3)).astype(np.float32)
    g = orc.backward(st, vp, s.means, s.shs, s.opacities, scales, s.rotations, cam, deg, background=bg)
    run = HipRun(pkg, s.means, s.shs, s.opacities, scales, s.rotations, cam, deg, tuple(bg), mode)
    run.forward()
    out = [None if o is None else o.cpu().numpy() for o in run.backward(vp)]
    tt = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=DT, requires_grad=True)
    means, shs, opac, sc, rots = tt(s.means), tt(s.shs), tt(s.opacities), tt(scales), tt(s.rotations)
    img = fm.render_dense(means, shs, opac, sc, rots, cam, deg, bg, mode, st.values_sorted, st.ranges, st.radii)
    (img * torch.tensor(vp, dtype=DT)).sum().backward()
    vis = st.radii > 0
    rel = lambda a, b: np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)
    print("seed", seed, "needles visible", int((m & vis).sum()))
    for nm, o, r, tr in (("vmeans", out[0], g.vmeans, means.grad.numpy()), ("vscales", out[3], g.vscales, sc.grad.numpy()), ("vrots", out[4], g.vrots, rots.grad.numpy())):
        o = o.reshape(r.shape)
        print("   %-8s oracle vs f64 %.3e   HIP vs f64 %.3e   HIP vs oracle %.3e   (needles only: %.3e / %.3e / %.3e)" % (
            nm, rel(r[vis], tr[vis]), rel(o[vis], tr[vis]), rel(o[vis], r[vis]), rel(r[m & vis], tr[m & vis]), rel(o[m & vis], tr[m & vis]), rel(o[m & vis], r[m & vis])))

# python tools/dbg_needle.py edge CASE ...: the same three-way comparison on scenes of tools/fuzz_parity.py's edge mode
if len(sys.argv) > 2 and sys.argv[1] == "edge":
    sys.path.insert(0, os.path.join(R_, "tools"))
    import fuzz_parity as F, inspect
    setup = inspect.getsource(F.edge_case).split("    run = HipRun(")[0].replace("def edge_case(case):\n", "")
    for case in [int(a) for a in sys.argv[2:]]:
        ns = dict(np=np, pkg=pkg, orc=orc, case=case)
        exec("\n".join(l[4:] for l in setup.splitlines()), ns)
        st, cam, mode, deg = ns["st"], ns["cam"], ns["mode"], ns["deg"]
        means_, shs_, opac_, scales_, s, bg, W, H = ns["means"], ns["shs"], ns["opac"], ns["scales"], ns["s"], ns["bg"], ns["W"], ns["H"]
        C = st.image.shape[2]
        vp = np.random.default_rng(1).standard_normal((H, W, C)).astype(np.float32)
        g = orc.backward(st, vp, means_, shs_, opac_, scales_, s.rotations, cam, deg, background=bg)
        run = HipRun(pkg, means_, shs_, opac_, scales_, s.rotations, cam, deg, bg, mode)
        run.forward()
        out = [None if o is None else o.cpu().numpy() for o in run.backward(vp)]
        tt = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=DT, requires_grad=True)
        means, shs, opac, sc, rots = tt(means_), tt(shs_), tt(opac_), tt(scales_), tt(s.rotations)
        img = fm.render_dense(means, shs, opac, sc, rots, cam, deg, np.asarray(bg, np.float32), mode, st.values_sorted, st.ranges, st.radii)
        (img * torch.tensor(vp, dtype=DT)).sum().backward()
        vis = st.radii > 0
        rel = lambda a, b: np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)
        print("edge case", case, mode, W, H, "visible", int(vis.sum()))
        for nm, o, r, tr in (("vmeans", out[0], g.vmeans, means.grad.numpy()), ("vopac", out[2], g.vopacities, opac.grad.numpy()), ("vscales", out[3], g.vscales, sc.grad.numpy()), ("vrots", out[4], g.vrots, rots.grad.numpy())):
            o = o.reshape(r.shape); tr = tr.reshape(r.shape)
            print("   %-8s oracle vs f64 %.3e   HIP vs f64 %.3e   HIP vs oracle %.3e" % (nm, rel(r[vis], tr[vis]), rel(o[vis], tr[vis]), rel(o[vis], r[vis])))
