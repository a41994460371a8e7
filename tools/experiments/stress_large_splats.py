import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import gsr_pkg
pkg = gsr_pkg.load()
W, H, deg = 1920, 1080, 3
for N, sig in ((100_000, 30.0), (20_000, 120.0), (1_000_000, 3.0)):
    s = pkg.synthetic.make_scene(N, W, H, deg, 7, sigma_px=sig)
    cam = pkg.Camera(W, H, tuple(s.focal))
    to = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    p = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", exact_tile_cull=True)
    vp = to(pkg.synthetic.make_vpixels(W, H, 3, 1))
    for it in range(3):
        rast.forward_raw(*p, cam, deg, (0, 0, 0)); rast.backward_raw(vp, *p, cam, deg, (0, 0, 0))
    torch.cuda.synchronize()
    rast.profile(True)
    t0 = time.perf_counter()
    for it in range(5):
        rast.forward_raw(*p, cam, deg, (0, 0, 0)); rast.backward_raw(vp, *p, cam, deg, (0, 0, 0))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    pr = rast.profile_read()
    print(f"N={N} sigma_px={sig}: D={rast.stats.n_rendered} max_tile={rast.stats.max_tile_instances} step {dt*1e3:.2f} ms",
          {k: round(v[0] / max(v[1], 1), 3) for k, v in pr.items() if v[1]})
