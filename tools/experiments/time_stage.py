"""Stage times of the forward on config 3's scene for A/B builds (GSR_HIP_LIB selects the library):
    python tools/experiments/time_stage.py [stage ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import gsr_pkg
pkg = gsr_pkg.load()
W, H, N, deg = 1920, 1080, 1_000_000, 3
s = pkg.synthetic.make_scene(N, W, H, deg, 1003)
if os.environ.get("GSR_ORDER") == "morton":
    s = pkg.synthetic.reorder(s, pkg.synthetic.morton_order(s.means))
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
cam = pkg.Camera(W, H, tuple(s.focal))
rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", device=dev)
for _ in range(3):
    rast.forward_raw(*t, cam, deg, (0, 0, 0))
torch.cuda.synchronize()
rast.profile(True)
for _ in range(10):
    rast.forward_raw(*t, cam, deg, (0, 0, 0))
torch.cuda.synchronize()
p = rast.profile_read()
print(os.environ.get("GSR_HIP_LIB", "default").split("/")[-1], "D", int(rast.stats.n_rendered),
      " ".join(f"{k}={ms / max(c, 1):.4f}" for k, (ms, c) in p.items() if c and (not sys.argv[1:] or k in sys.argv[1:])))
