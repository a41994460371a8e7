"""Round 6 (round-5 verdict, next #7: two-level binning at 4K, bar config-5 `preprocess` <= 0.45 ms from 0.59): what is the floor?
Times the `preprocess` stage (HIP events of the library's stage profiler) of forward-only views at configs 3 and 5 — with the library
given by GSR_HIP_LIB; a build with -DGSR_PRE_NO_BINNING projects, evaluates SH and writes the records but emits no instance."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gsr_pkg
import torch

pkg = gsr_pkg.load()
for name, n, W, H, order in (("config3", 1_000_000, 1920, 1080, "random"), ("config5", 5_000_000, 3840, 2160, "random"),
                             ("config5 morton", 5_000_000, 3840, 2160, "morton")):
    s = pkg.synthetic.make_scene(n, W, H, 3, 1003)
    if order == "morton":
        s = pkg.synthetic.reorder(s, pkg.synthetic.morton_order(s.means))
    cam = pkg.Camera(W, H, tuple(s.focal))
    t = [torch.from_numpy(x).cuda() for x in (s.means, s.shs, s.opacities.reshape(-1, 1), s.scales, s.rotations)]
    for form in ("direct", "aggregating"):
        rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", preprocess_form=form, form_tuner=False)
        for _ in range(5):
            rast.forward_raw(*t, cam, 3, (0, 0, 0), forward_only=True)
        torch.cuda.synchronize()
        rast.profile(True)
        k = 20
        for _ in range(k):
            rast.forward_raw(*t, cam, 3, (0, 0, 0), forward_only=True)
        torch.cuda.synchronize()
        pr = rast.profile_read()
        st = rast.stats
        print(f"{os.environ.get('GSR_HIP_LIB', 'default').split('/')[-1]} {name} form={form} ran={int(st.preprocess_form)} D={int(st.n_rendered)} "
              f"preprocess={pr['preprocess'][0] / k:.4f} ms", flush=True)
        rast.close()
