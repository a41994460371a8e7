"""Round 6 fuzz campaign, edge 10426: torch.equal(exact-cull render, reference-lists render) failed INSIDE the campaign and does
not when the case runs alone.  Replays the campaign's HIP side only (no oracle): cases c0 .. c1 in order, each scene rendered
in both list modes exactly as tools/fuzz_parity.py does (fresh handles, first view), then AGAIN; any mismatch is printed with
the pixels, which of the renders changed between the two tries, and the oracle's value there."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import gsr_pkg
pkg = gsr_pkg.load()
from oracle import oracle as orc
import fuzz_scenes
from hip_helpers import HipRun
c0, c1 = int(sys.argv[1]), int(sys.argv[2])
with_bwd = len(sys.argv) > 3
bad = 0
for case in range(c0, c1 + 1):
    fs = fuzz_scenes.edge_scene(pkg, case)
    W, H = fs.cam.width, fs.cam.height
    def render(cull):
        r = HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, fs.mode, exact_tile_cull=cull,
                   pose_dev=(fs.pose and not cull))
        img = r.forward().clone()
        if with_bwd:
            r.backward(fs.cotangent())
        return r, img
    ra, a = render(False)
    rb, b = render(True)
    if not torch.equal(a, b):
        bad += 1
        ra2, a2 = render(False)
        rb2, b2 = render(True)
        d = (a - b).abs().amax(-1)
        ys, xs = torch.nonzero(d > 0, as_tuple=True)
        print(f"case {case}: {ys.numel()} pixels differ (max {float(d.max()):.3e}); second try: ref equal to first {torch.equal(a, a2)}, cull equal to first "
              f"{torch.equal(b, b2)}, second tries equal to each other {torch.equal(a2, b2)}; scene n {fs.means.shape[0]} {W}x{H} {fs.mode}", flush=True)
        st = orc.forward(fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, mode=fs.mode)
        for y, x in list(zip(ys.tolist(), xs.tolist()))[:8]:
            print(f"   ({x},{y}) tile ({x//16},{y//16}) ref {a[y,x].tolist()} cull {b[y,x].tolist()} oracle {st.image[y,x].tolist()} "
                  f"n_contrib ref {int(ra.rast.n_contrib[y,x])} cull {int(rb.rast.n_contrib[y,x])} oracle {int(st.n_contrib.reshape(H, W)[y,x])}")
        print("   stats ref", ra.rast.stats.history(), int(ra.rast.stats.compact_binning), int(ra.rast.stats.preprocess_form), "cull", rb.rast.stats.history(),
              int(rb.rast.stats.compact_binning), int(rb.rast.stats.preprocess_form))
print(f"cases {c0}..{c1}: {bad} mismatching", flush=True)
