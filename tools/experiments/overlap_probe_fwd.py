"""Round-4 experiment (after 4b): can an HBM-bound pass ride under `preprocess`?  preprocess = projection + SH colours (69 us) +
returning atomics (88 us) + key stores (16 us), and the three ADD — the atomics do not hide the SH loads inside one kernel.
Here the forward of config 3 runs on stream A and a STREAM triad of ~0.2 GB (the SH coefficients' size, ~40 us alone) starts on
stream B at the same moment: if `together` ~ `A alone`, SH evaluation as a second, concurrent kernel would come for free.
python tools/experiments/overlap_probe_fwd.py [triad elements]"""

import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import gsr_pkg
pkg = gsr_pkg.load()
lib = pkg._lib.load()
W, H, N, deg = 1920, 1080, 1_000_000, 3
s = pkg.synthetic.make_scene(N, W, H, deg, 1003)
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
cam = pkg.Camera(W, H, tuple(s.focal))
tgt = to(pkg.synthetic.make_target(W, H, 1003))
rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", device=dev)
arena = torch.empty((11 + 3 * 16) * N, device=dev)
n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 1024 * 1024   # 12 B/element -> 0.2 GB
ta, tb, tc = (torch.ones(n_tri, device=dev) for _ in range(3))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ev = lambda: torch.cuda.Event(enable_timing=True)


def fwd():
    img = rast.forward_raw(*t, cam, deg, (0, 0, 0))
    _, vp = pkg.fused_ssim.l1_ssim_loss(rast, img, tgt)
    return vp


def bwd(vp):
    rast.backward_raw(vp, *t, cam, deg, (0, 0, 0), arena=arena)


def triad(stream):
    pkg._lib.check(lib.gsr_stream_triad(ta.data_ptr(), tb.data_ptr(), tc.data_ptr(), n_tri, 0.5, stream.cuda_stream))


def run(mode, iters=30):
    tot = 0.0
    for _ in range(iters):
        torch.cuda.synchronize()
        e0, ea, eb = ev(), ev(), ev()
        e0.record(sa)
        sb.wait_event(e0)
        if mode in ("A", "AB"):
            with torch.cuda.stream(sa):
                rast.forward_raw(*t, cam, deg, (0, 0, 0))
        if mode in ("B", "AB"):
            triad(sb)
        ea.record(sa); eb.record(sb)
        torch.cuda.synchronize()
        tot += max(e0.elapsed_time(ea), e0.elapsed_time(eb))
    return tot / iters


for m in ("A", "B", "AB"):
    run(m, 5)
a, b, ab = run("A"), run("B"), run("AB")
print(f"forward alone {a:.4f} ms; triad ({12 * n_tri / 1e9:.2f} GB) alone {b:.4f} ms = {12 * n_tri / b / 1e6:.0f} GB/s; "
      f"together {ab:.4f} ms -> hidden {a + b - ab:.4f} ms of {b:.4f} ({100 * (a + b - ab) / b:.0f} %)")
