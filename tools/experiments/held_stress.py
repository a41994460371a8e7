#!/usr/bin/env python3
"""Soak test of the forward's stream choreography (held fused launch + tier walk on the second stream, overflow scatter, long-list
backward on the second stream): N views in random order over scenes with no / mid / big / overflowing tier tiles on ONE handle, with
side outputs; every image, list, side output and gradient must equal the scene's reference render bit for bit.
  python tools/experiments/held_stress.py [N = 3000] [seed = 1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gsr_pkg  # noqa: E402
from hip_helpers import HipRun, dev  # noqa: E402
from oracle import oracle as orc  # noqa: E402

pkg = gsr_pkg.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W, H, deg = 480, 272, 1
base = pkg.synthetic.make_scene(20000, W, H, deg, 501)
T = ((W + 15) // 16) * ((H + 15) // 16)
scenes = [base, pkg.synthetic.add_skew(base, "hot:3000", seed=502), pkg.synthetic.add_skew(base, "hot:12000", seed=503),
          pkg.synthetic.add_skew(pkg.synthetic.add_skew(base, "dense:0.02:40", seed=504), "hot:9000", seed=505)]
cam = orc.Camera(W, H, base.focal)
for mode, budget in (("rgb", 0), ("rgbd", (T + 1) * 8 * 1536)):
    C = {"rgb": 3, "rgbd": 5}[mode]
    vp = np.random.default_rng(7).standard_normal((H, W, C)).astype(np.float32)
    tensors = [[dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)] for s in scenes]
    ref = []
    for s in scenes:
        r = HipRun(pkg, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, (0.1, 0.2, 0.3), mode, exact_tile_cull=True,
                   want_covis=True, want_uncert=True)
        img = r.forward().clone()
        ref.append((img, [g.clone() for g in r.backward(vp)[:5]], r.covis.clone(), r.unc.clone(), r.rast.values_sorted.clone(),
                    int(r.rast.stats.max_tile_instances)))
        r.rast.close()
    one = HipRun(pkg, base.means, base.shs, base.opacities, base.scales, base.rotations, cam, deg, (0.1, 0.2, 0.3), mode,
                 exact_tile_cull=True, bins_budget_bytes=budget)
    one.unc = torch.zeros(H, W, device="cuda")
    bad, modes = 0, {0: 0, 1: 0, 2: 0}
    for it in range(N):
        k = int(rng.integers(0, len(scenes)))
        one.t = tensors[k]
        one.covis = torch.zeros(scenes[k].means.shape[0], dtype=torch.uint8, device="cuda")
        img = one.forward()
        modes[int(one.rast.stats.compact_binning)] += 1
        ok = torch.equal(img, ref[k][0]) and torch.equal(one.covis, ref[k][2]) and torch.equal(one.unc, ref[k][3]) and \
            torch.equal(one.rast.values_sorted, ref[k][4])
        if rng.integers(0, 3):   # (some views are not differentiated)
            ok = ok and all(torch.equal(a, b) for a, b in zip(one.backward(vp)[:5], ref[k][1]))
        if not ok:
            bad += 1
            print("MISMATCH view", it, "scene", k, flush=True)
    print(f"{mode}: {N - bad} / {N} views identical to the reference renders; longest lists {[r[5] for r in ref]}; binning modes {modes}")
    one.rast.close()
sys.exit(1 if bad else 0)
