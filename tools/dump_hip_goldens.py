#!/usr/bin/env python3
"""The HIP path's counterpart of julia/dump_reference_goldens.jl: renders the three committed golden scenes
(tests/golden/scene_{rgb,rgbd,rgbdn}.npz) through libgsr_hip.so (reference-list mode) and writes `ref_scene_<mode>.npz` in
the SAME format the Julia script writes, so that tools/compare_reference_dump.py — the tool a maintainer with Julia uses
to pin the oracle against the live reference — can be exercised end to end here (tests/test_golden.py) and its expected
console output shown in INTEGRATION.md.

    python tools/dump_hip_goldens.py tests/golden out_dir && python tools/compare_reference_dump.py tests/golden out_dir
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsr_pkg  # noqa: E402


def main(golden_dir, out_dir):
    pkg = gsr_pkg.load()
    os.makedirs(out_dir, exist_ok=True)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()  # noqa: E731
    for mode in ("rgb", "rgbd", "rgbdn"):
        f = np.load(os.path.join(golden_dir, f"scene_{mode}.npz"))
        W, H, deg = int(f["width"]), int(f["height"]), int(f["sh_degree"])
        cam = pkg.Camera(W, H, tuple(float(x) for x in f["focal"]), (0.5, 0.5), f["R"], f["t"])
        rast = pkg.rasterizer.GaussianRasterizer(W, H, mode=mode, exact_tile_cull=False)
        t = [dev(f["means"]), dev(f["shs"]), dev(f["opacities"].reshape(-1, 1)), dev(f["scales"]), dev(f["rotations"])]
        bg = tuple(float(b) for b in f["background"])
        img = rast.forward_raw(*t, cam, deg, bg).clone()
        out = rast.backward_raw(dev(f["vpixels"]), *t, cam, deg, bg)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"ref_scene_{mode}.npz"), image=img.cpu().numpy(), radii=rast.radii.cpu().numpy(),
                 n_rendered=int(rast.stats.n_rendered), ranges=rast.ranges.cpu().numpy(),
                 values_sorted=rast.values_sorted.cpu().numpy(), n_contrib=rast.n_contrib.cpu().numpy(),
                 accum_alpha=rast.accum_alpha.cpu().numpy(), vmeans=out[0].cpu().numpy(), vshs=out[1].cpu().numpy(),
                 vopacities=out[2].cpu().numpy().reshape(-1), vscales=out[3].cpu().numpy(), vrots=out[4].cpu().numpy(),
                 vmeans2d=rast.grad_means_2d.cpu().numpy())
        print(f"scene_{mode}: n_rendered = {int(rast.stats.n_rendered)} (oracle: {int(f['n_rendered'])})")
        rast.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
