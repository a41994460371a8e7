#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (HBM traffic and SQ counters per kernel launch):
a calibration stream (float4 copy of a known byte count, far larger than the 256 MiB
Infinity Cache) followed by a few steps of ONE bench.py configuration, named by the same
flags bench.py takes; the configuration key and the instance count are written next to
the counter files so tools/pmc_parse.py can file the measurement under the right key.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o fetch -- python3 tools/pmc_workload.py [flags]
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -o write -- python3 tools/pmc_workload.py [flags]
  rocprofv3 --pmc SQ_INSTS_VALU ... --kernel-trace --output-format csv -d OUT -o sq_pass1 -- python3 tools/pmc_workload.py [flags]
then  tools/pmc_parse.py OUT profiles/pmc_traffic.json  (merges under configs[key])
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402  (config_key)
import gsr_pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", dest="n", type=int, default=int(os.environ.get("GSR_PMC_N", 1_000_000)))
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--sh-degree", type=int, default=3)
ap.add_argument("--seed", type=int, default=1003)
ap.add_argument("--mode", default="rgb")
ap.add_argument("--scene", default="uniform", choices=["uniform", "trained"], help="synthetic.scene_by_name (bench.py --scene)")
ap.add_argument("--no-loss", action="store_true")
ap.add_argument("--reference-lists", action="store_true")
ap.add_argument("--forward-only", action="store_true", help="GSR_FORWARD_ONLY renders (a step = one forward)")
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--meta", default=os.path.join(ROOT, "gpurun_out", "pmc_meta.json"))
args = ap.parse_args()

pkg = gsr_pkg.load()
N, W, H, deg, seed = args.n, args.width, args.height, args.sh_degree, args.seed
dev = torch.device("cuda:0")
# calibration: 1 GiB read + 1 GiB written by one elementwise copy kernel, 3 times
src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
del src, dst

s = pkg.synthetic.scene_by_name(args.scene, N, W, H, deg, seed)
cam = pkg.Camera(W, H, tuple(s.focal))
to = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)  # noqa: E731
params = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
target = to(pkg.synthetic.make_target(W, H, seed))
C = pkg.rasterizer.n_color_features(args.mode)
vpf = to(pkg.synthetic.make_vpixels(W, H, C, seed))
rast = pkg.rasterizer.GaussianRasterizer(W, H, mode=args.mode, device=dev, exact_tile_cull=not args.reference_lists)
arena = torch.empty(pkg.distributed.arena_numel(N, s.shs.shape[1]), device=dev)
for _ in range(args.steps):
    if args.forward_only:
        rast.forward_raw(*params, cam, deg, (0.0, 0.0, 0.0), forward_only=True)
        continue
    img = rast.forward_raw(*params, cam, deg, (0.0, 0.0, 0.0))
    vp = vpf if args.no_loss else pkg.fused_ssim.l1_ssim_loss(rast, img, target)[1]
    # (as bench.py's step: the loss head's cotangent is announced as colour-only in :rgbd / :rgbdn)
    rast.backward_raw(vp, *params, cam, deg, (0.0, 0.0, 0.0), arena=arena, color_cotangent=not args.no_loss)
torch.cuda.synchronize()
key = bench.config_key(N, W, H, deg, args.mode, not args.reference_lists, not args.no_loss, scene=args.scene)
if args.forward_only:
    key = key.rsplit("_", 1)[0] + "_fwdonly"
meta = {"key": key, "tile_instances": int(rast.stats.n_rendered), "n_visible": int(rast.stats.n_visible), "steps": args.steps}
os.makedirs(os.path.dirname(args.meta), exist_ok=True)
json.dump(meta, open(args.meta, "w"))
print("pmc workload done:", json.dumps(meta))
