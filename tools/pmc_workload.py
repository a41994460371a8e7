#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (HBM traffic per kernel launch):
a calibration stream (float4 copy of a known byte count, far larger than the 256 MiB
Infinity Cache) followed by a few steps of the bench workload.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o fetch -- python3 tools/pmc_workload.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -o write -- python3 tools/pmc_workload.py
then tools/pmc_parse.py OUT -> profiles/pmc_traffic.json
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsr_pkg  # noqa: E402

pkg = gsr_pkg.load()
N = int(os.environ.get("GSR_PMC_N", 1_000_000))
W, H, deg, seed = 1920, 1080, 3, 1003
dev = torch.device("cuda:0")
# calibration: 1 GiB read + 1 GiB written by one elementwise copy kernel, 3 times
src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()

s = pkg.synthetic.make_scene(N, W, H, deg, seed)
cam = pkg.Camera(W, H, tuple(s.focal))
to = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)  # noqa: E731
params = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
target = to(pkg.synthetic.make_target(W, H, seed))
rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", device=dev, exact_tile_cull=True)  # as bench.py
arena = torch.empty(pkg.distributed.arena_numel(N, 16), device=dev)
for _ in range(4):
    img = rast.forward_raw(*params, cam, deg, (0.0, 0.0, 0.0))
    _, vp = pkg.fused_ssim.l1_ssim_loss(rast, img, target)
    rast.backward_raw(vp, *params, cam, deg, (0.0, 0.0, 0.0), arena=arena)
torch.cuda.synchronize()
print("pmc workload done: D =", rast.stats.n_rendered)
