#!/usr/bin/env python3
"""Execution counts of composite_bwd_kernel<3, 4, false>'s code paths on ONE view (default: config 3), measured from the
library's own lists on the GPU box (torch is only the calculator): how many instances the back-to-front walk stages,
how many survive the row-mask ballot, how many 16x4 pixel groups they visit, in how many of those at least one lane is
active (the `if (active)` block runs), how many instances reach the wave reduction, how many flush rows are written.
tools/isa_cost.py multiplies these by the VALU instruction counts of the matching basic blocks of the ISA listing and
compares the total with the PMC-counted SQ_INSTS_VALU (profiles/pmc_traffic.json).

  python tools/bwd_exec_counts.py [--gaussians N --width W --height H --seed S] > profiles/r03/bwd_exec_counts_cfg3.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsr_pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", dest="n", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--seed", type=int, default=1003)
ap.add_argument("--reference-lists", action="store_true")
args = ap.parse_args()

pkg = gsr_pkg.load()
dev = torch.device("cuda:0")
N, W, H, deg = args.n, args.width, args.height, 3
s = pkg.synthetic.make_scene(N, W, H, deg, args.seed)
cam = pkg.Camera(W, H, tuple(s.focal))
to = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)  # noqa: E731
p = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", device=dev, exact_tile_cull=not args.reference_lists)
rast.forward_raw(*p, cam, deg, (0.0, 0.0, 0.0))
torch.cuda.synchronize()
D = int(rast.stats.n_rendered)
gx, gy = rast.grid
T = gx * gy
ranges = rast.ranges.long()                       # (T,2)
ids = rast.values_sorted.long()                   # (D,)
masks = rast.instance_masks.long() & 0xFFFFF      # (D,) rows 0..15, quadrants 16..19
nc = rast.n_contrib.long()                        # (H,W)
g = rast.geometry()
m2, con, opa = g["means2d"], g["conics"], g["opacities"]

# per-instance tile / position
ln = ranges[:, 1] - ranges[:, 0]
tile = torch.repeat_interleave(torch.arange(T, device=dev), ln)
order = torch.argsort(ranges[:, 0] + (ln == 0) * (D + 1))  # instances are stored tile after tile in `ranges` order
start_of = ranges[:, 0]
# (tiles are contiguous slices [start, end): position = global index - start)
gidx = torch.arange(D, device=dev)
# map global index -> tile through a searchsorted on the sorted starts of non-empty tiles
nz = torch.nonzero(ln > 0).squeeze(1)
st_sorted, perm = torch.sort(start_of[nz])
tile = nz[perm][torch.searchsorted(st_sorted, gidx, right=True) - 1]
pos = gidx - start_of[tile]

# tile_last = deepest list position any pixel of the tile blended (n_contrib is 1-based position of the last contributor)
pad_h, pad_w = gy * 16, gx * 16
ncp = torch.zeros((pad_h, pad_w), dtype=torch.long, device=dev)
ncp[:H, :W] = nc
nct = ncp.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(T, 256)   # (T, 256) pixel (ly, lx)
tile_last = nct.max(1).values
walked = pos < tile_last[tile]                    # staged by the backward
batches = int(((tile_last + 63) // 64).sum())
zero_rows = int((ln - tile_last).clamp(min=0).sum())

rowgrp = torch.stack([((masks >> (4 * q)) & 0xF) != 0 for q in range(4)], 1)   # (D,4): footprint touches rows 4q..4q+3
cand = walked & ((masks & 0xFFFF) != 0)

# exact per-pixel test for the candidates, in chunks
X0 = (tile % gx) * 16
Y0 = (tile // gx) * 16
lx = torch.arange(16, device=dev).view(1, 1, 16).float()
ly = torch.arange(16, device=dev).view(1, 16, 1).float()
cidx = torch.nonzero(cand).squeeze(1)
grp_visits = int(rowgrp[cidx].sum())
grp_visits_q = [int(rowgrp[cidx][:, q].sum()) for q in range(4)]
act_groups_q = [0, 0, 0, 0]
act_groups = 0
reduced = 0
active_px = 0
evaluated_px = grp_visits * 64
B = 200_000
for b0 in range(0, cidx.numel(), B):
    ii = cidx[b0:b0 + B]
    gid = ids[ii]
    mx, my = m2[gid, 0].view(-1, 1, 1), m2[gid, 1].view(-1, 1, 1)
    a, b, c = con[gid, 0].view(-1, 1, 1), con[gid, 1].view(-1, 1, 1), con[gid, 2].view(-1, 1, 1)
    o = opa[gid].view(-1, 1, 1)
    dx = mx - (X0[ii].view(-1, 1, 1).float() + lx)
    dy = my - (Y0[ii].view(-1, 1, 1).float() + ly)
    sig = b * dx * dy + 0.5 * (a * dx * dx + c * dy * dy)
    al = torch.minimum(torch.tensor(0.99, device=dev), o * torch.exp(-sig))
    live = pos[ii].view(-1, 1, 1) < nct[tile[ii]].view(-1, 16, 16)
    act = live & (sig >= 0) & (al >= 1.0 / 255.0)                       # (n,16,16) [ly][lx]
    g4 = act.view(-1, 4, 4, 16).any(3).any(2)                            # rows 4q..4q+3
    g4 = g4 & rowgrp[ii]                                                 # only visited groups are evaluated
    act_groups += int(g4.sum())
    for q in range(4):
        act_groups_q[q] += int(g4[:, q].sum())
    reduced += int(g4.any(1).sum())
    active_px += int((act & rowgrp[ii].view(-1, 4, 1, 1).expand(-1, 4, 4, 16).reshape(-1, 16, 16)).sum())

out = {
    "config": {"n": N, "width": W, "height": H, "seed": args.seed, "tile_lists": "reference" if args.reference_lists else "exact cull"},
    "tiles": T, "tiles_nonempty": int((ln > 0).sum()), "instances_D": D,
    "instances_walked": int(walked.sum()), "batches_of_64": batches, "zero_rows_beyond_tile_last": zero_rows,
    "candidates_after_row_mask_ballot": int(cand.sum()),
    "group_visits_16x4": grp_visits, "group_visits_with_an_active_lane": act_groups,
    "group_visits_by_q": grp_visits_q, "active_group_visits_by_q": act_groups_q,
    "instances_reduced": reduced, "evaluated_lane_slots": evaluated_px, "active_lane_slots": active_px,
    "per_instance": {"group_visits": round(grp_visits / max(int(cand.sum()), 1), 4),
                     "active_group_visits": round(act_groups / max(int(cand.sum()), 1), 4),
                     "lane_efficiency": round(active_px / max(evaluated_px, 1), 4)},
}
print(json.dumps(out, indent=1))
