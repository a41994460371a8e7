#!/usr/bin/env python3
"""On the GPU box: the launch-to-launch interval of every step of bench.py's timed region, in order (is the mean above the
median because of the first steps, or of scattered ones?).   python tools/experiments/step_times.py [steps = 40] [warmup = 3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import gsr_pkg  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sys.argv = [sys.argv[0]]
args = bench.parse_args([])
pkg = gsr_pkg.load()
for rep in range(3):
    wl = bench.Workload(pkg, torch.device("cuda", 0), 0, 1, n=args.n, width=args.width, height=args.height,
                        sh_degree=args.sh_degree, seed=args.seed)
    rast = wl.rast
    for _ in range(warm):
        wl.step()
    wl.sync()
    rast.profile(True)
    for _ in range(5):
        wl.step()
    wl.sync()
    rast.profile_read()
    rast.profile(True, stages=["composite_bwd"])
    for _ in range(steps):
        wl.step()
    wl.sync()
    iv = rast.profile_intervals("composite_bwd")
    rast.profile(False)
    print("run", rep, " ".join(f"{x:.3f}" for x in iv))
    print("   mean %.4f median %.4f" % (sum(iv) / len(iv), sorted(iv)[len(iv) // 2]))
    wl.close()
