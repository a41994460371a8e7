"""Where do the 0.9 ms at the edges of `extra_configs.rgbd`'s 10-step region come from (mean 1.62 vs median 1.54, while the
same workload as the headline reads 1.545 / 1.540)?  Runs bench.extra_configs with chosen spec lists in ONE process and prints
mean / median per entry.   python tools/experiments/extras_order_probe.py"""
import os, sys, types
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); os.chdir(R)
import torch
import bench, gsr_pkg
pkg = gsr_pkg.load()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
args = types.SimpleNamespace(extra_steps=10, warmup=5)
S = {n: (n, kw, what) for n, kw, what in bench.EXTRA_SPECS}
def run(names, tag):
    out = bench.extra_configs(pkg, dev, args, specs=[S[n] for n in names])
    for n in names:
        v = out
        for part in n.split("."): v = v[part]
        print(tag, n, v.get("ms_per_step"), v.get("ms_per_step_median"), v.get("error"))
run(["rgbd"], "alone:")
run(["rgbd", "rgbd"], "twice:")
run(["config2", "rgbd"], "after config2:")
run(["config5", "rgbd"], "after config5:")
run(["config5", "morton_order", "rgbd"], "after config5+morton:")
