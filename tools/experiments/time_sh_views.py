"""Time gsr_sh_grad_from_views at config 3's size for V views:  python tools/experiments/time_sh_views.py [V]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import gsr_pkg
pkg = gsr_pkg.load()
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, K, deg = 1_000_000, 16, 3
dev = torch.device("cuda:0")
means = torch.randn(N, 3, device=dev)
vc = torch.randn(V, N, 3, device=dev)
centers = torch.randn(V, 3, device=dev) * 5
out = torch.empty(N, K, 3, device=dev)
for _ in range(3):
    pkg.rasterizer.sh_grad_from_views(means, vc, centers, K, deg, out=out)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    pkg.rasterizer.sh_grad_from_views(means, vc, centers, K, deg, out=out)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(f"V={V}: {ms:.4f} ms per call, {(V * N * 12 + N * 12 + N * K * 12) / ms * 1e-6:.0f} GB/s")
