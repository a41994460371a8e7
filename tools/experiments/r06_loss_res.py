"""Round 6 (round-5 verdict, next #5): loss_bwd at 2560x1440 took 1.38 x its pixel-scaled config-3 cost while loss_fwd scaled exactly.
Times gsr_loss_l1_ssim's two launches (HIP-event stage timing) over a set of resolutions and modes: is it the width (row stride),
the tile count (rounds of workgroups), or the mode?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gsr_pkg
import torch

pkg = gsr_pkg.load()
res = [(1920, 1080), (2560, 1440), (2560, 1424), (2560, 1456), (2544, 1440), (2576, 1440), (2304, 1600), (3840, 2160), (3856, 2160), (1280, 720)]
print("W x H  mode  tiles  loss_fwd_ms  loss_bwd_ms  fwd_ns/px  bwd_ns/px")
for mode in ("rgb", "rgbd"):
    C = 3 if mode == "rgb" else 5
    for W, H in res:
        rast = pkg.rasterizer.GaussianRasterizer(W, H, mode=mode)
        img = torch.rand(H, W, C, device="cuda")
        tgt = torch.rand(3, H, W, device="cuda")
        for _ in range(20):
            pkg.fused_ssim.l1_ssim_loss(rast, img, tgt)
        torch.cuda.synchronize()
        rast.profile(True)
        n = 100
        for _ in range(n):
            pkg.fused_ssim.l1_ssim_loss(rast, img, tgt)
        torch.cuda.synchronize()
        pr = rast.profile_read()
        f, b = pr["loss_fwd"][0] / n, pr["loss_bwd"][0] / n
        T = ((W + 15) // 16) * ((H + 15) // 16)
        print(f"{W}x{H} {mode} {T} {f:.4f} {b:.4f} {1e6 * f / (W * H):.3f} {1e6 * b / (W * H):.3f}", flush=True)
        rast.close()
