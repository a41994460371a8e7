"""Round 6: how fast does the training harness grow the model?  (choosing the reduced-size test's and the full-size protocol's
densification threshold / initial cloud so that N passes 3x / 5x: the round-5 verdict asks for 50 k -> 150 k and 200 k -> > 1 M)"""
import json
import sys
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gsr_pkg
import train_harness as TH
import torch

pkg = gsr_pkg.load()
which = sys.argv[1]
if which == "small":
    for n_init, n_gt, thr, sig in ((50_000, 150_000, 2.5e-5, 4.0), (50_000, 150_000, 2.5e-5, 6.0), (50_000, 200_000, 1.5e-5, 4.0)):
        p = TH.Protocol(width=480, height=272, n_gt=n_gt, n_init=n_init, n_views=16, densify_from_iter=100, densification_interval=50,
                        sh_ramp_interval=100, seed=2024, densify_grad_threshold=thr, gt_sigma_px=sig)
        h = TH.Harness(pkg, p)
        p0 = h.psnr()
        t0 = time.time()
        h.run(300)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(json.dumps(dict(n_init=n_init, n_gt=n_gt, thr=thr, sigma=sig, psnr=(round(p0, 2), round(h.psnr(), 2)), s=round(dt, 2),
                              n=[d["n_after"] for d in h.densify_log], hist=h.history[-1])), flush=True)
        h.close()
else:
    for n_init, n_gt, thr in ((200_000, 1_500_000, 4e-5),):
        p = TH.Protocol(n_gt=n_gt, n_init=n_init, densify_grad_threshold=thr)
        t0 = time.time()
        rec, h = TH.protocol_run(pkg, p, 500, 1000, verbose=True)
        rec["densification"].pop("log")
        print(json.dumps(dict(n_init=n_init, n_gt=n_gt, thr=thr, total_s=round(time.time() - t0, 1), rec=rec)), flush=True)
