"""The ORACLE twin of tools/train_harness.py's training step — test infrastructure (imports oracle/): the same chain
`Trainer.step!` runs on the path (src/training.jl:575-811), statement by statement on the CPU restatement:

    update_lr!, SH ramp, shuffled view  ->  orc.prologue_forward  ->  orc.forward  ->  orc.loss_head  ->  orc.backward
    ->  orc.prologue_backward  ->  orc.adam_step x 6  ->  oracle/densify.py post_train_step

It shares the protocol's pure definitions (schedule, view order, poses, lr) with the harness; nothing else."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import train_harness as TH  # noqa: E402

from oracle import densify as dz  # noqa: E402


class OracleChain:
    def __init__(self, orc, p: TH.Protocol, init: dict, targets, focal):
        self.orc, self.p = orc, p
        self.poses = TH.poses(p)
        self.extent = TH.camera_extent(self.poses)
        self.cams = [orc.Camera(p.width, p.height, focal, (0.5, 0.5), R, t) for R, t in self.poses]
        self.targets = targets   # list of (3,H,W) numpy
        self.gs = dz.Model(*(np.ascontiguousarray(init[k]).copy() for k in dz.PARAMS))
        self.opts = dz.new_optimizers(self.gs)
        self.lrs = dict(points=p.lr_points_start * self.extent, features_dc=p.lr_feature, features_rest=p.lr_feature / 20.0,
                        opacities=p.lr_opacities, scales=p.lr_scales, rotations=p.lr_rotations)
        self.strategy = dz.Strategy.for_model(len(self.gs), dense_percent=p.dense_percent, densify_from_iter=p.densify_from_iter,
                                              densify_until_iter=p.densify_until_iter, densification_interval=p.densification_interval,
                                              densify_grad_threshold=p.densify_grad_threshold,
                                              opacity_reset_interval=p.opacity_reset_interval, min_opacity=p.min_opacity)
        self.step_no, self.sh_degree = 0, 0
        self.last = None

    def step(self, split_seed):
        orc, p, gs = self.orc, self.p, self.gs
        self.step_no += 1
        step = self.step_no
        self.lrs["points"] = TH.lr_points(p, self.extent, step)
        if step % p.sh_ramp_interval == 0 and self.sh_degree < p.max_sh_degree:
            self.sh_degree += 1
        v = TH.view_of_step(p, step)
        cam = self.cams[v]
        rest = gs.features_rest if gs.features_rest.size else None
        shs, oa, sa = orc.prologue_forward(gs.features_dc, rest, gs.opacities, gs.scales)
        st = orc.forward(gs.points, shs, oa, sa, gs.rotations, cam, self.sh_degree, mode=p.mode)
        loss, vp = orc.loss_head(st.image, self.targets[v], np.float32(p.lambda_dssim))
        g = orc.backward(st, vp, gs.points, shs, oa, sa, gs.rotations, cam, self.sh_degree)
        vdc, vrest, vo, vs = orc.prologue_backward(oa, sa, g.vshs, g.vopacities.reshape(-1, 1), g.vscales, gs.scales.shape[1])
        grads = dict(points=g.vmeans, features_dc=vdc, features_rest=vrest, opacities=vo, scales=vs, rotations=g.vrots)
        for k in dz.PARAMS:
            theta = getattr(gs, k).reshape(-1)
            if theta.size == 0:
                continue
            self.opts[k]["step"] += 1
            orc.adam_step(theta, np.ascontiguousarray(grads[k]).reshape(-1), self.opts[k]["mu"], self.opts[k]["nu"],
                          self.opts[k]["step"], self.lrs[k], 0.9, 0.999, 1e-15)
        self.last = dict(view=v, loss=float(loss), n_rendered=int(st.n_rendered), state=st)
        return dz.post_train_step(self.strategy, gs, self.opts, st.radii, g.vmeans2d, (p.width, p.height), step, self.extent,
                                  seed=0 if split_seed is None else split_seed)
