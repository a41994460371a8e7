"""The reference's own benchmark protocol on the hot path — a HARNESS, not a Trainer product (round-5 verdict, next #1).

The reference's only timing protocol is a training run (benchmark/pipeline.jl:19-39: 500 warm-up + 1000 timed `step!`, with
densification): every number of the earlier rounds was a STATIC scene.  This module drives exactly the calls `Trainer.step!`
makes on the path (src/training.jl:575-811), through the C ABI, over a multi-view batch of targets rendered from a hidden
ground-truth scene:

    update_lr! (utils.jl:75-83)  ->  SH-degree ramp (training.jl:584-586)  ->  shuffled view (:588-591)
    functor prologue (rasterizer.jl:200-253)         gsr_prologue_forward (only after the raw arrays were re-composed)
    rasterize (rasterizer.jl:255-408)                gsr_forward
    loss head (training.jl:656,684-694)              gsr_loss_l1_ssim
    ∇rasterize + NU.step! x 6 (training.jl:768-779)  gsr_backward_trainer_tail   (or gsr_backward + gsr_trainer_tail_step)
    post_train_step! (strategy.jl:78-105)            gsr_update_stats + densification.py (gsr_densify_* / gsr_compose_rows)

Used by tests/test_gpu_train_protocol.py (reduced size: oracle-chain parity of the first steps, PSNR, checkpoint -> resume bit
identity, the handle's view-history assertions), by bench.py's `train_protocol` section (full size, the reference's 500 + 1000
steps) and by tools/record_train_history.py.  numpy + torch + the package; NO oracle import (the oracle twin of the chain
lives in tests/train_oracle_chain.py).  The dataset side of the reference (COLMAP I/O, images) is out of scope: the "dataset" is a
procedural scene (synthetic.make_trained_like) seen from `n_views` poses."""
from __future__ import annotations

import math
import os
import sys
import time
from dataclasses import asdict, dataclass, field

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SH0 = 0.28209479177387814
GROUPS = ("points", "features_dc", "features_rest", "opacities", "scales", "rotations")  # training.jl:415-416


@dataclass
class Protocol:
    """Everything that defines a run (deterministic in these numbers).  Optimisation / strategy defaults are the reference's
    (`OptimizationParams`, src/utils.jl:2-12; `DefaultStrategy`, src/strategy.jl:28-66)."""
    width: int = 1920
    height: int = 1080
    mode: str = "rgbd"                # the reference's default training mode (rasterizer.jl:57-58)
    max_sh_degree: int = 3
    n_gt: int = 600_000               # Gaussians of the hidden ground-truth scene the targets are rendered from
    n_init: int = 200_000             # initial point cloud (the reference starts from COLMAP's sparse points)
    n_views: int = 32
    seed: int = 2024
    gt_sigma_px: float = 4.0
    # OptimizationParams
    lambda_dssim: float = 0.2
    lr_points_start: float = 16e-5
    lr_points_end: float = 16e-7
    lr_points_steps: int = 30_000
    lr_feature: float = 25e-4
    lr_opacities: float = 5e-2
    lr_scales: float = 5e-3
    lr_rotations: float = 1e-3
    # DefaultStrategy
    dense_percent: float = 0.01
    densify_from_iter: int = 500
    densify_until_iter: int = 15_000
    densification_interval: int = 100
    densify_grad_threshold: float = 2e-4
    opacity_reset_interval: int = 3_000
    min_opacity: float = 0.005
    sh_ramp_interval: int = 1000      # training.jl:584: `trainer.step % 1000 == 0 && gs.sh_degree < gs.max_sh_degree`
    fused_tail: bool = True           # gsr_backward_trainer_tail; False: gsr_backward + gsr_trainer_tail_step
    spatial_reorder: bool = False     # (not in the reference) Morton re-sort at the end of a densification round
    bins_budget_bytes: int = 0
    grad_precision: str = None        # None: the library default (∇scales / ∇rotations through the float64 chain); "fp32_reference":
                                      # the reference's own fp32 expression trees (what an oracle-chain comparison runs with: an
                                      # isotropic Gaussian's rotation gradient is then EXACTLY zero on both sides, where the float64
                                      # chain leaves 1e-17-level noise that NU.Adam's eps = 1e-15 turns into visible steps)
    init_jitter: float = 0.01
    init_color_noise: float = 0.05
    torch_pool_gb: int = 8            # device memory the harness takes from the driver ONCE, up front, and hands to torch's caching
                                      # allocator (1 GiB blocks): a densification round composes ~60 arrays of sizes no earlier round
                                      # had, and every one would otherwise be a hipMalloc inside a training step (the reference's
                                      # allocation cache plays the same role, strategy.jl:92).  0: off


# ------------------------------------------------------------------------------------------------------------------
# the "dataset": ground truth, poses, initial cloud — numpy, deterministic
# ------------------------------------------------------------------------------------------------------------------
def poses(p: Protocol):
    """n_views world->camera poses (row-major R, t) looking at the scene from a disc of camera centres around the origin
    (COLMAP convention: x right, y down, z forward) — a golden-angle spiral, so any prefix of the views is spread evenly."""
    out = []
    target = np.array([0.0, 0.2, 7.0])
    for j in range(p.n_views):
        u = (j + 0.5) / p.n_views
        phi = j * 2.399963229728653
        r = math.sqrt(u)
        c = np.array([1.4 * r * math.cos(phi), 0.45 * r * math.sin(phi), 0.5 * (u - 0.5)])
        z = target - c
        z /= np.linalg.norm(z)
        x = np.cross(np.array([0.0, 1.0, 0.0]), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z]).astype(np.float32)
        t = (-R.astype(np.float64) @ c).astype(np.float32)
        out.append((R, t))
    return out


def camera_extent(pose_list) -> float:
    """dataset.jl:166-170: 1.1 x the largest distance of a camera centre from their mean."""
    cs = np.stack([-(R.astype(np.float64).T @ t.astype(np.float64)) for R, t in pose_list])
    return float(np.float32(1.1 * np.linalg.norm(cs - cs.mean(0), axis=1).max()))


def ground_truth(pkg, p: Protocol):
    return pkg.synthetic.make_trained_like(p.n_gt, p.width, p.height, p.max_sh_degree, p.seed, sigma_px=p.gt_sigma_px,
                                           subclip_fraction=0.1)


def compute_scales(points: np.ndarray) -> np.ndarray:
    """dataset.jl:236-249: log of the root mean squared distance to the 3 nearest neighbours, the same on all three axes."""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(points).query(points, k=4)
    md = (d[:, 1:].astype(np.float64) ** 2).mean(1)
    s = np.log(np.sqrt(np.maximum(1e-7, md))).astype(np.float32)
    return np.ascontiguousarray(np.repeat(s[:, None], 3, 1))


def initial_model(p: Protocol, gt):
    """What the reference builds from a dataset's point cloud (gaussians.jl:22-58): positions + colours of a sparse subset of the
    scene's surface points (jittered, as a structure-from-motion cloud is), k-NN scales, identity rotations, opacity 0.1, SH
    bands above 0 zero.  Returns a dict of the six raw parameter arrays (numpy, Gaussian index first)."""
    rng = np.random.default_rng(p.seed + 17)
    f32 = np.float32
    solid = np.flatnonzero(gt.opacities_raw > 0.0)
    idx = rng.choice(solid, p.n_init, replace=solid.size < p.n_init)
    pts = gt.means[idx].astype(np.float64)
    pts += rng.normal(0.0, p.init_jitter, pts.shape) * np.maximum(pts[:, 2:3], 0.5)
    rgb = np.clip(SH0 * gt.shs[idx, 0, :].astype(np.float64) + 0.5 + rng.normal(0.0, p.init_color_noise, (p.n_init, 3)), 0.0, 1.0)
    K = (p.max_sh_degree + 1) ** 2
    pts = np.ascontiguousarray(pts, f32)
    rot = np.zeros((p.n_init, 4), f32)
    rot[:, 0] = 1.0
    return dict(points=pts, features_dc=np.ascontiguousarray(((rgb - 0.5) / SH0)[:, None, :], f32),
                features_rest=np.zeros((p.n_init, K - 1, 3), f32), scales=compute_scales(pts), rotations=rot,
                opacities=np.full((p.n_init, 1), math.log(0.1 / 0.9), f32))


def lr_points(p: Protocol, extent: float, step: int) -> float:
    """lr_exp_scheduler (utils.jl:75-83) with the extent-scaled end points of training.jl:241-244, in float32 as there."""
    f32 = np.float32
    a, b = f32(p.lr_points_start) * f32(extent), f32(p.lr_points_end) * f32(extent)
    t = np.clip(f32(step / p.lr_points_steps), f32(0), f32(1))
    return float(np.exp(np.log(a) * (f32(1) - t) + np.log(b) * t, dtype=np.float32))


def view_order(p: Protocol, epoch: int) -> np.ndarray:
    """`shuffle!(trainer.ids)` at the start of every pass over the views (training.jl:588-590) — here a pure function of
    (seed, epoch), so that a resumed run continues the same sequence."""
    return np.random.default_rng([p.seed, 7, epoch]).permutation(p.n_views)


def view_of_step(p: Protocol, step: int) -> int:
    k = step - 1
    return int(view_order(p, k // p.n_views)[k % p.n_views])


# ------------------------------------------------------------------------------------------------------------------
# the HIP side
# ------------------------------------------------------------------------------------------------------------------
class Harness:
    def __init__(self, pkg, p: Protocol, device="cuda:0", targets=None, init=None, verbose=False):
        import torch
        self.pkg, self.p, self.torch = pkg, p, torch
        self.dev = torch.device(device)
        self.verbose = verbose
        R, O, Dz = pkg.rasterizer, pkg.optim, pkg.densification
        self.R, self.O, self.Dz = R, O, Dz
        W, H = p.width, p.height
        self.poses = poses(p)
        self.extent = camera_extent(self.poses)
        gt = None
        if targets is None or init is None:
            gt = ground_truth(pkg, p)
        self.focal = (float(gt.focal[0]), float(gt.focal[1])) if gt is not None else None
        if self.focal is None:
            fx = 0.5 * W / math.tan(math.radians(30.0))
            self.focal = (float(np.float32(fx)), float(np.float32(fx)))
        self.cams = [pkg.Camera(W, H, self.focal, (0.5, 0.5), Rm, t) for Rm, t in self.poses]
        self.rast = R.GaussianRasterizer(W, H, mode=p.mode, device=self.dev, bins_budget_bytes=p.bins_budget_bytes,
                                         grad_precision=p.grad_precision)
        if p.torch_pool_gb and torch.cuda.memory_reserved(self.dev) < (p.torch_pool_gb << 30):
            pool = [torch.empty(1 << 30, dtype=torch.uint8, device=self.dev) for _ in range(p.torch_pool_gb)]
            del pool
        self.eval_rast = None    # evaluation renders take a handle of their own: the training handle's view history stays the run's
        self.bg = (0.0, 0.0, 0.0)
        to = self.to = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(self.dev)  # noqa: E731
        # targets: the ground truth rendered from every pose (forward-only, :rgb, full SH degree), (3,H,W) each
        if targets is None:
            targets = self.render_scene(gt)
        self.targets = targets
        ini = init if init is not None else initial_model(p, gt)
        self.gs = Dz.GaussianModel(*(to(ini[k]) for k in ("points", "features_dc", "features_rest", "scales", "rotations", "opacities")))
        lrs = dict(points=p.lr_points_start * self.extent, features_dc=p.lr_feature, features_rest=p.lr_feature / 20.0,
                   opacities=p.lr_opacities, scales=p.lr_scales, rotations=p.lr_rotations)          # training.jl:233-239
        self.opts = {k: O.Adam(getattr(self.gs, k), lrs[k], eps=1e-15) for k in GROUPS}
        self.strategy = Dz.DefaultStrategy(self.gs, dense_percent=p.dense_percent, densify_from_iter=p.densify_from_iter,
                                           densify_until_iter=p.densify_until_iter, densification_interval=p.densification_interval,
                                           densify_grad_threshold=p.densify_grad_threshold,
                                           opacity_reset_interval=p.opacity_reset_interval, min_opacity=p.min_opacity, seed=p.seed,
                                           spatial_reorder=p.spatial_reorder)
        self.step_no, self.sh_degree = 0, 0
        self.act = None          # (shs, opacities_act, scales_act) of the current raw arrays, or None: run the prologue
        self.losses = []         # device scalars, one per step
        self.history = []        # one record per forward: what the policies saw and did
        self.densify_log = []    # (step, n_before, n_after, host_ms)
        self.last_split_seed = None

    # -- rendering helpers ------------------------------------------------------------------------------------------
    def render_scene(self, scene):
        """Forward-only :rgb renders of a synthetic.Scene from every pose -> list of (3,H,W) device tensors."""
        torch, R, p = self.torch, self.R, self.p
        r = R.GaussianRasterizer(p.width, p.height, mode="rgb", device=self.dev)
        t = [self.to(scene.means), self.to(scene.shs), self.to(scene.opacities.reshape(-1, 1)), self.to(scene.scales), self.to(scene.rotations)]
        out = []
        for cam in self.cams:
            img = r.forward_raw(*t, cam, scene.sh_degree, self.bg, forward_only=True)
            out.append(img.permute(2, 0, 1).contiguous())
        torch.cuda.synchronize()
        r.close()
        return out

    def raw(self):
        return {k: getattr(self.gs, k) for k in GROUPS}

    def prologue(self):
        gs = self.gs
        rest = gs.features_rest if gs.features_rest.numel() else None
        self.act = list(self.R.prologue_forward(gs.features_dc, rest, gs.opacities, gs.scales))

    def psnr(self, views=None):
        """Mean PSNR of the model's forward-only renders against the targets (training views: the reference's benchmark trains
        on every view, `holdout=0`, benchmark/pipeline.jl:7)."""
        torch = self.torch
        if self.act is None:
            self.prologue()
        if self.eval_rast is None:
            self.eval_rast = self.R.GaussianRasterizer(self.p.width, self.p.height, mode=self.p.mode, device=self.dev)
        vals = []
        for v in (range(self.p.n_views) if views is None else views):
            img = self.eval_rast.forward_raw(self.gs.points, *self.act, self.gs.rotations, self.cams[v], self.sh_degree, self.bg,
                                        forward_only=True)
            mse = ((img[:, :, :3].permute(2, 0, 1) - self.targets[v]) ** 2).mean()
            vals.append(-10.0 * torch.log10(mse.clamp_min(1e-12)))
        return float(torch.stack(vals).mean())

    # -- one training step ------------------------------------------------------------------------------------------
    def step(self):
        p, gs, rast, O = self.p, self.gs, self.rast, self.O
        self.step_no += 1
        step = self.step_no
        self.opts["points"].lr = lr_points(p, self.extent, step)                       # update_lr!
        if step % p.sh_ramp_interval == 0 and self.sh_degree < p.max_sh_degree:        # training.jl:584-586
            self.sh_degree += 1
        v = view_of_step(p, step)
        cam = self.cams[v]
        if self.act is None:
            self.prologue()
        shs, oa, sa = self.act
        img = rast.forward_raw(gs.points, shs, oa, sa, gs.rotations, cam, self.sh_degree, self.bg)
        st = rast.stats
        self.history.append(dict(step=step, view=v, n=len(gs), n_rendered=int(st.n_rendered), n_visible=int(st.n_visible),
                                 max_tile=int(st.max_tile_instances), binning=int(st.compact_binning), form=int(st.preprocess_form),
                                 bin_capacity=int(st.bin_capacity), tiers=[int(x) for x in st.tier_tiles], sh_degree=self.sh_degree,
                                 **st.history()))
        loss, vp = self.pkg.fused_ssim.l1_ssim_loss(rast, img, self.targets[v], p.lambda_dssim)
        self.losses.append(loss)
        color = p.mode != "rgb"   # the loss head's own cotangent: zeros above the colour channels (training.jl:656,684-685)
        if p.fused_tail:
            O.fused_backward_tail_step(rast, vp, self.opts, self.raw(), shs, oa, sa, cam, self.sh_degree, self.bg,
                                       forward_generation=int(st.generation), color_cotangent=color)
        else:
            vm, vsh, vo, vsc, vr, _, _ = rast.backward_raw(vp, gs.points, shs, oa, sa, gs.rotations, cam, self.sh_degree, self.bg,
                                                           forward_generation=int(st.generation), color_cotangent=color)
            O.trainer_tail_step(self.opts, self.raw(), dict(vmeans=vm, vshs=vsh, vopacities=vo, vscales=vsc, vrot=vr), shs, oa, sa)
        self.post_train_step(step)
        return v

    def post_train_step(self, step):
        """post_train_step! (strategy.jl:78-105) — densification.post_train_step with the split seed drawn HERE, so that an
        oracle twin of the chain can be handed the same seed."""
        s, Dz, torch = self.strategy, self.Dz, self.torch
        if step > s.densify_until_iter:
            return
        will_densify = step >= s.densify_from_iter and step % s.densification_interval == 0
        seed, t0, n0 = None, None, len(self.gs)
        if will_densify:
            seed = s.next_split_seed()
            torch.cuda.synchronize()     # (so that host_ms below is the densification, not the queue in front of it)
            t0 = time.perf_counter()
        self.last_split_seed = seed
        densified, reset = Dz.post_train_step(s, self.gs, self.opts, self.rast, step, self.extent, seed=seed)
        if densified or reset:
            # the raw arrays were re-composed / the logits reset: the activated copies are stale.  Re-run the prologue HERE — inside
            # the step that re-composed them, which is a slow one anyway — so that the step after it is a plain step
            self.prologue()
        if densified:
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
            self.densify_log.append(dict(step=step, n_before=n0, n_after=len(self.gs), host_ms=round(ms, 3)))
            if self.verbose:
                print(f"  step {step}: densified {n0} -> {len(self.gs)} Gaussians in {ms:.1f} ms", file=sys.stderr)

    def run(self, steps):
        for _ in range(steps):
            self.step()

    def loss_values(self):
        torch = self.torch
        return torch.stack(self.losses).cpu().numpy() if self.losses else np.zeros(0, np.float32)

    def nonfinite(self) -> int:
        torch = self.torch
        bad = sum(int((~torch.isfinite(getattr(self.gs, k))).sum()) for k in GROUPS if getattr(self.gs, k).numel())
        return bad + int((~torch.isfinite(torch.stack(self.losses))).sum() if self.losses else 0)

    # -- checkpoint / resume ----------------------------------------------------------------------------------------
    def host_model(self):
        ply = self.pkg.ply
        g = self.gs
        c = lambda t: t.detach().cpu().numpy()  # noqa: E731
        return ply.GaussianModel(c(g.points), c(g.features_dc), c(g.features_rest), c(g.scales), c(g.rotations), c(g.opacities),
                                 self.sh_degree, self.p.max_sh_degree)

    def save(self, path):
        """The reference's checkpoint (training.jl:418-445: gaussians + the six NU.Adam + step) + what a BIT-IDENTICAL resume also
        needs and the reference does not save: the strategy's running statistics and its split-noise position."""
        self.torch.cuda.synchronize()
        self.pkg.checkpoint.save_state(path, self.host_model(), self.opts, self.step_no, strategy=self.strategy,
                                       extra_meta={"harness.sh_degree": str(self.sh_degree)})

    @classmethod
    def resume(cls, pkg, p: Protocol, path, device="cuda:0", targets=None):
        """A fresh harness (fresh handle: empty view history) continued from a checkpoint."""
        import torch
        ck = pkg.checkpoint.load_checkpoint(path)
        g = pkg.checkpoint.read_gaussians(ck, "gaussians")
        init = dict(points=g.points, features_dc=g.features_dc, features_rest=g.features_rest, scales=g.scales, rotations=g.rotations,
                    opacities=g.opacities)
        h = cls(pkg, p, device, targets=targets, init=init)
        _, step = pkg.checkpoint.load_state(path, h.opts, strategy=h.strategy)
        for k in GROUPS:   # lr is configuration, not state: the schedule sets the points' lr every step
            assert h.opts[k].mu.numel() == getattr(h.gs, k).numel()
        h.step_no = int(step)
        h.sh_degree = int(ck.read_scalar("harness.sh_degree"))
        torch.cuda.synchronize()
        return h

    def export_ply(self, path):
        self.pkg.ply.export_ply(self.host_model(), path)

    def close(self):
        self.rast.close()
        if self.eval_rast is not None:
            self.eval_rast.close()


def summarize_steps(ms):
    """mean / median / p99 / max of a list of per-step milliseconds."""
    a = np.sort(np.asarray(ms, np.float64))
    if a.size == 0:
        return {}
    return {"mean": round(float(a.mean()), 4), "median": round(float(a[a.size // 2]), 4),
            "p99": round(float(a[min(a.size - 1, int(math.ceil(0.99 * a.size)) - 1)]), 4), "max": round(float(a[-1]), 4),
            "p99_over_mean": round(float(a[min(a.size - 1, int(math.ceil(0.99 * a.size)) - 1)] / a.mean()), 3)}


def protocol_run(pkg, p: Protocol, warmup=500, steps=1000, device="cuda:0", survey_steps=30, ply_out=None, verbose=False):
    """benchmark/pipeline.jl:19-39 on the path: `warmup` untimed steps, `steps` timed ones (one HIP event per step on the launch
    stream: per-step intervals; and the wall clock around the region), then a short per-stage survey.  Returns the record
    bench.py's `train_protocol` section prints."""
    import torch
    h = Harness(pkg, p, device, verbose=verbose)
    n0 = len(h.gs)
    psnr0 = h.psnr(range(min(8, p.n_views)))
    t_w = time.perf_counter()
    h.run(warmup)
    torch.cuda.synchronize()
    warm_s = time.perf_counter() - t_w
    n_warm = len(h.gs)
    marks = []
    t0 = time.perf_counter()
    for _ in range(steps):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        h.step()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append(e)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    first = warmup + 1
    dens_steps = {d["step"] for d in h.densify_log}
    is_d = np.array([(first + i) in dens_steps for i in range(steps)])
    after_d = np.array([(first + i - 1) in dens_steps for i in range(steps)])
    ps = np.asarray(per_step)
    # Hiccups, separated from growth: the scene grows several-fold over the timed region (so late steps are legitimately slower
    # than the run's mean); each step is therefore also priced against the MEDIAN PLAIN STEP OF ITS OWN densification round
    round_of = np.array([(first + i - 1) // p.densification_interval for i in range(steps)])
    ratio = np.ones(steps)
    for r_ in np.unique(round_of):
        m_ = round_of == r_
        plain = ps[m_ & ~is_d & ~after_d]
        if plain.size:
            ratio[m_] = ps[m_] / np.median(plain)
    hist_t = [r for r in h.history if r["step"] >= first]
    stats_end = dict(h.history[-1])
    # stage survey AFTER the timed region (every stage with its event pair costs ~3 % of a step: not inside it)
    stages = {}
    if survey_steps > 0:
        h.rast.profile(True)
        s0 = h.step_no
        while h.step_no < s0 + survey_steps and (h.step_no + 1) % p.densification_interval != 0:
            h.step()
        torch.cuda.synchronize()
        n_sv = h.step_no - s0
        stages = {k: round(ms / max(n_sv, 1), 4) for k, (ms, c) in h.rast.profile_read().items() if c > 0}
        h.rast.profile(False)
    psnr1 = h.psnr(range(min(8, p.n_views)))
    losses = h.loss_values()
    hist = h.history
    rec = {
        "protocol": "benchmark/pipeline.jl:19-39 on the path: warm-up + timed training steps (update_lr!, SH ramp, shuffled views, "
                    "prologue, gsr_forward, gsr_loss_l1_ssim, gsr_backward_trainer_tail, post_train_step! with densification)",
        "warmup_steps": warmup, "timed_steps": steps, "mode": p.mode, "resolution": [p.width, p.height], "views": p.n_views,
        "gaussians": {"ground_truth": p.n_gt, "initial": n0, "after_warmup": n_warm, "final": len(h.gs)},
        "ms_per_step": {"wall_mean": round(1e3 * wall / steps, 4), **summarize_steps(per_step),
                        "plain_steps": summarize_steps(ps[~is_d & ~after_d]),
                        "densification_steps": summarize_steps(ps[is_d]),
                        "first_step_after_a_densification": summarize_steps(ps[after_d]),
                        "over_own_round_median": {
                            "all_steps": {"p99": round(float(np.sort(ratio)[min(steps - 1, int(math.ceil(0.99 * steps)) - 1)]), 3),
                                          "max": round(float(ratio.max()), 3)},
                            "plain_steps": {"p99": round(float(np.quantile(ratio[~is_d & ~after_d], 0.99)), 3),
                                            "max": round(float(ratio[~is_d & ~after_d].max()), 3)},
                            "first_step_after_a_densification": {"mean": round(float(ratio[after_d].mean()), 3) if after_d.any() else None,
                                                                 "max": round(float(ratio[after_d].max()), 3) if after_d.any() else None},
                            "densification_steps": {"mean": round(float(ratio[is_d].mean()), 3) if is_d.any() else None,
                                                    "max": round(float(ratio[is_d].max()), 3) if is_d.any() else None},
                            "note": "each step's time over the median plain step of ITS OWN densification round: the scene grows "
                                    "several-fold during the timed region, so p99 / mean of the raw times measures growth, this "
                                    "measures hiccups"}},
        "per_round_median_ms": [round(float(np.median(ps[(round_of == r_) & ~is_d & ~after_d])), 4) for r_ in np.unique(round_of)
                                if ((round_of == r_) & ~is_d & ~after_d).any()],
        "warmup_wall_s": round(warm_s, 2), "timed_wall_s": round(wall, 2),
        "steps_per_s": round(steps / wall, 1),
        "Mpixels_per_s": round(p.width * p.height * steps / wall / 1e6, 1),
        "densification": {"rounds_total": len(h.densify_log), "rounds_in_timed_region": int(is_d.sum()),
                          "host_ms_per_round": summarize_steps([d["host_ms"] for d in h.densify_log]),
                          "log": h.densify_log},
        "final_view": {k: stats_end[k] for k in ("n", "n_rendered", "n_visible", "max_tile", "binning", "form", "bin_capacity", "sh_degree")},
        "max_tile_instances_over_run": int(max(r["max_tile"] for r in hist)),
        "history": {k: int(hist[-1][k]) for k in ("bins_regrowths", "compact_fallbacks", "tuner_rearms", "scratch_regrowths",
                                                  "fused_relaunches", "held_views")},
        "binning_modes_seen": sorted({r["binning"] for r in hist_t}), "forms_seen": sorted({r["form"] for r in hist_t}),
        "stages_ms_survey": stages, "survey_note": f"{survey_steps} further steps after the timed region with every stage timed",
        "loss": {"first": round(float(losses[:10].mean()), 5), "at_warmup_end": round(float(losses[max(0, warmup - 10):warmup].mean()), 5),
                 "last": round(float(losses[-10:].mean()), 5)},
        "psnr_db": {"initial": round(psnr0, 2), "final": round(psnr1, 2), "views": min(8, p.n_views)},
        "nonfinite": h.nonfinite(),
        "camera_extent": round(h.extent, 4),
        "reference_context": "the reference's only published step figure is 6.7 ms per step at unstated hardware and scene "
                             "(src/GaussianSplatting.jl:97-98): context, not a target",
    }
    if ply_out:
        h.export_ply(ply_out)
        rec["ply"] = os.path.basename(ply_out)
    h.close()
    return rec, h


if __name__ == "__main__":
    import argparse
    import json

    import gsr_pkg
    ap = argparse.ArgumentParser()
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--n-init", type=int, default=200_000)
    ap.add_argument("--n-gt", type=int, default=600_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--views", type=int, default=32)
    ap.add_argument("--mode", default="rgbd")
    ap.add_argument("--grad-threshold", type=float, default=2e-4)
    ap.add_argument("--interval", type=int, default=100)
    ap.add_argument("--from-iter", type=int, default=500)
    ap.add_argument("--ply", default=None)
    ap.add_argument("--history-out", default=None, help="write the per-view history (tests/golden/train_history.json format)")
    a = ap.parse_args()
    pr = Protocol(width=a.width, height=a.height, mode=a.mode, n_gt=a.n_gt, n_init=a.n_init, n_views=a.views,
                  densify_grad_threshold=a.grad_threshold, densification_interval=a.interval, densify_from_iter=a.from_iter)
    rec, hh = protocol_run(gsr_pkg.load(), pr, a.warmup, a.steps, ply_out=a.ply, verbose=True)
    print(json.dumps(rec))
    if a.history_out:
        json.dump({"width": pr.width, "height": pr.height, "protocol": asdict(pr),
                   "densify_steps": [d["step"] for d in hh.densify_log], "views": hh.history}, open(a.history_out, "w"))
