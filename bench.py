#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json): fwd + loss + bwd Mpixels/s at
1920x1080 on 1 M synthetic Gaussians, SH degree 3 (config 3), one view per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A step = gsr_forward (project + SH + binning + per-tile sort + composite) + the L1/SSIM
loss head and its pullback + gsr_backward (composite backward + per-Gaussian backward),
with all inputs already resident in HBM, plus — for N > 1 — the single RCCL all-reduce
of the 59·N-float gradient arena.  Rank r renders view r of the batch (weak scaling:
per-GPU work is fixed, `value` counts the pixels of all views).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     — dominant kernel: algorithmic HBM bytes per launch / mean launch time
                 (HIP events on the launch stream, gsr_profile_*), against 8 TB/s; `traffic` =
                 PMC-counted HBM bytes per launch of that kernel, ONLY when profiles/pmc_traffic.json
                 holds a measurement of exactly this configuration (else null); `valu` = the VALU-issue
                 roofline of the same kernel (SQ_INSTS_VALU x 2 cycles / 1024 SIMDs, the bound that
                 actually binds the compositing kernels), from the same file;
  cpu_baseline — the oracle (C restatement of the reference algorithm, OpenMP) timed on
                 this host's cores on the same workload (rank 0, N = 1 only).
`value` / `ms_per_step` follow the driver contract (K steps between two barriers+synchronize, total / K);
`ms_per_step_median` is the median of the K-1 launch-to-launch intervals of the dominant stage's HIP events
inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import gsr_pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD, CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 2.0  # 256 CUs x 4 SIMD-32; a wave64 VALU op issues over 2 cycles


def config_key(N, W, H, deg, mode, exact_cull, loss):
    """Key of a measured configuration in profiles/pmc_traffic.json (tools/pmc_workload.py writes the same)."""
    return f"N{N}_{W}x{H}_SH{deg}_{mode}_{'cull' if exact_cull else 'reflists'}_{'loss' if loss else 'noloss'}"


def algorithmic_bytes(stage, N, V, D, P, T, C=3, K=16):
    """SURVEY.md §8(d) per-stage algorithmic HBM bytes (one launch = one view)."""
    return {
        # preprocess + emit (SURVEY.md §8d rows "preprocess" and "emit": one fused kernel here)
        "preprocess": 40 * N + 8 * N + 12 * K * V + 39 * V + 8 * N + 12 * V + 12 * D,
        "tile_scan": 8 * T,
        "scatter": 0,  # (stage of earlier builds; now part of preprocess)
        "tile_sort": 12 * D + 12 * D + 8 * D + 8 * T,
        "composite_fwd": (28 + 4 * C) * D + 8 * T + (4 * C + 8) * P,
        # the fused launch does the work of both stages (the stream is still written once and read once)
        "sort_composite_fwd": 12 * D + 12 * D + 8 * D + 8 * T + (28 + 4 * C) * D + 8 * T + (4 * C + 8) * P,
        "composite_bwd": (4 * C + 8) * P + 8 * T + (28 + 4 * C) * D + 4 * (C + 6) * D,
        "pergauss_bwd": 4 * N + (87 + 12 * K) * V + (44 + 12 * K) * N,
        "zero_acc": 48,  # pose-gradient accumulators only (when requested)
        "loss_fwd": 72 * P,
        "loss_bwd": 84 * P + 24 * P,
    }[stage]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", dest="n", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1003)
    ap.add_argument("--no-loss", action="store_true", help="config 2 style: random cotangent instead of L1/SSIM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--views", type=int, default=8, help="size of the multi-view batch the poses are drawn from")
    ap.add_argument("--with-optimizer", action="store_true",
                    help="also run the functor prologue, its pullback and the 6-group Adam step each iteration "
                         "(SURVEY.md §8f rank 1; NOT part of the headline metric, reported under `trainer_tail`)")
    ap.add_argument("--unfused-tail", action="store_true",
                    help="with --with-optimizer: run prologue pullback, Adam and prologue as three kernels "
                         "instead of gsr_trainer_tail_step")
    ap.add_argument("--tail-in-backward", action="store_true",
                    help="with --with-optimizer on one GPU: gsr_backward_trainer_tail — the tail applied in the epilogue of "
                         "the per-Gaussian backward, the gradients never written")
    ap.add_argument("--mode", default="rgb", choices=["rgb", "rgbd", "rgbdn"],
                    help="render mode (the headline metric is :rgb; :rgbd is the reference's default training mode)")
    ap.add_argument("--ply", default=None, help="render a 3DGS .ply scene (gaussians.jl export_ply layout) instead of "
                                                "the synthetic one; N and the SH degree come from the file")
    ap.add_argument("--reference-lists", action="store_true",
                    help="GSR_FLAG_REFERENCE_TILE_LISTS: keep the reference's (Gaussian, tile) instance lists instead of "
                         "the library default (exact footprint culling)")
    ap.add_argument("--skew", default=None, metavar="KIND",
                    help="skewed variant of the synthetic scene (synthetic.add_skew): 'hot:K' = K extra Gaussians in ONE tile, "
                         "'dense:P:F' = a fraction P of the tiles at F x the mean density; reports tile_sort time and bins bytes")
    ap.add_argument("--order", default="random", choices=["random", "morton"],
                    help="order of the Gaussians in memory: 'random' = the synthetic scene as generated (the headline "
                         "configuration); 'morton' = the same Gaussians sorted along a 3-D Z-order curve (what a caller "
                         "could do at densification time) - reported, never the headline")
    ap.add_argument("--no-other-lists", action="store_true", help="skip the secondary timing of the other tile-list mode")
    args = ap.parse_args()

    if os.environ.get("GSR_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["GSR_BENCH_WATCHDOG"]), exit=True)
    pkg = gsr_pkg.load()
    D = pkg.distributed
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (the product path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    W, H, N, deg = args.width, args.height, args.n, args.sh_degree
    s = pkg.synthetic.make_scene(N if args.ply is None else 16, W, H, deg, args.seed)
    if args.skew:
        s = pkg.synthetic.add_skew(s, args.skew, args.seed)
    if args.order == "morton":
        s = pkg.synthetic.reorder(s, pkg.synthetic.morton_order(s.means))
        N = s.n
    if args.ply is not None:
        gm = pkg.ply.import_ply(args.ply)
        N, deg = gm.n, gm.max_sh_degree
        s.means, s.rotations = gm.points, gm.rotations
        s.shs = np.ascontiguousarray(np.concatenate([gm.features_dc, gm.features_rest], 1))
        s.scales_raw, s.opacities_raw, s.sh_degree = gm.scales, gm.opacities.reshape(-1), deg
    K = s.shs.shape[1]
    view = rank % args.views
    if world == 1:
        R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)  # §8(d): R = I, t = 0
    else:
        R, t = pkg.synthetic.view_pose(view, args.views)
    cam = pkg.Camera(W, H, tuple(s.focal), (0.5, 0.5), R, t)
    to = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    params = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
    target = to(pkg.synthetic.make_target(W, H, args.seed + view))
    vpix_fixed = to(pkg.synthetic.make_vpixels(W, H, pkg.rasterizer.n_color_features(args.mode), args.seed + view))
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode=args.mode, device=dev, exact_tile_cull=not args.reference_lists)
    # world > 1: factored exchange (distributed.py) unless GSR_DIST_FULL_ARENA=1 asks for the plain
    # all-reduce of the whole (11+3K)·N arena
    dist_on = world > 1 or D.forced()  # GSR_DIST_FORCE=1: the collectives run on a 1-rank RCCL communicator
    if args.tail_in_backward and (dist_on or not args.with_optimizer):
        raise SystemExit("--tail-in-backward is the single-GPU trainer step: it needs --with-optimizer and no gradient exchange")
    factored = dist_on and os.environ.get("GSR_DIST_FULL_ARENA", "0") != "1"
    overlap = factored and os.environ.get("GSR_DIST_NO_OVERLAP", "0") != "1"
    if overlap:
        D.overlap_groups()
    arena = torch.empty(D.factored_arena_numel(N) if factored else D.arena_numel(N, K), device=dev, dtype=torch.float32)
    if factored:
        centers = []
        for r in range(world):
            Rr, tr = pkg.synthetic.view_pose(r % args.views, args.views) if world > 1 else (R, t)
            centers.append(pkg.Camera(W, H, tuple(s.focal), (0.5, 0.5), Rr, tr).camera_center)
        centers_d = to(np.stack(centers).astype(np.float32))
        gathered = torch.empty(world * 3 * N, device=dev, dtype=torch.float32)
        vshs_sum = torch.empty((N, K, 3), device=dev, dtype=torch.float32)
    bg = (0.0, 0.0, 0.0)

    tail = None
    if args.with_optimizer:
        # raw parameters the trainer optimises (training.jl:234-239): points, f_dc, f_rest, opacity logits,
        # log-scales, rotations; one NU.Adam each
        raw = [params[0].clone(), params[1][:, :1].contiguous(), params[1][:, 1:].contiguous(),
               to(s.opacities_raw.reshape(-1, 1)), to(s.scales_raw), params[4].clone()]
        lrs = [1.6e-4, 2.5e-3, 2.5e-3 / 20, 2.5e-2, 5e-3, 1e-3]
        opts = [pkg.optim.Adam(t, lr, eps=1e-15) for t, lr in zip(raw, lrs)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tail = {"prologue_fwd": 0.0, "prologue_bwd": 0.0, "adam": 0.0, "n": 0, "n_steps": 0}
        opt_map = dict(zip(pkg.optim.GROUPS, opts))
        raw_map = dict(zip(pkg.optim.GROUPS, raw))

    def step():
        if tail is not None:
            ev[0].record()
            if args.unfused_tail or tail["n_steps"] == 0:
                shs, oa, sa = pkg.rasterizer.prologue_forward(raw[1], raw[2], raw[3], raw[4])
                params[1], params[2], params[3] = shs, oa, sa
            # (fused tail: the previous step's gsr_trainer_tail_step already wrote the activated copies)
            ev[1].record()
            params[0], params[4] = raw[0], raw[5]
            tail["n_steps"] += 1
        img = rast.forward_raw(*params, cam, deg, bg)
        if args.no_loss:
            vp = vpix_fixed
        else:
            _, vp = pkg.fused_ssim.l1_ssim_loss(rast, img, target)
        if tail is not None and args.tail_in_backward:
            ev[2].record(); ev[3].record()
            pkg.optim.fused_backward_tail_step(rast, vp, opt_map, raw_map, params[1], params[2], params[3], cam, deg, bg)
            e4 = torch.cuda.Event(enable_timing=True); e4.record()
            tail["_last"] = (ev[0], ev[1], ev[2], ev[3], e4)
            return
        rast.backward_raw(vp, *params, cam, deg, bg, arena=arena, factored_sh=factored)
        if factored and overlap:
            # all-gather(vc) || all-reduce(11·N): the ∇shs rebuild runs while the all-reduce is in flight
            D.exchange_factored_overlapped(arena, N, gathered, lambda vc_all: pkg.rasterizer.sh_grad_from_views(
                params[0], vc_all, centers_d, K, deg, out=vshs_sum))
        elif factored:
            vc_all = D.exchange_factored(arena, N, gathered)
            pkg.rasterizer.sh_grad_from_views(params[0], vc_all, centers_d, K, deg, out=vshs_sum)
        else:
            D.allreduce_arena(arena)
        if tail is not None:
            g = D.split_arena(arena, N, K) if not factored else dict(D.split_factored_arena(arena, N), vshs=vshs_sum)
            ev[2].record()
            if args.unfused_tail:
                vdc, vrest, vo, vs = pkg.rasterizer.prologue_backward(params[2], params[3], g["vshs"], g["vopacities"].view(-1, 1),
                                                                      g["vscales"], 3)
                ev[3].record()
                pkg.optim.step_all(opts, raw, [g["vmeans"], vdc, vrest, vo, vs, g["vrot"]])
            else:
                ev[3].record()
                pkg.optim.trainer_tail_step(opt_map, raw_map, dict(g, vopacities=g["vopacities"].view(-1, 1)),
                                            params[1], params[2], params[3])
            e4 = torch.cuda.Event(enable_timing=True); e4.record()
            tail["_last"] = (ev[0], ev[1], ev[2], ev[3], e4)

    def tail_collect():
        if tail is not None and "_last" in tail:
            a, b, c, d, e = tail["_last"]
            e.synchronize()
            tail["prologue_fwd"] += a.elapsed_time(b); tail["prologue_bwd"] += c.elapsed_time(d)
            tail["adam"] += d.elapsed_time(e); tail["n"] += 1

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # Stage survey (untimed): every stage timed with HIP events for a few steps -> stages_ms and the dominant stage.
    # In the TIMED region only the dominant stage keeps its event pair: an event record is a marker packet between two
    # kernels, and eight pairs per step cost 0.06 ms of a 1.8 ms step (measured: 1.81 vs 1.745 ms).
    rast.profile(True)
    for _ in range(5):
        step()
    sync()
    survey = {k: (ms / max(c, 1), c) for k, (ms, c) in rast.profile_read().items() if c > 0}
    dom = max(survey, key=lambda k: survey[k][0] * survey[k][1])
    rast.profile(True, stages=[dom])
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    # per-step times = launch-to-launch intervals of the dominant stage's own event pairs (a separate per-step
    # marker would be one more ~6 us bubble on the stream, tools/gap_report.py): K-1 samples
    per_step = sorted(rast.profile_intervals(dom))
    ms_median = (None if not per_step else per_step[len(per_step) // 2] if len(per_step) % 2 else
                 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2]))
    tail_collect()
    prof = rast.profile_read()
    rast.profile(False)
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    Dn, V = int(rast.stats.n_rendered), int(rast.stats.n_visible)
    ms_step = 1e3 * dt / args.steps
    value = world * P / (dt / args.steps) / 1e6

    live = {k: (ms / max(c, 1), c) for k, (ms, c) in prof.items() if c > 0}
    dom_ms = live[dom][0]            # the dominant kernel's mean launch time, HIP events INSIDE the timed region
    stages = dict(survey)            # the other stages: from the survey pass just before it
    stages[dom] = live[dom]
    Cn = pkg.rasterizer.n_color_features(args.mode)
    dom_bytes = algorithmic_bytes(dom, N, V, Dn, P, T, Cn, K)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    # PMC-counted HBM bytes and VALU instructions per launch: only a measurement of EXACTLY this configuration
    # (tools/pmc_workload.py + tools/pmc_parse.py under rocprofv3 --pmc, committed per round) is reported
    key = (config_key(N, W, H, deg, args.mode, not args.reference_lists, not args.no_loss)
           if args.ply is None and not args.skew and args.order == "random" else None)
    traffic, valu, pmc_src = None, None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if key is not None and os.path.exists(tpath) and not (tail is not None):
        try:
            rec = json.load(open(tpath)).get("configs", {}).get(key)
            if rec is not None and int(rec.get("tile_instances", -1)) == Dn:
                traffic = rec.get("hbm_bytes", {}).get(dom)
                insts = rec.get("sq", {}).get(dom, {}).get("SQ_INSTS_VALU")
                pmc_src = rec.get("source")
                if insts:
                    issue_ms = insts * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3
                    valu = {"kernel": dom, "insts": int(insts), "issue_cycles_peak": int(insts * VALU_CYCLES / N_SIMD),
                            "peak_ms_at_2.4GHz": round(issue_ms, 4), "frac": round(issue_ms / dom_ms, 4)}
        except Exception:
            traffic, valu = None, None
    # measured HBM ceiling of this device in this run: STREAM triad over 3 x 512 MiB
    lib = pkg._lib.load()
    n_tri = 128 * 1024 * 1024
    ta, tb, tc = (torch.ones(n_tri, device=dev) for _ in range(3))
    cs = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        pkg._lib.check(lib.gsr_stream_triad(ta.data_ptr(), tb.data_ptr(), tc.data_ptr(), n_tri, 0.5, cs))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pkg._lib.check(lib.gsr_stream_triad(ta.data_ptr(), tb.data_ptr(), tc.data_ptr(), n_tri, 0.5, cs))
    e1.record(); e1.synchronize()
    triad_gbs = 5 * 12.0 * n_tri / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del ta, tb, tc
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "valu": valu, "pmc_source": pmc_src,
                "pmc_config_key": key,
                "measured_triad_GBps": round(triad_gbs, 1), "frac_of_measured_triad": round(achieved / triad_gbs, 5),
                "algorithmic_bytes": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4),
                "avg_launch_ms_source": f"HIP events around {dom} on the launch stream, {live[dom][1]} launches inside the timed region",
                "stages_ms": {k: round(v[0], 4) for k, v in stages.items()},
                "stages_ms_source": "5-step survey with every stage timed, just before the timed region (all stages timed "
                                    "inside it would slow the step by 3 %); the dominant stage: the timed region",
                "whole_step_algorithmic_GBps": round(
                    sum(algorithmic_bytes(k, N, V, Dn, P, T, Cn, K) for k in stages) / (ms_step * 1e-3) / 1e9, 2)}

    out = {
        "metric": "fwd+bwd Mpixels/s @1920x1080, 1M Gaussians SH=3",
        "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "ms_per_step_median": round(ms_median if ms_median is not None else ms_step, 4),
        "ms_per_step_max": round(per_step[-1], 4) if per_step else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if args.ply is None else "ply scene, synthetic camera and target",
        "config": {"workload": ("config3: 1M Gaussians, SH deg 3, 1920x1080, fwd + L1/0.2*DSSIM loss + bwd"
                                if not args.no_loss and args.ply is None and args.mode == "rgb" and (N, W, H, deg) == (1_000_000, 1920, 1080, 3) else
                                f"ply scene {os.path.basename(args.ply)}: N={N} SH{deg} {W}x{H} fwd{'' if args.no_loss else '+loss'}+bwd"
                                if args.ply is not None else
                                f"N={N} SH{deg} {W}x{H} :{args.mode} fwd{'' if args.no_loss else '+loss'}+bwd"
                                + (f" skew={args.skew}" if args.skew else "")
                                + (" [Gaussians in Morton order: NOT the headline configuration]" if args.order == "morton" else "")),
                   "n_gaussians": N, "visible": V, "tile_instances": Dn, "views_per_gpu": 1,
                   "binning": {"mode": "compact (count -> scan -> scatter)" if rast.stats.compact_binning else "fixed-capacity bins",
                               "unsorted_key_bytes": int(rast.stats.bins_bytes), "longest_tile_list": int(rast.stats.max_tile_instances),
                               "handle_bytes": int(rast.memory_usage())},
                   "tile_lists": ("reference lists (GSR_FLAG_REFERENCE_TILE_LISTS)" if args.reference_lists else
                                  "library default: exact footprint cull (same image / gradients)"),
                   "parallelism": (f"view-parallel x{world}, all-reduce of {11 * N * 4 / 1e6:.0f} MB + all-gather of "
                                   f"{world} x {3 * N * 4 / 1e6:.0f} MB colour cotangents (factored SH gradient"
                                   f"{'; the two collectives overlapped on two communicators' if overlap else ''}); "
                                   f"GSR_DIST_FULL_ARENA=1 selects the plain all-reduce of the whole arena" if factored else
                                   f"view-parallel x{world}, 1 all-reduce of {arena.numel() * 4 / 1e6:.0f} MB")},
        "roofline": roofline,
    }

    if tail is not None and tail["n"]:
        # last iteration's HIP-event times of the three extra stages (they ARE inside ms_per_step here)
        out["trainer_tail"] = {k: round(tail[k] / tail["n"], 4) for k in ("prologue_fwd", "prologue_bwd", "adam")}
        out["trainer_tail"]["algorithmic_bytes"] = {"prologue_fwd": 2 * 4 * (3 * K + 4) * N, "prologue_bwd": 2 * 4 * (3 * K + 4) * N + 16 * N,
                                                    "adam": 7 * 4 * (3 * K + 11) * N}
        out["trainer_tail"]["form"] = ("three kernels" if args.unfused_tail else
                                       "inside the backward (gsr_backward_trainer_tail: 'adam' = composite_bwd + per-Gaussian backward + tail)"
                                       if args.tail_in_backward else "fused (gsr_trainer_tail_step: 'adam' is the whole tail)")
        out["config"]["workload"] += " + prologue + Adam (trainer tail, not the headline metric)"
    if world == 1 and not dist_on and tail is None and not args.no_other_lists:
        # the same step with the OTHER tile-list mode, timed in the same run (headline = the library default)
        rast2 = pkg.rasterizer.GaussianRasterizer(W, H, mode=args.mode, device=dev, exact_tile_cull=args.reference_lists)
        rast_main, rast = rast, rast2
        for _ in range(max(args.warmup, 2)):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt2 = (time.perf_counter() - t0) / args.steps
        out["other_tile_lists"] = {"tile_lists": "exact footprint cull" if args.reference_lists else "reference lists",
                                   "ms_per_step": round(1e3 * dt2, 4), "value": round(P / dt2 / 1e6, 3),
                                   "tile_instances": int(rast2.stats.n_rendered)}
        rast = rast_main
        rast2.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, s, W, H, deg, args)
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


def cpu_baseline(pkg, s, W, H, deg, args):
    """The oracle (line-for-line C restatement of the reference algorithm; the reference
    itself has no CPU path) on this host's cores: one full step of the same workload."""
    from oracle import oracle as orc
    cores = orc.num_threads()
    cam = orc.Camera(W, H, s.focal)
    tgt = pkg.synthetic.make_target(W, H, args.seed)
    t0 = time.perf_counter()
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg)
    if args.no_loss:
        vp = pkg.synthetic.make_vpixels(W, H, 3, args.seed)
    else:
        _, vp = orc.loss_head(st.image, tgt)
    orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, deg, deterministic=False)
    dt = time.perf_counter() - t0
    return {"value": round(W * H / dt / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"1 full step of the same workload ({dt:.1f} s wall); backward = the reference's atomic-accumulation "
                      f"form over OpenMP threads (non-deterministic summation order, as render.jl:242,275-282)"}


if __name__ == "__main__":
    main()
