"""One rank of the multi-view step on the HIP path: HIP forward + backward of view `rank` into the
gradient arena, then the exchange.  Launched as a fresh process per rank by tests/_launcher.py; several ranks may
share one GPU (GSR_DIST_BACKEND=gloo moves the collectives through host memory) — the logic under test is the
same as with one GPU per rank over RCCL (GSR_DIST_BACKEND unset -> "nccl")."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsr_pkg  # noqa: E402

N, W, H, DEG, SEED = 3000, 160, 96, 2, 41


def scene(pkg):
    return pkg.synthetic.make_scene(N, W, H, DEG, SEED, sigma_px=4.0)


def view_inputs(pkg, s, view, n_views):
    R, t = pkg.synthetic.view_pose(view, n_views)
    cam = pkg.Camera(W, H, tuple(s.focal), (0.5, 0.5), R, t)
    vp = pkg.synthetic.make_vpixels(W, H, 3, 100 + view) * 1e3
    return cam, vp


def main():
    out_dir = sys.argv[1]
    pkg = gsr_pkg.load()
    D = pkg.distributed
    rank, world, local = D.init_from_env()
    dev = torch.device("cuda", local)
    s = scene(pkg)
    K = s.shs.shape[1]
    to = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    p = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
    cam, vp = view_inputs(pkg, s, rank, world)
    rast = pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", device=dev)
    # plain form: ONE all-reduce of the (11+3K)·N arena
    arena = torch.empty(D.arena_numel(N, K), device=dev)
    rast.forward_raw(*p, cam, DEG, (0, 0, 0))
    rast.backward_raw(to(vp), *p, cam, DEG, (0, 0, 0), arena=arena)
    D.allreduce_arena(arena)
    np.save(os.path.join(out_dir, f"plain_{rank}.npy"), arena.cpu().numpy())
    # factored form, collectives overlapped on two communicators
    centers = []
    for v in range(world):
        c, _ = view_inputs(pkg, s, v, world)
        centers.append(c.camera_center)
    centers_d = to(np.stack(centers).astype(np.float32))
    farena = torch.empty(D.factored_arena_numel(N), device=dev)
    gathered = torch.empty(world * 3 * N, device=dev)
    vshs = torch.empty((N, K, 3), device=dev)
    D.overlap_groups()
    rast.forward_raw(*p, cam, DEG, (0, 0, 0))
    rast.backward_raw(to(vp), *p, cam, DEG, (0, 0, 0), arena=farena, factored_sh=True)
    D.exchange_factored_overlapped(farena, N, gathered, lambda vc_all: pkg.rasterizer.sh_grad_from_views(
        p[0], vc_all, centers_d, K, DEG, out=vshs))
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"fact_small_{rank}.npy"), farena[:11 * N].cpu().numpy())
    np.save(os.path.join(out_dir, f"fact_vshs_{rank}.npy"), vshs.cpu().numpy())
    # the trainer step made self-contained (gsr_sh_grad_from_views_tail): from the SAME exchanged data, rebuild + Adam in one
    # pass must leave the parameters where rebuild -> gsr_trainer_tail_step leaves them (bit for bit, on every rank)
    O = pkg.optim
    lrs = dict(points=1.6e-4, features_dc=2.5e-3, features_rest=2.5e-3 / 20, opacities=2.5e-2, scales=5e-3, rotations=1e-3)

    def trainer():
        raw = dict(points=p[0].clone(), features_dc=p[1][:, :1].contiguous(), features_rest=p[1][:, 1:].contiguous(),
                   opacities=to(s.opacities_raw.reshape(-1, 1)), scales=to(s.scales_raw), rotations=p[4].clone())
        opts = {k: O.Adam(raw[k], lrs[k], eps=1e-15) for k in O.GROUPS}
        act = list(pkg.rasterizer.prologue_forward(raw["features_dc"], raw["features_rest"], raw["opacities"], raw["scales"]))
        return raw, opts, act

    g = D.split_factored_arena(farena, N)
    small = dict(vmeans=g["vmeans"], vopacities=g["vopacities"].view(-1, 1), vscales=g["vscales"], vrot=g["vrot"])
    vc_all = gathered.view(world, N, 3)
    raw_a, opt_a, act_a = trainer()
    raw_b, opt_b, act_b = trainer()
    O.sh_views_tail_step(opt_a, raw_a, small, vc_all, centers_d, DEG, *act_a)
    O.trainer_tail_step(opt_b, raw_b, dict(small, vshs=vshs), *act_b)
    torch.cuda.synchronize()
    same = all(torch.equal(raw_a[k], raw_b[k]) and torch.equal(opt_a[k].mu, opt_b[k].mu) and torch.equal(opt_a[k].nu, opt_b[k].nu)
               for k in O.GROUPS) and all(torch.equal(a, b) for a, b in zip(act_a, act_b))
    moved = float((raw_a["features_rest"] - p[1][:, 1:]).abs().max()) > 0
    np.save(os.path.join(out_dir, f"tail_{rank}.npy"), np.array([int(same), int(moved)]))
    np.save(os.path.join(out_dir, f"tail_points_{rank}.npy"), raw_a["points"].cpu().numpy())
    # per-view side outputs stay local (they feed per-view densification statistics)
    np.save(os.path.join(out_dir, f"radii_{rank}.npy"), rast.gstate.radii.cpu().numpy())
    if torch.distributed.is_initialized():
        print(f"backend {torch.distributed.get_backend()}, overlap groups {'on' if D._overlap_groups else 'off'}")
        torch.distributed.destroy_process_group()
    print(f"rank {rank}/{world} ok on {dev}")


if __name__ == "__main__":
    main()
