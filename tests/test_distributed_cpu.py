"""world_size-2 gloo test of the multi-view path's host logic: each rank
owns one view, fills the gradient arena (here with the ORACLE's per-view gradients — the
HIP rasterizer has no CPU path), and the single all-reduce must equal the sum of the
per-view gradients computed sequentially."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _view_arena(pkg, orc, D, s, view, n_views):
    R, t = pkg.synthetic.view_pose(view, n_views)
    cam = orc.Camera(s.width, s.height, s.focal, R=R, t=t)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, s.sh_degree)
    vp = pkg.synthetic.make_vpixels(s.width, s.height, 3, 100 + view) * 1e3
    g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, s.sh_degree)
    n, K = s.n, s.shs.shape[1]
    arena = torch.zeros(D.arena_numel(n, K))
    v = D.split_arena(arena, n, K)
    v["vrot"].copy_(torch.from_numpy(g.vrots)); v["vmeans"].copy_(torch.from_numpy(g.vmeans))
    v["vshs"].copy_(torch.from_numpy(g.vshs)); v["vopacities"].copy_(torch.from_numpy(g.vopacities))
    v["vscales"].copy_(torch.from_numpy(g.vscales))
    return arena


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import gsr_pkg
    pkg = gsr_pkg.load()
    from oracle import oracle as orc
    D = pkg.distributed
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    s = pkg.synthetic.make_scene(200, 64, 48, 1, 31, sigma_px=4.0)
    mine = D.views_of_rank(rank, world, world)
    assert mine == [rank]
    arena = _view_arena(pkg, orc, D, s, mine[0], world)
    D.allreduce_arena(arena)
    np.save(os.path.join(out_dir, f"arena_{rank}.npy"), arena.numpy())
    torch.distributed.destroy_process_group()


def test_two_rank_allreduce_equals_sum_of_views(tmp_path, pkg, orc):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    D = pkg.distributed
    s = pkg.synthetic.make_scene(200, 64, 48, 1, 31, sigma_px=4.0)
    ref = sum(_view_arena(pkg, orc, D, s, v, world) for v in range(world)).numpy()
    for r in range(world):
        got = np.load(tmp_path / f"arena_{r}.npy")
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-9)
    assert np.abs(ref).max() > 0


def test_arena_layout(pkg):
    D = pkg.distributed
    n, K = 10, 16
    assert D.arena_numel(n, K) == 59 * n
    a = torch.arange(D.arena_numel(n, K), dtype=torch.float32)
    v = D.split_arena(a, n, K)
    assert v["vrot"].shape == (n, 4) and v["vshs"].shape == (n, K, 3)
    assert float(v["vmeans"][0, 0]) == 4 * n and float(v["vscales"][-1, -1]) == 59 * n - 1
    assert D.views_of_rank(1, 4, 8) == [1, 5]


# ---- factored exchange: all-reduce of 11·N floats + all-gather of the (N,3) colour cotangents ----
def _view_factored(pkg, orc, D, s, view, n_views):
    R, t = pkg.synthetic.view_pose(view, n_views)
    cam = orc.Camera(s.width, s.height, s.focal, R=R, t=t)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, cam, s.sh_degree)
    vp = pkg.synthetic.make_vpixels(s.width, s.height, 3, 100 + view) * 1e3
    g = orc.backward(st, vp, s.means, s.shs, s.opacities, s.scales, s.rotations, cam, s.sh_degree)
    n = s.n
    arena = torch.zeros(D.factored_arena_numel(n))
    v = D.split_factored_arena(arena, n)
    v["vrot"].copy_(torch.from_numpy(g.vrots)); v["vmeans"].copy_(torch.from_numpy(g.vmeans))
    v["vopacities"].copy_(torch.from_numpy(g.vopacities)); v["vscales"].copy_(torch.from_numpy(g.vscales))
    vc = g.vfeatures[:, :3] * (1.0 - st.clamped.astype(np.float32))   # colour cotangent after the clamp mask
    v["vcolors"].copy_(torch.from_numpy(np.ascontiguousarray(vc, np.float32)))
    return arena, g, cam


def _rebuild_vshs(orc, s, vcolors_all, cams):
    """Σ_v basis(dir_v) x vc_v with the oracle's ∇SH (clamp mask already folded into vc)."""
    tot = np.zeros_like(s.shs, dtype=np.float64)
    for vc, cam in zip(vcolors_all, cams):
        scratch = np.zeros((s.n, 3), np.float32)
        tot += orc.sh_backward(s.means, cam.camera_center, s.shs, np.zeros((s.n, 3), bool),
                               np.ascontiguousarray(vc, np.float32), s.sh_degree, scratch)
    return tot


def _worker_factored(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import gsr_pkg
    pkg = gsr_pkg.load()
    from oracle import oracle as orc
    D = pkg.distributed
    D.init_from_env("gloo")
    s = pkg.synthetic.make_scene(200, 64, 48, 2, 33, sigma_px=4.0)
    arena, _, _ = _view_factored(pkg, orc, D, s, rank, world)
    # overlapped form (two communicators, the rebuild between the two waits) == sequential form
    arena2 = arena.clone()
    D.overlap_groups()
    gathered = torch.empty(world * 3 * s.n)
    vc_ov = D.exchange_factored_overlapped(arena2, s.n, gathered, lambda vc_all: vc_all.clone())
    vc_all = D.exchange_factored(arena, s.n)
    assert torch.equal(arena2, arena) and torch.equal(vc_ov, vc_all)
    np.save(os.path.join(out_dir, f"farena_{rank}.npy"), arena.numpy())
    np.save(os.path.join(out_dir, f"vcall_{rank}.npy"), vc_all.numpy())
    torch.distributed.destroy_process_group()


def test_two_rank_factored_exchange_equals_sum_of_views(tmp_path, pkg, orc):
    world, port = 2, _free_port()
    mp.spawn(_worker_factored, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    D = pkg.distributed
    s = pkg.synthetic.make_scene(200, 64, 48, 2, 33, sigma_px=4.0)
    per_view = [_view_factored(pkg, orc, D, s, v, world) for v in range(world)]
    n = s.n
    ref_small = sum(a[:11 * n] for a, _, _ in per_view).numpy()
    ref_vshs = sum(g.vshs.astype(np.float64) for _, g, _ in per_view)
    cams = [c for _, _, c in per_view]
    for r in range(world):
        arena = np.load(tmp_path / f"farena_{r}.npy")
        vc_all = np.load(tmp_path / f"vcall_{r}.npy")
        assert np.allclose(arena[:11 * n], ref_small, rtol=1e-6, atol=1e-9)
        for v in range(world):  # rank-major == view-major
            assert np.array_equal(vc_all[v], per_view[v][0][11 * n:].view(n, 3).numpy())
        rebuilt = _rebuild_vshs(orc, s, vc_all, cams)
        assert np.abs(ref_vshs).max() > 0
        assert np.linalg.norm(rebuilt - ref_vshs) <= 1e-6 * np.linalg.norm(ref_vshs)
    assert D.factored_arena_numel(10) == 140
