"""-m gpu tests of the multi-GPU path as far as ONE GPU allows (the 8-GPU scaling run is the
driver's): (a) two fresh rank processes sharing the device run the HIP forward + backward of their view into the
arena and then both exchange forms — the result must equal the sum of the per-view HIP gradients computed
sequentially; (b) a one-rank RCCL communicator drives gsr_allreduce_grads (dlopen / dlsym / argument order of the
raw-ncclComm_t entry point)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch

from hip_helpers import dev, rel_l2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_hip_backward_then_exchange_equals_sum_of_views(tmp_path, pkg, launch_ranks):
    """Two ranks share the one GPU of the box; gloo carries the collectives (RCCL refuses two ranks on one device)."""
    _ranks_vs_sequential(tmp_path, pkg, launch_ranks, 2, {"GSR_DIST_BACKEND": "gloo"})


def test_one_rank_rccl_communicator_runs_both_exchange_forms(tmp_path, pkg, launch_ranks):
    """GSR_DIST_FORCE=1: the SAME worker on a one-rank "nccl" (= RCCL) process group — communicator creation, the two
    extra communicators of the overlapped factored exchange, async work handles and stream joins all execute for real;
    a sum over one rank must reproduce the single-view HIP gradients."""
    _ranks_vs_sequential(tmp_path, pkg, launch_ranks, 1, {"GSR_DIST_FORCE": "1"})


def _ranks_vs_sequential(tmp_path, pkg, launch_ranks, world, extra_env):
    import dist_gpu_worker as Wk
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}
    env.update(extra_env)
    rc, out = launch_ranks([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), str(tmp_path)], world, env=env, timeout=300)
    assert rc == [0] * world, "\n".join(out)
    # sequential single-process reference: the same HIP kernels, view after view
    D = pkg.distributed
    s = Wk.scene(pkg)
    N, K = s.n, s.shs.shape[1]
    p = [dev(s.means), dev(s.shs), dev(s.opacities.reshape(-1, 1)), dev(s.scales), dev(s.rotations)]
    rast = pkg.rasterizer.GaussianRasterizer(Wk.W, Wk.H, mode="rgb")
    total = torch.zeros(D.arena_numel(N, K), device="cuda")
    radii = []
    for v in range(world):
        cam, vp = Wk.view_inputs(pkg, s, v, world)
        arena = torch.empty_like(total)
        rast.forward_raw(*p, cam, Wk.DEG, (0, 0, 0))
        rast.backward_raw(dev(vp), *p, cam, Wk.DEG, (0, 0, 0), arena=arena)
        total += arena
        radii.append(rast.gstate.radii.cpu().numpy().copy())
    ref = D.split_arena(total, N, K)
    assert float(total.abs().max()) > 0
    for r in range(world):
        plain = torch.from_numpy(np.load(tmp_path / f"plain_{r}.npy"))
        assert torch.equal(plain, total.cpu()), "all-reduce of two views == sum of the two (same fp32 adds)"
        small = np.load(tmp_path / f"fact_small_{r}.npy")
        got = dict(vrot=small[:4 * N], vmeans=small[4 * N:7 * N], vopacities=small[7 * N:8 * N], vscales=small[8 * N:11 * N])
        for k, a in got.items():
            assert np.array_equal(a.reshape(-1), ref[k].cpu().numpy().reshape(-1)), k
        vshs = np.load(tmp_path / f"fact_vshs_{r}.npy")
        assert rel_l2(vshs.reshape(-1), ref["vshs"].cpu().numpy().reshape(-1)) <= 1e-6
        assert np.array_equal(np.load(tmp_path / f"radii_{r}.npy"), radii[r]), "per-view side outputs stay local"
        # the multi-GPU trainer step in one pass == rebuild -> gsr_trainer_tail_step, and every rank ends on the same parameters
        assert list(np.load(tmp_path / f"tail_{r}.npy")) == [1, 1]
        assert np.array_equal(np.load(tmp_path / f"tail_points_{r}.npy"), np.load(tmp_path / "tail_points_0.npy"))


class _NcclId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_gsr_allreduce_grads_on_a_one_rank_rccl_communicator(pkg):
    """ncclCommInitRank(nranks = 1) + gsr_allreduce_grads: proves librccl is found by the library's dlopen, the
    symbol resolves and the argument order / dtype / op constants are right (a sum over one rank is the identity;
    a wrong dtype or count would corrupt the arena or fail)."""
    L = pkg._lib
    lib = L.load()
    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        rccl = C.CDLL("librccl.so.1")
    uid = _NcclId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_NcclId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0 and comm.value
    n = 59 * 4099  # an arena-sized, odd element count
    arena = torch.randn(n, device="cuda")
    keep = arena.clone()
    stream = torch.cuda.current_stream().cuda_stream
    L.check(lib.gsr_allreduce_grads(comm, arena.data_ptr(), n, stream))
    torch.cuda.synchronize()
    assert torch.equal(arena, keep)
    assert lib.gsr_allreduce_grads(None, arena.data_ptr(), n, stream) == L.GSR_E_INVALID_ARG
    rccl.ncclCommDestroy(comm)
