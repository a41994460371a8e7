"""Round 6: the host time of a densification round varies from run to run (1.6-5.5 ms in one bench run, 44 and 82 ms for the last two
rounds in another).  Three protocol runs in one process and per round: densify_and_prune, rast.reserve, the eager prologue."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gsr_pkg, torch
import train_harness as TH
pkg = gsr_pkg.load()
Dz = pkg.densification
sync = torch.cuda.synchronize
for rep in range(3):
    h = TH.Harness(pkg, TH.Protocol(densify_grad_threshold=4e-5))
    log = []
    real_dp, real_res, real_pro = Dz.densify_and_prune, h.rast.reserve, h.prologue
    def dp(*a, **k):
        sync(); t0 = time.perf_counter(); r = real_dp(*a, **k); sync(); log.append(["densify", round(1e3 * (time.perf_counter() - t0), 2)]); return r
    def res(*a, **k):
        sync(); t0 = time.perf_counter(); r = real_res(*a, **k); sync(); log[-1] += ["reserve", round(1e3 * (time.perf_counter() - t0), 2), a]; return r
    Dz.densify_and_prune, h.rast.reserve = dp, res
    for _ in range(1500):
        h.step()
    sync()
    Dz.densify_and_prune = real_dp
    print("run", rep, "N", len(h.gs), "host_ms per round", [d["host_ms"] for d in h.densify_log])
    for l in log:
        print("   ", l)
    st = torch.cuda.memory_stats()
    print("   torch: device allocs", st["num_device_alloc"], "frees", st["num_device_free"], "reserved GB", round(st["reserved_bytes.all.current"] / 2**30, 2),
          "handle MB", h.rast.memory_usage() >> 20)
    h.rast.close(); del h
