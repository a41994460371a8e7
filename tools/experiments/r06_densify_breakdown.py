"""Round 6: where do the host milliseconds of one densification round go?  Runs the training protocol to step 1299, then times the
pieces of post_train_step at step 1300 (N ~ 1.0 M -> 1.13 M): every helper of densification.py wrapped with a synchronise before and
after.  Synchronised pieces add up to MORE than the unsynchronised round (printed last, step 1400)."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gsr_pkg, torch
import train_harness as TH
pkg = gsr_pkg.load()
Dz = pkg.densification
p = TH.Protocol(densify_grad_threshold=4e-5)
h = TH.Harness(pkg, p)
for _ in range(1299):
    h.step()
sync = torch.cuda.synchronize
acc = collections.OrderedDict()
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        sync(); t0 = time.perf_counter()
        r = f(*a, **k)
        sync(); acc[name] = acc.get(name, 0.0) + 1e3 * (time.perf_counter() - t0); acc[name + "#"] = acc.get(name + "#", 0) + 1
        return r
    setattr(mod, name, g)
    return f
saved = {n: wrap(Dz, n) for n in ("_mask", "findall", "_compose", "_reset_stats", "select")}
alloc = {"ms": 0.0, "n": 0}
real_empty, real_zeros = torch.empty, torch.zeros
def timed(fn):
    def g(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); alloc["ms"] += 1e3 * (time.perf_counter() - t0); alloc["n"] += 1
        return r
    return g
torch.empty, torch.zeros = timed(real_empty), timed(real_zeros)
sync(); t0 = time.perf_counter()
h.step()   # step 1300: densifies
sync(); total = 1e3 * (time.perf_counter() - t0)
torch.empty, torch.zeros = real_empty, real_zeros
for n, f in saved.items():
    setattr(Dz, n, f)
print(f"step 1300 (synchronised pieces), N -> {len(h.gs)}: whole step {total:.2f} ms")
for k, v in acc.items():
    if not k.endswith("#"):
        print(f"  {k:14s} {v:7.3f} ms in {acc[k + '#']} calls")
print(f"  torch.empty / zeros (host time inside the above): {alloc['ms']:.3f} ms in {alloc['n']} calls")
for _ in range(99):
    h.step()
sync(); t0 = time.perf_counter()
h.step()
sync()
print(f"step 1400 unsynchronised: whole step {1e3 * (time.perf_counter() - t0):.2f} ms, densify_log {h.densify_log[-1]}")
print("torch allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.current", "allocated_bytes.all.current")})
