"""Do two independent views on two streams overlap on one MI355X?  (memory- / atomic-bound stages of one view under the
VALU-bound compositing of the other)   python tools/experiments/two_streams.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import gsr_pkg
pkg = gsr_pkg.load()
W, H, N, deg = 1920, 1080, 1_000_000, 3
s = pkg.synthetic.make_scene(N, W, H, deg, 1003)
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t = [to(s.means), to(s.shs), to(s.opacities.reshape(-1, 1)), to(s.scales), to(s.rotations)]
cam = pkg.Camera(W, H, tuple(s.focal))
vp = torch.randn(H, W, 3, device=dev)
K = 16


def worker(rast, stream, iters, arena):
    with torch.cuda.stream(stream):
        for _ in range(iters):
            rast.forward_raw(*t, cam, deg, (0, 0, 0))
            rast.backward_raw(vp, *t, cam, deg, (0, 0, 0), arena=arena)


rasts = [pkg.rasterizer.GaussianRasterizer(W, H, mode="rgb", device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
arenas = [torch.empty((11 + 3 * K) * N, device=dev) for _ in range(2)]
for r, st, a in zip(rasts, streams, arenas):
    worker(r, st, 3, a)
torch.cuda.synchronize()
t0 = time.perf_counter(); worker(rasts[0], streams[0], 40, arenas[0]); torch.cuda.synchronize(); one = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(r, st, 20, a)) for r, st, a in zip(rasts, streams, arenas)]
[x.start() for x in th]; [x.join() for x in th]
torch.cuda.synchronize(); two = time.perf_counter() - t0
print(f"40 views on one stream: {1e3 * one / 40:.3f} ms per view; 2 x 20 views on two streams: {1e3 * two / 40:.3f} ms per view "
      f"({100 * (1 - two / one):.1f} % less)")
