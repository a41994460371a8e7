"""Round-4 paper experiment: how much of composite_bwd's time is load imbalance across SIMDs, and how does it move with the
number of resident waves per SIMD?  (Measured: :rgb 789 / 735 / 677 / 695 us at 4 / 5 / 6 / 7 waves; :rgbd 724 us at 5
waves, 770 at 6 — more resident waves are not monotonically better, although the kernel is VALU-issue-bound.)

Model: 1024 SIMDs (256 CUs x 4), S wave slots each; one wave per tile, work(tile) = c0 + instances(tile) (the tile lists of the
config-3 scene, from the CPU oracle: the reference's lists — the exact-cull lists are a 0.69 x subsequence with the same
spatial distribution); tiles are dispatched in descending list length (the library's launch order) to the SIMD with a free slot
that was freed first (workgroup ids go to XCDs round-robin, inside an XCD to the first CU with room); a SIMD shares its issue
slots equally among its resident waves (processor sharing), with an efficiency eff(k) for k resident waves that models latency
hiding: eff = min(1, k / k_sat).  Output: makespan / (total work / 1024) per S, for k_sat = 1 (pure issue-bound) .. 6.

  python tools/experiments/bwd_balance_sim.py            (CPU only: oracle + numpy; ~1 min)"""
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gsr_pkg  # noqa: E402

pkg = gsr_pkg.load()
from oracle import oracle as orc  # noqa: E402


def tile_lengths(n=1_000_000, W=1920, H=1080, deg=3, seed=1003):
    s = pkg.synthetic.make_scene(n, W, H, deg, seed)
    st = orc.forward(s.means, s.shs, s.opacities, s.scales, s.rotations, orc.Camera(W, H, s.focal), deg)
    return (st.ranges[:, 1] - st.ranges[:, 0]).astype(np.float64)


def simulate(work, n_simd, slots, k_sat):
    """Event-driven processor sharing: returns the makespan."""
    order = np.argsort(-work, kind="stable")
    remaining = [dict() for _ in range(n_simd)]    # simd -> {tile: remaining work}
    t = 0.0
    nxt = 0
    # initial fill: round-robin over SIMDs, slot by slot
    for sl in range(slots):
        for i in range(n_simd):
            if nxt < len(order):
                remaining[i][int(order[nxt])] = float(work[order[nxt]]); nxt += 1
    # per-SIMD next-completion events
    def rate(k):
        return min(1.0, k / k_sat) / k   # progress per wave per unit time
    clock = [0.0] * n_simd               # time up to which SIMD i's remaining[] is valid
    heap = []
    for i in range(n_simd):
        if remaining[i]:
            k = len(remaining[i]); m = min(remaining[i].values())
            heapq.heappush(heap, (m / rate(k), i))
    end = 0.0
    while heap:
        t, i = heapq.heappop(heap)
        k = len(remaining[i])
        adv = (t - clock[i]) * rate(k)
        done = [tl for tl, r in remaining[i].items() if r - adv <= 1e-9]
        for tl in list(remaining[i]):
            remaining[i][tl] -= adv
        for tl in done:
            del remaining[i][tl]
            if nxt < len(order):            # the freed slot takes the next tile of the launch order
                remaining[i][int(order[nxt])] = float(work[order[nxt]]); nxt += 1
        clock[i] = t
        end = max(end, t)
        if remaining[i]:
            k = len(remaining[i]); m = min(remaining[i].values())
            heapq.heappush(heap, (t + m / rate(k), i))
    return end


if __name__ == "__main__":
    L = tile_lengths()
    print(f"tiles {L.size}, instances {int(L.sum())}, longest {int(L.max())}, mean {L.mean():.1f}, cv {L.std() / L.mean():.3f}")
    for c0 in (0.0, 60.0):
        work = L + c0
        ideal = work.sum() / 1024
        print(f"\nper-tile overhead c0 = {c0:.0f} instances; ideal (perfect balance, full issue rate) = {ideal:.0f}")
        print("  slots/SIMD   " + "   ".join(f"k_sat={k}" for k in (1, 3, 4, 5, 6)))
        for S in (3, 4, 5, 6, 7, 8):
            row = [simulate(work, 1024, S, k) / ideal for k in (1, 3, 4, 5, 6)]
            print(f"  {S}            " + "   ".join(f"{r:6.3f}" for r in row))
