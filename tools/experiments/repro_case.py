#!/usr/bin/env python3
"""Re-run single fuzz cases with the forward comparison's numbers printed (which pixels differ, by how much, the binning mode).
  python tools/experiments/repro_case.py sweep|edge|deep CASE ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import fuzz_parity as F  # noqa: E402
import numpy as np  # noqa: E402
from hip_helpers import blend_boundary_pixels  # noqa: E402

T = F.T
orig = T._compare_forward


def verbose(st, run, img, opacities=None):
    im = img.cpu().numpy()
    d = np.abs(im - st.image).max(axis=-1)
    bad = d > 1e-4 * np.maximum(1.0, np.abs(st.image).max(axis=-1))
    keep = np.ones_like(bad)
    if opacities is not None and im.shape[0] * im.shape[1] < 20000:
        keep = ~blend_boundary_pixels(st, opacities, im.shape[1], im.shape[0])
    ys, xs = np.nonzero(bad & keep)
    print(f"  image {im.shape}, binning mode {int(run.rast.stats.compact_binning)}, longest list {int(run.rast.stats.max_tile_instances)}, "
          f"pixels off {int(bad.sum())} (kept {int((bad & keep).sum())}), max diff {d.max():.3e}", flush=True)
    for y, x in list(zip(ys, xs))[:6]:
        print(f"    pixel ({x},{y}) tile ({x // 16},{y // 16}) hip {im[y, x]} oracle {st.image[y, x]} n_contrib {int(run.rast.n_contrib.cpu().numpy().reshape(im.shape[:2])[y, x])} / {int(st.n_contrib.reshape(im.shape[:2])[y, x])}")
    ids = run.rast.values_sorted.cpu().numpy().astype(np.uint32)
    print("  ids equal:", ids.shape == st.values_sorted.shape and bool(np.array_equal(ids, st.values_sorted)))
    return orig(st, run, img, opacities)


T._compare_forward = verbose
fam = args[0]
for c in args[1:]:
    c = int(c)
    print(fam, c, flush=True)
    try:
        if fam == "sweep":
            T.test_randomised_sweep_vs_oracle(F.pkg, F.orc, c)
        elif fam == "edge":
            F.edge_case(c)
        else:
            F.deep_case(c)
        print("  passed")
    except AssertionError as e:
        import traceback
        tb = traceback.extract_tb(e.__traceback__)
        print("  FAILED at", tb[-1].filename.split("/")[-1], tb[-1].lineno, str(e)[:200])
