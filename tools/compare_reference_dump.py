#!/usr/bin/env python3
"""Compare a live run of the reference (julia/dump_reference_goldens.jl) with the committed oracle fixtures
(tests/golden/scene_*.npz): the missing pin of SURVEY.md §8(c) ("parity unpinned": absolute image values, tie order).

    python tools/compare_reference_dump.py tests/golden out_dir
Tolerances: integers exact; image |Δ| <= 1e-4 on >= 99.99 %; gradients rel-L2 <= 1e-4 (the reference's atomics
are not deterministic, so gradients are compared in the L2 sense only)."""
import os
import sys

import numpy as np


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def main(golden_dir, out_dir):
    ok = True
    for mode in ("rgb", "rgbd", "rgbdn"):
        g = np.load(os.path.join(golden_dir, f"scene_{mode}.npz"))
        r = np.load(os.path.join(out_dir, f"ref_scene_{mode}.npz"))
        print(f"== scene_{mode}")
        for k in ("radii", "n_rendered", "ranges", "values_sorted", "n_contrib"):
            same = np.array_equal(np.asarray(g[k]).astype(np.int64).reshape(-1), np.asarray(r[k]).astype(np.int64).reshape(-1))
            print(f"   {k:14s} {'exact' if same else 'DIFFERENT'}")
            ok &= same or k == "n_contrib"
        for k in ("image", "accum_alpha"):
            bad = float((np.abs(g[k] - r[k]) > 1e-4).mean())
            print(f"   {k:14s} max |Δ| {np.abs(g[k] - r[k]).max():.3e}, fraction > 1e-4: {bad:.2e}")
            ok &= bad <= 1e-4
        for k in ("vmeans", "vshs", "vopacities", "vscales", "vrots", "vmeans2d"):
            e = rel_l2(r[k].reshape(-1), g[k].reshape(-1))
            print(f"   {k:14s} rel-L2 {e:.3e}")
            ok &= e <= 1e-4
    print("PINNED" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
