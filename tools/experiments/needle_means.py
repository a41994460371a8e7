#!/usr/bin/env python3
"""Per-tensor distance of HIP, of the fp32 oracle (double per-pixel sums) and — round 6 — of the REFERENCE'S OWN FORM (fp32 atomics
per pixel, render.jl:242,262-282: orc.backward(deterministic=False), best .. worst of five arrival orders) from the float64 model on
hostile (needle) scenes:
  python tools/experiments/needle_means.py edge 8498 5315 ...     (GSR_HIP_LIB selects the library build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
args, sys.argv = sys.argv[1:], sys.argv[:1]
import fuzz_parity as F  # noqa: E402
import numpy as np  # noqa: E402
import test_gpu_fuzz_regressions as R  # noqa: E402
import fuzz_scenes  # noqa: E402

build = {"sweep": fuzz_scenes.sweep_scene, "edge": fuzz_scenes.edge_scene, "deep": fuzz_scenes.deep_scene}[args[0]]
for c in args[1:]:
    fs = build(F.pkg, int(c))
    res, st = R.three_way(F.pkg, F.orc, fs)
    ref = R.reference_form_distance(F.orc, build(F.pkg, int(c)), st, res)
    vis = st.radii > 0
    row = []
    for nm, (o, h, f) in res.items():
        row.append(f"{nm} hip {R._rel(h[vis], f[vis]):.1e} orc {R._rel(o[vis], f[vis]):.1e} atomics {ref[nm][0]:.1e}..{ref[nm][1]:.1e}")
    print(args[0], c, " | ".join(row), flush=True)
