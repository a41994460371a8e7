#!/usr/bin/env python3
"""On the GPU box: the randomised HIP-vs-oracle sweep of tests/test_gpu_parity.py over MANY more seeds than the test suite
runs (cases 12 .. 12+N: modes, SH degrees, ragged resolutions, views, footprint sizes, opacity ranges, both list modes;
forward, every gradient).  Prints the failing cases with their assertion; exit code = number of failures.

  python tools/fuzz_parity.py [N = 300] [first case = 12]
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsr_pkg  # noqa: E402

pkg = gsr_pkg.load()
from oracle import oracle as orc  # noqa: E402
import test_gpu_parity as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = []
for case in range(first, first + n):
    try:
        T.test_randomised_sweep_vs_oracle(pkg, orc, case)
    except Exception as e:  # noqa: BLE001
        tb = traceback.extract_tb(e.__traceback__)
        bad.append((case, f"{type(e).__name__}: {str(e)[:200]}", f"{tb[-1].filename.split('/')[-1]}:{tb[-1].lineno}"))
        print("FAIL case", case, bad[-1][1], "at", bad[-1][2], flush=True)
print(f"{n - len(bad)} / {n} cases passed (cases {first}..{first + n - 1})")
sys.exit(min(len(bad), 100))
