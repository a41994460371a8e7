#!/usr/bin/env python3
"""Timeline of the LAST step of a `rocprofv3 --kernel-trace --output-format csv` run: every kernel's start / end relative to the
step's first kernel (preprocess), with its queue — shows which launches really overlapped.
  tools/experiments/timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import re
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"] and "Lb1ELb" not in r["Kernel_Name"].split("preprocess_kernel")[1][:24]]
# the last preprocess launch that is not a SCATTER instantiation: template args <DEG, NT, W32, SCATTER, BANDED>
def is_scatter(name):
    m = re.search(r"preprocess_kernel<\s*\d+\s*,\s*\d+\s*,\s*(true|false)\s*,\s*(true|false)", name)
    return bool(m and m.group(2) == "true")
starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"] and not is_scatter(r["Kernel_Name"])]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{a:9.1f} {b:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {name}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
