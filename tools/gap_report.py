"""Idle time between consecutive kernels of a step, from a rocprofv3 --kernel-trace CSV:
    python tools/gap_report.py <..._kernel_trace.csv>
A step starts at preprocess_kernel; the last 15 steps of the trace are averaged."""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for d in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", d["Kernel_Name"]).split("<")[0].split("(")[0]
        rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), name.split()[-1][-36:]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[2]]
gaps, steps = collections.OrderedDict(), []
for a, b in zip(idx[-16:-1], idx[-15:]):
    seq = rows[a:b]
    steps.append((rows[b][0] - seq[0][0], sum(e - s for s, e, _ in seq)))
    for (s0, e0, n0), (s1, e1, n1) in zip(seq, seq[1:] + [rows[b]]):
        gaps.setdefault((n0, n1), []).append(s1 - e0)
for (n0, n1), g in gaps.items():
    print(f"{n0:>36s} -> {n1:<36s} gap {sum(g) / len(g) / 1000:7.2f} us")
print("step span %.1f us, kernels busy %.1f us, idle %.1f us" % (
    sum(s for s, _ in steps) / len(steps) / 1000, sum(b for _, b in steps) / len(steps) / 1000,
    sum(s - b for s, b in steps) / len(steps) / 1000))
