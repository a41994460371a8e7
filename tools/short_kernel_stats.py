"""rocprofv3 --stats kernel_stats.csv with the kilobyte-long kernel names cut down to `name<template args>`:
    python tools/short_kernel_stats.py <in_kernel_stats.csv> <out.csv>"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void\s+)?([\w:]+)(<[^()]*?>)?\(", name)
    if m:
        return m.group(1).split("::")[-1] + (m.group(2) or "")
    return name.split("(")[0][-60:]


rows = list(csv.DictReader(open(sys.argv[1], newline="")))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        r["Name"] = short(r["Name"])
        w.writerow(r)
