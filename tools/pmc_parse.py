#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/pmc_workload.py into HBM bytes per launch.

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units that are uncalibrated on gfx950
(MI355X_MICROARCH.md §HBM: FETCH_SIZE reads exactly 1/2 of a wide coalesced stream), so
both are calibrated on the 1 GiB copy kernel of the same run: factor = known bytes /
counter value.  Output: {stage: bytes per launch} with the calibration factors recorded.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STAGES = {"preprocess_kernel": "preprocess", "tile_scan_kernel": "tile_scan", "scatter_kernel": "scatter",
          "tile_sort_kernel": "tile_sort", "composite_fwd_": "composite_fwd",
          "composite_bwd_kernel": "composite_bwd", "pergauss_bwd_kernel": "pergauss_bwd",
          "ssim_fwd_kernel": "loss_fwd", "ssim_bwd_kernel": "loss_bwd"}
CAL_BYTES = 256 * 1024 * 1024 * 4


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def main(d):
    out = {"unit": "bytes per launch", "calibration": {}}
    res = {}
    for counter, tag in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        files = glob.glob(os.path.join(d, "**", f"{tag}*counter_collection.csv"), recursive=True)
        if not files:
            print("missing", tag)
            continue
        acc = per_kernel(files[0], counter)
        # torch's Tensor.copy_ of the 1 GiB buffer runs as the runtime's blit kernel
        cal = [v for k, vs in acc.items() if "copyBuffer" in k for v in vs]
        cal = [v for v in cal if v > 0.5 * max(cal)] if cal else []
        if not cal:
            raise SystemExit(f"no calibration kernel found for {counter}")
        factor = CAL_BYTES / (sum(cal) / len(cal))
        out["calibration"][counter] = {"bytes_per_unit": factor, "copy_kernel_value": sum(cal) / len(cal)}
        for k, vs in acc.items():
            for key, stage in STAGES.items():
                if key in k:
                    # several kernel variants may belong to one stage (e.g. the two tile_sort launches):
                    # their per-launch means add up
                    r = res.setdefault(stage, {})
                    r[tag] = r.get(tag, 0.0) + factor * sum(vs) / len(vs)
    for stage, v in res.items():
        out[stage] = int(v.get("fetch", 0) + v.get("write", 0))
        out[stage + "_read"] = int(v.get("fetch", 0))
        out[stage + "_write"] = int(v.get("write", 0))
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    o = main(sys.argv[1])
    if len(sys.argv) > 2:
        json.dump(o, open(sys.argv[2], "w"), indent=1)
