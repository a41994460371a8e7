#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/pmc_workload.py into per-launch HBM bytes and SQ counters
of ONE configuration and merge them into profiles/pmc_traffic.json under configs[<key>].

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units that are uncalibrated on gfx950
(MI355X_MICROARCH.md §HBM: FETCH_SIZE reads exactly 1/2 of a wide coalesced stream), so
both are calibrated on the 1 GiB copy kernel of the same pass: factor = known bytes /
counter value.  SQ counters (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, ...) are taken as
reported (per-launch mean).

  tools/pmc_parse.py OUT_DIR profiles/pmc_traffic.json [--meta gpurun_out/pmc_meta.json] [--source r02_a]
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

KERNEL_SOURCES = ("composite.hip", "pergauss.hip", "binning.hip", "ssim.hip", "wave_reduce.h", "tile_sort_device.h", "tile_mask.h",
                  "gsr_kernels.h", "agg_plan.h", "gsr_api.cpp", "gsr_policy.cpp")  # == bench.py PMC_KERNEL_SOURCES


def kernel_source_hashes():
    """git blob hashes of the kernel sources the counters were collected on (bench.py: PMC_KERNEL_SOURCES, pmc_stale)."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussiansplatting.jl_amd", "csrc")
    out = {}
    for f in KERNEL_SOURCES:
        data = open(os.path.join(d, f), "rb").read()
        out[f] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return out


STAGES = {"preprocess_kernel": "preprocess", "tile_scan_kernel": "tile_scan", "sort_composite_fwd_kernel": "sort_composite_fwd", "tile_sort_kernel": "tile_sort", "tile_sort_wave_kernel": "tile_sort",
          "tile_count_kernel": "tile_sort", "tile_scatter_kernel": "tile_sort", "tile_radix": "tile_sort",
          "composite_fwd_": "composite_fwd", "composite_bwd_kernel": "composite_bwd",
          "pergauss_bwd_kernel": "pergauss_bwd", "ssim_fwd_kernel": "loss_fwd", "ssim_bwd_kernel": "loss_bwd"}
CAL_BYTES = 256 * 1024 * 1024 * 4


def per_kernel(path):
    """{counter: {kernel name: [values]}} — streaming parse (kernel names are kilobytes long)."""
    acc = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            acc[row["Counter_Name"]][row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def stage_of(kernel_name):
    for key, stage in STAGES.items():
        if key in kernel_name:
            return stage
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("json_path")
    ap.add_argument("--meta", default=None)
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    meta_path = a.meta or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pmc_meta.json")
    meta = json.load(open(meta_path))
    rec = {"source": a.source, "tile_instances": meta["tile_instances"], "n_visible": meta.get("n_visible"),
           "unit": "per launch (mean over the launches of the pass)", "calibration": {}, "hbm_bytes": {}, "hbm_read": {},
           "hbm_write": {}, "sq": {}, "kernel_sources": kernel_source_hashes()}
    for f in sorted(glob.glob(os.path.join(a.out_dir, "**", "*counter_collection.csv"), recursive=True)):
        acc = per_kernel(f)
        for counter, by_kernel in acc.items():
            if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                # torch's Tensor.copy_ of the 1 GiB buffer runs as an elementwise copy kernel
                cal = [v for k, vs in by_kernel.items() if ("copyBuffer" in k or "direct_copy" in k or "CopyFunctor" in k) for v in vs]
                cal = [v for v in cal if v > 0.5 * max(cal)] if cal else []
                if not cal:
                    raise SystemExit(f"no calibration kernel found for {counter} in {f}")
                factor = CAL_BYTES / (sum(cal) / len(cal))
                rec["calibration"][counter] = {"bytes_per_unit": factor, "copy_kernel_value": sum(cal) / len(cal)}
                dst = rec["hbm_read"] if counter == "FETCH_SIZE" else rec["hbm_write"]
                for k, vs in by_kernel.items():
                    st = stage_of(k)
                    if st:  # several kernel variants may belong to one stage: their per-launch means add up
                        dst[st] = dst.get(st, 0) + int(factor * sum(vs) / len(vs))
            else:
                for k, vs in by_kernel.items():
                    st = stage_of(k)
                    if st:
                        d = rec["sq"].setdefault(st, {})
                        d[counter] = d.get(counter, 0) + sum(vs) / len(vs)
    for st in set(rec["hbm_read"]) | set(rec["hbm_write"]):
        rec["hbm_bytes"][st] = rec["hbm_read"].get(st, 0) + rec["hbm_write"].get(st, 0)
    doc = {"configs": {}}
    if os.path.exists(a.json_path):
        try:
            old = json.load(open(a.json_path))
            if "configs" in old:
                doc = old
        except Exception:
            pass
    doc["configs"][meta["key"]] = rec
    json.dump(doc, open(a.json_path, "w"), indent=1, sort_keys=True)
    print(meta["key"], json.dumps({k: rec[k] for k in ("tile_instances", "hbm_bytes")}, indent=1))
    for st, d in rec["sq"].items():
        print(st, {k: int(v) for k, v in d.items()})


if __name__ == "__main__":
    main()
