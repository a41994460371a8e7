#!/bin/bash
# On the GPU box: PMC passes (HBM bytes + SQ counters per launch) for ONE bench configuration, merged into
# profiles/pmc_traffic.json under its config key.   tools/measure_pmc.sh <tag> [pmc_workload flags ...]
# (counter passes are separate runs with --kernel-trace only, as the pool requires)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o fetch -- python3 tools/pmc_workload.py "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o write -- python3 tools/pmc_workload.py "$@" > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT" -o sq_pass1 -- python3 tools/pmc_workload.py "$@" > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT" -o sq_pass2 -- python3 tools/pmc_workload.py "$@" > "$OUT/sq2.log" 2>&1
mkdir -p gpurun_out/profiles_out
cp profiles/pmc_traffic.json gpurun_out/profiles_out/pmc_traffic.json 2>/dev/null
python3 tools/pmc_parse.py "$OUT" gpurun_out/profiles_out/pmc_traffic.json --source "$TAG" > "$OUT/parse.log" 2>&1
cp gpurun_out/profiles_out/pmc_traffic.json profiles/pmc_traffic.json
tail -12 "$OUT/parse.log"
# the raw counter CSVs are ~MBs of kilobyte-long kernel names: keep only our kernels' rows
for f in "$OUT"/*/*counter_collection.csv "$OUT"/*counter_collection.csv; do
  [ -f "$f" ] && python3 - "$f" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p, newline="")))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("_kernel", "composite_")) and "at::native" not in r["Kernel_Name"]]
if rows:
    w = csv.DictWriter(open(p, "w", newline=""), fieldnames=list(rows[0].keys()))
    w.writeheader(); w.writerows(keep)
PY
done
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
