#!/bin/bash
# On the GPU box: SQ counters of one kernel for each library in GSR_AB_LIBS (and the default build).
#   GSR_AB_LIBS="tools/bin/a.so tools/bin/b.so" tools/experiments/sq_ab.sh <kernel substring> [pmc_workload flags ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-composite_bwd}; shift
for lib in default $GSR_AB_LIBS; do
  if [ "$lib" = default ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB="$PWD/$lib"; fi
  O=gpurun_out/sq_$(basename $lib .so); rm -rf $O; mkdir -p $O
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O -o p1 -- python3 tools/pmc_workload.py "$@" > $O/p1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O -o p2 -- python3 tools/pmc_workload.py "$@" > $O/p2.log 2>&1
  echo "== $lib"
  python3 - $O "$K" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]; print(f"   {k:24s} {sum(v)/len(v)/1e6:12.2f} M   ({len(v)} launches)")
PY
  find $O -name "*.csv" -delete
done
