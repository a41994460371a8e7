#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of bench.py (headline config, no extras) for the default library and
# each library in GSR_AB_LIBS; prints the average duration of our kernels.   tools/kernel_times.sh [bench args ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in default $GSR_AB_LIBS; do
  if [ "$lib" = default ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB="$PWD/$lib"; fi
  tag=$(basename "$lib" .so)
  out=gpurun_out/ktimes_$tag
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o st -- python3 bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists "$@" > "$out/bench.json" 2> "$out/err.log"
  f=$(find "$out" -name "*kernel_stats.csv" | head -1)
  echo "== $lib"
  python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
for r in rows:
    n = r["Name"]
    m = re.search(r"(preprocess_kernel|tile_scan_kernel|sort_composite_fwd_kernel|composite_bwd_kernel|pergauss_bwd_kernel|ssim_fwd_kernel|ssim_bwd_kernel|tile_sort\w*|composite_fwd_strip_kernel)", n)
    if m:
        print(f"   {m.group(1):28s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}")
PY
  find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*agent_info.csv" -delete
done
