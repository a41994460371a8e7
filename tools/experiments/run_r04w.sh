set -x
O=gpurun_out/r04w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # tag, env...
  tag=$1; shift
  for kv in "$@"; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o b -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists --steps 30 --steady-steps 0 > $O/prof_$tag.log 2>&1
  python3 tools/short_kernel_stats.py $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/ks_$tag.csv
  rm -rf $O/prof_$tag
  echo "== $tag" >> $O/summary.txt; grep -E "preprocess|bin_inst|sh_color|tile_scan|sort_comp" $O/ks_$tag.csv | cut -d, -f1-4 >> $O/summary.txt
  for kv in "$@"; do unset ${kv%%=*}; done
}
run emit8 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8
run emit8_repl GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8 GSR_BIN_REPL=8192
run emit8_again GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8
run emit8_repl_again GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8 GSR_BIN_REPL=8192
cat $O/summary.txt
hipcc --offload-arch=gfx950 -O3 tools/atomic_rates.hip -o /tmp/atomic_rates && /tmp/atomic_rates > $O/atomic_rates.txt 2>&1; cat $O/atomic_rates.txt
