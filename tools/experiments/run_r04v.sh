set -x
O=gpurun_out/r04v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # tag, env...
  tag=$1; shift
  for kv in "$@"; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o b -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists --steps 30 --steady-steps 0 > $O/prof_$tag.log 2>&1
  python3 tools/short_kernel_stats.py $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/ks_$tag.csv
  rm -rf $O/prof_$tag
  echo "== $tag" >> $O/summary.txt; grep -E "preprocess|bin_inst|sh_color|tile_scan" $O/ks_$tag.csv | cut -d, -f1-4 >> $O/summary.txt
  grep '^{' $O/prof_$tag.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['roofline']['stages_ms'])" >> $O/summary.txt
  for kv in "$@"; do unset ${kv%%=*}; done
}
run base
run emit8 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8
run emit4 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=4
run emit2 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=2
run emit4_sh2 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=4 GSR_SPLIT_SH=2
cat $O/summary.txt
GSR_SPLIT_EMIT=1 GSR_BIN_PEND=4 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_emit4.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_emit4.log
