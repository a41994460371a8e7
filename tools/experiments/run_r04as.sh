set -x
O=gpurun_out/r04as; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do
  $B 2>/dev/null | line "fused" >> $O/ab.txt 2>&1
  GSR_SPLIT_SH=1 $B 2>/dev/null | line "split_sh" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
GSR_SPLIT_SH=1 GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GSR_SPLIT_SH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists --steps 20 --steady-steps 0 > $O/prof.log 2>&1
python3 tools/short_kernel_stats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_split.csv
rm -rf $O/prof
head -6 $O/kernel_stats_split.csv
