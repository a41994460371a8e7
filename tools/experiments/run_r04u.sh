set -x
O=gpurun_out/r04u; mkdir -p $O
python tools/experiments/overlap_probe_fwd.py > $O/probe.txt 2>&1
python tools/experiments/overlap_probe_fwd.py 33554432 >> $O/probe.txt 2>&1
for rep in 1 2; do for m in 0 1 2; do
  GSR_SPLIT_SH=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('split$m', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))" >> $O/ab.txt 2>&1
done; done
for m in 0 2; do
  GSR_SPLIT_SH=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('cfg5 split$m', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))" >> $O/ab.txt 2>&1
  GSR_SPLIT_SH=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 --gaussians 100000 --no-loss --seed 1002 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('cfg2 split$m', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))" >> $O/ab.txt 2>&1
done
cat $O/probe.txt $O/ab.txt
GSR_SPLIT_SH=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_split2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_split2.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GSR_SPLIT_SH=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists --steps 20 --steady-steps 0 > $O/prof.log 2>&1
python3 tools/short_kernel_stats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_split2.csv
find $O/prof -name "*.csv" -size +1M -delete
cat $O/kernel_stats_split2.csv | head -12
