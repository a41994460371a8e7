set -x
O=gpurun_out/r04bg; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py tests/test_gpu_parity.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_forward_only.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "agg rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do $B 2>/dev/null | line "cfg3" >> $O/ab.txt 2>&1; done
B2="$B --no-loss"
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 1896 2>/dev/null | line "odd119 agg$m" >> $O/ab.txt 2>&1
done
$B2 --steps 10 --warmup 3 --gaussians 5000000 --width 3840 --height 2160 --seed 1005 2>/dev/null | line "cfg5" >> $O/ab.txt 2>&1
$B2 --gaussians 100000 --seed 1002 2>/dev/null | line "cfg2" >> $O/ab.txt 2>&1
cat $O/ab.txt
