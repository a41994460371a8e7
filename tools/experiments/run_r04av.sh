set -x
O=gpurun_out/r04av; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest_forms.log 2>&1; echo "forms rc=$?"; tail -3 $O/pytest_forms.log
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do
  $B 2>/dev/null | line "flat" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "direct" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
