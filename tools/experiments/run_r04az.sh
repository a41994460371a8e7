set -x
O=gpurun_out/r04az; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest_forms.log 2>&1; echo "forms rc=$?"; tail -3 $O/pytest_forms.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steady-steps 0 --no-loss"
for rep in 1 2; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --steps 10 --warmup 3 --gaussians 5000000 --width 3840 --height 2160 --seed 1005 2>/dev/null | line "cfg5 agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --steps 30 --warmup 5 --width 2560 --height 1440 --gaussians 2000000 2>/dev/null | line "1440p_2M agg$m" >> $O/ab.txt 2>&1
done; done
cat $O/ab.txt
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_scale.py -x -q > $O/pytest_scale.log 2>&1; echo "scale rc=$?"; grep -E "passed|failed" $O/pytest_scale.log
