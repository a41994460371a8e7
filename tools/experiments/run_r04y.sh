set -x
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
for rep in 1 2; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 2>/dev/null | line "cfg5 agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 --gaussians 100000 --no-loss --seed 1002 2>/dev/null | line "cfg2 agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --mode rgbd 2>/dev/null | line "rgbd agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
GSR_PREPROCESS_AGG=1 timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_all_agg.log
