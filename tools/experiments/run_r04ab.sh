set -x
O=gpurun_out/r04ab; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
for rep in 1 2; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg$m" >> $O/ab.txt 2>&1
done; done
for n in 200000 400000 600000 2000000; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 30 --warmup 5 --steady-steps 0 --gaussians $n --no-loss 2>/dev/null | line "n$n agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --mode rgbdn 2>/dev/null | line "rgbdn agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --width 1280 --height 720 --no-loss 2>/dev/null | line "720p agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
