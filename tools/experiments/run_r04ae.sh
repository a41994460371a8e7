set -x
O=gpurun_out/r04ae; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest_forms.log 2>&1; echo "pytest forms rc=$?"; tail -5 $O/pytest_forms.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
  GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "direct" >> $O/ab.txt 2>&1
  $B 2>/dev/null | line "default" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v1b_replay.so GSR_PREPROCESS_AGG=512 $B 2>/dev/null | line "v1b_replay" >> $O/ab.txt 2>&1
done
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --no-loss"
for n in 100000 200000 300000 400000 600000; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --gaussians $n 2>/dev/null | line "n$n agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 1280 --height 720 2>/dev/null | line "720p agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --width 2560 --height 1440 --gaussians 2000000 2>/dev/null | line "1440p_2M agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --gaussians 3000000 2>/dev/null | line "1080p_3M agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_all.log
