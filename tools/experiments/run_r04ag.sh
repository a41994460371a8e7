set -x
O=gpurun_out/r04ag; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest_forms.log 2>&1; echo "pytest forms rc=$?"; tail -2 $O/pytest_forms.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do
  $B 2>/dev/null | line "scan_under_p2" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_kept.so $B 2>/dev/null | line "kept" >> $O/ab.txt 2>&1
done
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --no-loss"
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 2560 --height 1440 --gaussians 2000000 2>/dev/null | line "1440p_2M agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
