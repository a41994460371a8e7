set -x
O=gpurun_out/r04ba; mkdir -p $O
export GSR_HIP_LIB=$PWD/tools/bin/libgsr_flatdirect.so
GSR_PREPROCESS_AGG=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed" $O/pytest.log
unset GSR_HIP_LIB
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steady-steps 0 --no-loss"
for rep in 1 2; do for lib in kept flat; do
  if [ $lib = flat ]; then export GSR_HIP_LIB=$PWD/tools/bin/libgsr_flatdirect.so; else unset GSR_HIP_LIB; fi
  GSR_PREPROCESS_AGG=0 $B2 --steps 10 --warmup 3 --gaussians 5000000 --width 3840 --height 2160 --seed 1005 2>/dev/null | line "cfg5 direct_$lib" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=0 $B2 --steps 40 --warmup 5 2>/dev/null | line "cfg3 direct_$lib" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=0 $B2 --steps 40 --warmup 5 --gaussians 50000 2>/dev/null | line "n50k direct_$lib" >> $O/ab.txt 2>&1
done; done
cat $O/ab.txt
