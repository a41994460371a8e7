set -x
O=gpurun_out/r04bj; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_prevodd.so" timeout 900 bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 > $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-120
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
$B 2>/dev/null | line "cfg3"; $B --gaussians 100000 --no-loss --seed 1002 2>/dev/null | line "cfg2"
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py tests/test_gpu_parity.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log
