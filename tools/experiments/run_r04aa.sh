set -x
O=gpurun_out/r04aa; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
for rep in 1 2; do
  GSR_PREPROCESS_AGG=0 python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 direct" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=1 GSR_AGG_WAVES=6 python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg6" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=1 GSR_AGG_WAVES=4 python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg4" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
