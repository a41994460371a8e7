set -x
O=gpurun_out/r04bb; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --no-loss"
for n in 100000 200000 300000 600000 1000000 3000000; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --gaussians $n 2>/dev/null | line "n$n agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 1280 --height 720 2>/dev/null | line "720p agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --width 2560 --height 1440 --gaussians 2000000 2>/dev/null | line "1440p_2M agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --order morton 2>/dev/null | line "morton agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --reference-lists 2>/dev/null | line "reflists agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_all.log
