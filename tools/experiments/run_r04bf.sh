set -x
O=gpurun_out/r04bf; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do $B 2>/dev/null | line "final" >> $O/ab.txt 2>&1; done
$B --gaussians 100000 --no-loss --seed 1002 2>/dev/null | line "cfg2" >> $O/ab.txt 2>&1
cat $O/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_all.log
