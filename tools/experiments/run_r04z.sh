set -x
O=gpurun_out/r04z2; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
for rep in 1 2; do for m in 0 256 512 1024; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 512; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 --gaussians 100000 --no-loss --seed 1002 2>/dev/null | line "cfg2 agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
