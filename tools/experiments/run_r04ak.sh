set -x
O=gpurun_out/r04ak; mkdir -p $O
for r in 1 2 3 4 5; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 500 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run$r', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['steady_state']['ms_per_step'])" >> $O/runs.txt
done
cat $O/runs.txt
