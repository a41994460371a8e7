set -x
O=gpurun_out/r04al; mkdir -p $O
P="import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], (d.get('steady_state') or {}).get('ms_per_step'))"
for r in 1 2 3; do
  python3 bench.py --steps 10 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 300 --mode rgbd 2>/dev/null | python -c "$P" rgbd10 >> $O/runs.txt
  python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 300 --mode rgbd 2>/dev/null | python -c "$P" rgbd20 >> $O/runs.txt
  python3 bench.py --steps 10 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 0 --with-optimizer 2>/dev/null | python -c "$P" opt10 >> $O/runs.txt
  python3 bench.py --steps 10 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 300 2>/dev/null | python -c "$P" rgb10 >> $O/runs.txt
done
cat $O/runs.txt
