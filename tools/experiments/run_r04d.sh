set -x
O=gpurun_out/r04d; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
run() { # tag env... -- args
  tag=$1; shift; E=""; while [ "$1" != "--" ]; do E="$E $1"; shift; done; shift
  env $E $B "$@" 2>>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$tag', '$*', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('composite_bwd','pergauss_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
}
for rep in 1 2; do
  run off GSR_BWD_TAIL=0 -- --mode rgb
  run tail1 GSR_BWD_TAIL=1 -- --mode rgb
  run tail2 GSR_BWD_TAIL=2 -- --mode rgb
  run off GSR_BWD_TAIL=0 -- --mode rgbd
  run tail1 GSR_BWD_TAIL=1 -- --mode rgbd
  run tail2 GSR_BWD_TAIL=2 -- --mode rgbd
done
run tail1_s5120 GSR_BWD_TAIL=1 GSR_BWD_SLOTS=5120 -- --mode rgb
run tail1_s5632 GSR_BWD_TAIL=1 GSR_BWD_SLOTS=5632 -- --mode rgb
run tail1_s6656 GSR_BWD_TAIL=1 GSR_BWD_SLOTS=6656 -- --mode rgb
run tail1_s7168 GSR_BWD_TAIL=1 GSR_BWD_SLOTS=7168 -- --mode rgb
run off GSR_BWD_TAIL=0 -- --mode rgbdn
run tail1 GSR_BWD_TAIL=1 -- --mode rgbdn
run off5 GSR_BWD_TAIL=0 -- --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --steps 10
run tail5 GSR_BWD_TAIL=1 -- --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --steps 10
run off2 GSR_BWD_TAIL=0 -- --gaussians 100000 --no-loss --seed 1002
run tail2c GSR_BWD_TAIL=1 -- --gaussians 100000 --no-loss --seed 1002
cat $O/ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
