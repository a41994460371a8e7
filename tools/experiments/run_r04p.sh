set -x
O=gpurun_out/r04p; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do
for v in base bg0; do
  case $v in base) E="GSR_X=1";; bg0) E="GSR_BG0_RGB=1";; esac
  env $E $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$v rep$rep', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('composite_bwd','pergauss_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
done
done
cat $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -m gpu -k "rgbd or depth_and_normal or forward_backward_vs_oracle or long_lists" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
