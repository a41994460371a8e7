set -x
O=gpurun_out/r04c; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
for v in default NO_BG0 SSIM_EXACT; do
  case $v in default) E="";; NO_BG0) E="GSR_NO_BG0=1";; SSIM_EXACT) E="GSR_SSIM_EXACT=1";; esac
  for mode in rgb rgbd rgbdn; do
    env $E $B --mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$v $mode rep$rep', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('loss_fwd','loss_bwd','composite_bwd','pergauss_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
