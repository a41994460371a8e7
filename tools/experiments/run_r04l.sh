set -x
O=gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_scale.py -x -q -m gpu -k "ssim or loss or golden or config3_full_step or rgbd_1m" > $O/pytest_ssim.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ssim.log
tail -4 $O/pytest_ssim.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
for v in tile32 tile16; do
  case $v in tile32) E="GSR_X=1";; tile16) E="GSR_SSIM_TILE16=1";; esac
  for mode in rgb rgbd; do
    env $E $B --mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$v $mode rep$rep', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('loss_fwd','loss_bwd','composite_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
