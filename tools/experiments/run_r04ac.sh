set -x
O=gpurun_out/r04ac; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
  GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "direct" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=1 $B 2>/dev/null | line "v2" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v2_norep.so GSR_PREPROCESS_AGG=1 $B 2>/dev/null | line "v2_norep" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v1b.so GSR_PREPROCESS_AGG=512 $B 2>/dev/null | line "v1b_512" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v1b.so GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "v1b_direct" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
