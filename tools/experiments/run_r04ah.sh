set -x
O=gpurun_out/r04ah; mkdir -p $O
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate sweep 3127 3285 3648 3661 3808 > $O/arb_sweep.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate edge 2003 2157 2223 2356 2466 > $O/arb_edge.txt 2>&1
# the same cases in the direct form: the gradients are bit-identical between the forms, so must be the verdicts
for c in 3127 3285 3648 3661 3808; do GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py 1 $c 2>&1 | tail -2; done > $O/direct_sweep.txt 2>&1
for c in 2003 2157 2223 2356 2466; do GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py edge 1 $c 2>&1 | tail -2; done > $O/direct_edge.txt 2>&1
GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py deep 1 1299 2>&1 | tail -3 > $O/direct_deep.txt
tail -30 $O/arb_sweep.txt; tail -30 $O/arb_edge.txt; cat $O/direct_sweep.txt $O/direct_edge.txt $O/direct_deep.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], d['config']['tile_instances'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0"
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B --reference-lists 2>/dev/null | line "reflists agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B --order morton 2>/dev/null | line "morton agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B 2>/dev/null | line "cfg3 agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
