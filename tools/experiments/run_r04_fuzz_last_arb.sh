set -x
O=gpurun_out/r04_fuzz_last; mkdir -p $O
timeout 600 python tools/fuzz_parity.py arbitrate sweep 9647 > $O/arb_sweep.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 600 python tools/fuzz_parity.py arbitrate sweep 10071 10339 10494 >> $O/arb_sweep.txt 2>&1
timeout 600 python tools/fuzz_parity.py arbitrate edge 5107 5133 > $O/arb_edge.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate edge 5315 5321 5378 5381 5457 5463 5464 5502 5545 >> $O/arb_edge.txt 2>&1
grep -E "^sweep|^edge|^[0-9]+ / " $O/arb_sweep.txt $O/arb_edge.txt | cut -c1-220
