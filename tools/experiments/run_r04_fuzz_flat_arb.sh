set -x
O=gpurun_out/r04_fuzz_flat; mkdir -p $O
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate sweep 5124 5453 5664 5763  > $O/arb_sweep.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 1200 python tools/fuzz_parity.py arbitrate edge 3020 3027 3054 3065 3186 3266 3279 3296 3304 3351 3441  > $O/arb_edge.txt 2>&1
for c in 5124 5453 5664 5763 ; do GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py 1 $c 2>&1 | tail -2; done > $O/direct_sweep.txt 2>&1
for c in 3020 3027 3054 3065 3186 3266 3279 3296 3304 3351 3441 ; do GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py edge 1 $c 2>&1 | tail -2; done > $O/direct_edge.txt 2>&1
GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py deep 1 1716 2>&1 | tail -3 > $O/direct_deep.txt
tail -3 $O/arb_sweep.txt $O/arb_edge.txt | cut -c1-200; grep -c "^FAIL" $O/direct_sweep.txt $O/direct_edge.txt $O/direct_deep.txt
