set -x
# third campaign of the round: the FINAL aggregating preprocess (flattened walks, 16-bit words on large grids) forced
export GSR_PREPROCESS_AGG=1
O=gpurun_out/r04_fuzz_flat; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1200 5000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 700 python tools/fuzz_parity.py deep 300 1500 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 700 python tools/fuzz_parity.py edge 500 3000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
for f in sweep deep edge; do tail -n 4 $O/$f.txt; done
