set -x
# fourth campaign of the round: the FINAL build in its DEFAULT configuration (the fuzz scenes are small: direct form of preprocess,
# now with the flattened walk; grids of odd width take the per-lane walk)
O=gpurun_out/r04_fuzz_final; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1200 7000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 700 python tools/fuzz_parity.py deep 300 2000 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 700 python tools/fuzz_parity.py edge 500 4000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
for f in sweep deep edge; do tail -n 4 $O/$f.txt; done
