set -x
# the round's second campaign: the aggregating form of preprocess FORCED (small scenes take the direct form by default)
export GSR_PREPROCESS_AGG=1
O=gpurun_out/r04_fuzz_agg; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1200 3000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 700 python tools/fuzz_parity.py deep 300 1000 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 700 python tools/fuzz_parity.py edge 500 2000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
tail -3 $O/sweep.txt $O/deep.txt $O/edge.txt
