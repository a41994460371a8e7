set -x
# last campaign of the round: the final build (flattened walks on grids of odd width too, direct form capped at four waves), default
# configuration, and once more with the aggregating form forced; fresh case ranges
O=gpurun_out/r04_fuzz_last; mkdir -p $O
timeout 700 python tools/fuzz_parity.py 800 9000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 500 python tools/fuzz_parity.py deep 150 2500 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 500 python tools/fuzz_parity.py edge 300 5000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
GSR_PREPROCESS_AGG=1 timeout 700 python tools/fuzz_parity.py 800 9800 > $O/sweep_agg.txt 2>&1; echo "rc=$?" >> $O/sweep_agg.txt
GSR_PREPROCESS_AGG=1 timeout 500 python tools/fuzz_parity.py edge 300 5300 > $O/edge_agg.txt 2>&1; echo "rc=$?" >> $O/edge_agg.txt
for f in sweep deep edge sweep_agg edge_agg; do grep -E "^FAIL|cases passed" $O/$f.txt | awk '{ if ($1=="FAIL") printf "%s:%s ", $3, $NF; else print }'; echo; done
