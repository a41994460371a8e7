set -x
O=gpurun_out/r04n; mkdir -p $O
timeout 1500 python tools/fuzz_parity.py arbitrate edge 25 43 51 77 205 257 301 316 358 383 384 395 406 418 420 559 569 576 622 728 803 880 897 931 957 1001 1155 1220 1263 1323 1342 > $O/arbitrate_edge.txt 2>&1; echo "rc=$?" >> $O/arbitrate_edge.txt
grep "NOT EXPLAINED" $O/arbitrate_edge.txt | cut -c1-500; tail -2 $O/arbitrate_edge.txt
timeout 900 python -m pytest tests/test_gpu_fuzz_regressions.py -q -m gpu > $O/pytest_fuzz.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fuzz.log
tail -3 $O/pytest_fuzz.log
