set -x
O=gpurun_out/r04k; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/measure_all.sh r04 > $O/measure_all.log 2>&1
tail -3 $O/measure_all.log
