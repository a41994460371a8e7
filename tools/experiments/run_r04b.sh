set -x
O=gpurun_out/r04b; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python tools/experiments/overlap_probe.py > $O/overlap.txt 2>&1; tail -2 $O/overlap.txt
