set -x
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04a/pytest.log
tail -3 gpurun_out/r04a/pytest.log
timeout 600 python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?"
timeout 300 python tools/experiments/overlap_probe.py > gpurun_out/r04a/overlap.txt 2>&1; cat gpurun_out/r04a/overlap.txt | tail -2
