cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gap; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-other-lists --steps 30 > $O/bench.json 2>$O/bench.err
cut -c1-200 $O/bench.json
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o bench -- python3 bench.py --in-process --no-cpu-baseline --no-other-lists --steps 20 --warmup 3 > $O/prof.log 2>&1
python tools/gap_report.py $(find $O/prof -name "*kernel_trace.csv" | head -1) | tee $O/gaps.txt
