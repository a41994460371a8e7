set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_trainer.py -x -q 2>&1 | tail -15
timeout 300 python bench.py --with-optimizer --no-cpu-baseline --steps 30 > gpurun_out/opt_tail.json 2> gpurun_out/opt_tail.err
timeout 300 python bench.py --with-optimizer --tail-in-backward --no-cpu-baseline --steps 30 > gpurun_out/opt_fused.json 2> gpurun_out/opt_fused.err
tail -c 1500 gpurun_out/opt_tail.json; tail -c 1500 gpurun_out/opt_fused.json; tail -5 gpurun_out/opt_fused.err
