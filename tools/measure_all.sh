#!/bin/bash
# On the GPU box: everything the round's profiles/<round>/final directory is built from.   tools/measure_all.sh <tag>
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r06}
O=gpurun_out/$R
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
# PMC passes first: bench.py reads roofline.traffic / roofline.valu of exactly the measured configuration from them
tools/measure_pmc.sh ${R}_cfg3
tools/measure_pmc.sh ${R}_cfg2 --gaussians 100000 --no-loss --seed 1002
tools/measure_pmc.sh ${R}_cfg5 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --steps 3
# GSR_FORWARD_ONLY (round 4): the fused forward without the backward state; keyed ..._fwdonly
tools/measure_pmc.sh ${R}_cfg3_fwdonly --forward-only
# round 6 (round-5 verdict, next #5): what the reference actually trains — :rgbd (its default mode) at config-3 size, and the
# trained-like 1 M scene of extra_configs.scenes — so that their roofline.traffic / valu_frac stop being null
tools/measure_pmc.sh ${R}_cfg3_rgbd --mode rgbd
tools/measure_pmc.sh ${R}_trained_1m_rgbd --mode rgbd --scene trained --seed 1010
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# the driver's line (config 3 + extra_configs: config 2, config 5, :rgbd, trainer step both ways + cpu_baseline)
python bench.py > $O/bench_driver_line.json 2> $O/bench_driver_line.err
# the same command under the profiler: per-kernel average durations to set against the line's HIP-event times
# (--no-extra --no-other-lists: only config 3's kernels in the default list mode, so that the averages are the headline's)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists > $O/prof.log 2>&1
python3 tools/short_kernel_stats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*agent_info.csv" -delete
# two ranks on this one GPU, gloo carrying the collectives: the N > 1 code path of bench.py and its `exchange` object
# (NOT a scaling number: both ranks share the device)
GSR_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 2> $O/bench_2ranks.err | grep "^{" > $O/bench_2ranks_one_gpu_gloo.json  # (gloo prints a connection note to stdout)
# the same two ranks as the DRIVER starts them (torchrun): every worker is a GPU-free supervisor, one fresh rank group per exchange form
GSR_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 2> $O/bench_2ranks_torchrun.err | grep "^{" > $O/bench_2ranks_one_gpu_gloo_torchrun.json
# ... and the multi-GPU trainer step (gsr_sh_grad_from_views_tail after the factored exchange)
GSR_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --with-optimizer 2> $O/bench_2ranks_opt.err | grep "^{" > $O/bench_2ranks_one_gpu_gloo_optimizer.json
tail -2 $O/smoke.log; cut -c1-300 $O/bench_driver_line.json
