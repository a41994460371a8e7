#!/bin/bash
# On the GPU box: everything the round's profiles/ directory is built from.   tools/measure_all.sh <tag>
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --gaussians 100000 --no-loss --seed 1002 > $O/bench_cfg2.json 2>/dev/null
python bench.py --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --steps 10 > $O/bench_cfg5.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-lists --with-optimizer > $O/bench_optimizer.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-lists --mode rgbd > $O/bench_rgbd.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --no-cpu-baseline --no-other-lists --steps 20 --warmup 3 > $O/prof.log 2>&1
tools/measure_pmc.sh ${R}_cfg3
tools/measure_pmc.sh ${R}_cfg2 --gaussians 100000 --no-loss --seed 1002
tools/measure_pmc.sh ${R}_cfg5 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --steps 3
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# the driver's line again, now that the PMC file of this build exists
python bench.py --no-cpu-baseline > $O/bench_cfg3_with_pmc.json 2>/dev/null
tail -2 $O/smoke.log; cut -c1-300 $O/bench_cfg3.json
