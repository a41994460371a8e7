#!/bin/bash
# on the GPU box: rocprofv3 --kernel-trace --stats of the whole training protocol (1 500 steps of the harness) -> per-kernel totals
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/protostats; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 tools/experiments/r06_train_gaps.py 1500 > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-200
python3 tools/short_kernel_stats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_protocol.csv
head -25 $O/kernel_stats_train_protocol.csv | cut -c1-160
rm -rf $O/prof
