#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/held; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/prof -o t -- python3 tools/experiments/r06_train_gaps.py 1290 > $O/log.txt 2>&1
tail -2 $O/log.txt | cut -c1-200
ls $O/prof/* | head
python tools/experiments/r06_held_latency.py $O/prof | tee $O/held_latency.txt
rm -rf $O/prof
