#!/bin/bash
# on the GPU box: kernel traces of the training protocol at two model sizes -> idle time between the kernels of a step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/gaps; rm -rf $O; mkdir -p $O
for steps in 660 1290; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof$steps -o t -- python3 tools/experiments/r06_train_gaps.py $steps > $O/log$steps.txt 2>&1
  tail -1 $O/log$steps.txt | cut -c1-300
  python tools/gap_report.py $(find $O/prof$steps -name "*kernel_trace.csv" | head -1) | tee $O/gaps$steps.txt
  rm -rf $O/prof$steps
done
