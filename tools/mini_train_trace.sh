#!/bin/bash
# Kernel trace of one step of tools/mini_train.py (the Lightning-module path at the reference's batch size): launches, gaps.
R=/root/repo
O=$R/gpurun_out/mini_trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/mini_train.py 120 ${1:-256} > $O/run.log 2>&1
python3 $R/tools/step_trace.py $O
