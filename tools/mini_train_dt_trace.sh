#!/bin/bash
# Kernel trace of one step of tools/mini_train_dt.py (density-temperature module path): launches, gaps.
R=/root/repo
O=$R/gpurun_out/mini_trace_dt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/mini_train_dt.py 60 > $O/run.log 2>&1
tail -4 $O/run.log
python3 $R/tools/step_trace.py $O
