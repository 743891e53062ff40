#!/bin/bash
# Kernel trace of the training step at the reference's batch size (3072 rays): which launches make up a step, their
# durations and the gaps between them (tools/step_trace.py).  Run on the GPU box through gpurun.
R=/root/repo
O=$R/gpurun_out/small_trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-two-pass --no-small-batch --no-half --no-exact --no-dt --batch 3072 --steps 20 --warmup 5 > $O/run.log 2>&1
python3 $R/tools/step_trace.py $O
