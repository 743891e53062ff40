#!/bin/bash
# Kernel-trace stats of the DEFAULT bench command (headline + small batch + two-pass + density-temperature sub-lines) and the
# SQ counters of the three big kernels; run on the GPU box through gpurun.  Outputs under gpurun_out/prof_default/.
R=/root/repo
O=$R/gpurun_out/prof_default
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/stats.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
head -14 $O/kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
ex="--no-cpu-baseline --no-two-pass --no-small-batch --no-half --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $O/sq_train -- python3 $R/bench.py $ex > $O/sq_train.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $O/sq_fwd -- python3 $R/bench.py $ex --mode fwd > $O/sq_fwd.log 2>&1
python3 $R/tools/pmc_summary.py $O/sq_train > $O/pmc_sq_train.txt
python3 $R/tools/pmc_summary.py $O/sq_fwd render_fwd > $O/pmc_sq_fwd.txt
echo done
