#!/bin/bash
# Kernel-trace stats + HBM traffic counters of the two bench modes (run on the GPU box through gpurun).
# PMC passes are separate runs without any trace domain (gpurun refuses --pmc combined with traces).
set -e
R=/root/repo
OUT=$R/gpurun_out/prof_round
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in train fwd; do
  extra="--no-cpu-baseline --no-two-pass --no-small-batch --no-half --no-exact --steps 6 --warmup 2 --mode $mode"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${mode}_stats -- python3 $R/bench.py $extra > $OUT/${mode}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${mode}_fetch -- python3 $R/bench.py $extra > $OUT/${mode}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${mode}_write -- python3 $R/bench.py $extra > $OUT/${mode}_write.log 2>&1
  python3 $R/tools/hbm_traffic.py $R/gpurun_out/hbm_traffic.json $mode $([ $mode = train ] && echo 32768 || echo 1048576) 128 256 $OUT/${mode}_fetch $OUT/${mode}_write
  echo "$mode done"
done
# matrix-pipe busy / clock of the training step's kernels (SQ counters, their own pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/train_sq -- python3 $R/bench.py --no-cpu-baseline --no-two-pass --no-small-batch --no-half --no-exact --steps 6 --warmup 2 --mode train > $OUT/train_sq.log 2>&1 || echo "SQ pass failed"
