#!/bin/bash
# Round 4, pipelined backward, all pieces on the weight-gradient waves: 6 each (shipped, mode 2) against 3 / 7 / 7 / 7 with the
# protocol wave relieved (dmaw3 = -DPIPE_DMA_ON_WEIGHT=3).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  printf "%-9s" dmaw3; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_dmaw3.so
done
