#!/bin/bash
# Round 4, pipelined backward: the hand-off protocol on one weight-gradient wave (shipped: wave 4 publishes, polls and gates)
# against split over two (splitp = -DPIPE_SPLIT_PROTOCOL=1: wave 4 publishes, wave 5 polls and gates).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  printf "%-9s" splitp; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_splitp.so
done
