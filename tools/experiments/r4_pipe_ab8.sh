#!/bin/bash
# Round 4, pipelined backward, counters polled one iteration ahead: ring of 10 / 12 / 16 / 24 slots (two pipelines x 7 links x
# 16 slots x 16 KB = 3.6 MB of an XCD's 4 MB L2).
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for v in lag1 ring10lag1 ring12lag1 ring24lag1; do
    printf "%-12s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so
  done
done
