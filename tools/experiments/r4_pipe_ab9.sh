#!/bin/bash
# Round 4, pipelined backward: distance at which a hidden stage requests dZ_l from the hand-off ring (-DPIPE_ZD; the phases keep
# their 4 iterations).  shipped = 2; zd4 = the former single distance for both.
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  echo -n "shipped     "; tools/experiments/r4_train_line.sh 1 A=1
  for v in zd1 zd3 zd4 zd2lag1; do
    printf "%-12s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so
  done
done
