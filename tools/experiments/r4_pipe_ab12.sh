#!/bin/bash
# Round 4, pipelined backward with all pieces on the weight-gradient waves (shipped): prefetch depth of the data-gradient waves'
# B fragments (-DPIPE_PF, shipped 4), polls 1 ahead, dZ 3 ahead.
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  for v in pf2 pf6 pf8 w2lag1 w2zd3; do printf "%-9s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so; done
done
