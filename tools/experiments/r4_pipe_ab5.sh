#!/bin/bash
# Round 4, pipelined backward on the phase stash: shipped (every data-gradient wave fetches the two phase fragments of its own tile)
# against phw (-DPIPE_PHASE_ON_WEIGHT=1: the weight-gradient waves fetch them, two each, between their matrix instructions).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  echo -n "phw      "; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_phw.so
done
