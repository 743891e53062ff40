#!/bin/bash
# Round 4, training forward: cache policy of the stash stores (-DSUNERF_NT_STORE_AUX: 2 = nt shipped, 0 = none, 18 = sc1 nt,
# 19 = sc0 sc1 nt).  The forward is power-bound; 18 GB of phases leave it per step.
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  for v in st0 st18 st19; do printf "%-9s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so; done
done
