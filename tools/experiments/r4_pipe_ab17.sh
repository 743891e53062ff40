#!/bin/bash
# Round 4, pipelined backward: the data-gradient waves' 16 matrix instructions per chunk as ONE dependent chain (shipped) against
# two (chains2 = -DPIPE_TWO_CHAINS=1: even / odd k-steps on two accumulator tiles, 16 adds behind the loop).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  printf "%-9s" chains2; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_chains2.so
done
