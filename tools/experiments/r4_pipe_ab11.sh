#!/bin/bash
# Round 4, pipelined backward (hand-off spins gone, data-gradient waves the long ones): who issues the LDS-DMA pieces.
# shipped = mode 0 (data-gradient waves); dmaw1 = -DPIPE_DMA_ON_WEIGHT=1 (phase fragments by the weight-gradient waves);
# dmaw2 = 2 (all 24 pieces by the weight-gradient waves, one behind each of their first six pairs of matrix instructions).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  for v in dmaw1 dmaw2; do printf "%-9s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so; done
done
