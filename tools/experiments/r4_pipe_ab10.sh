#!/bin/bash
# Round 4, pipelined backward: where the data-gradient waves issue their six DMA pieces.  shipped = between the vector work of
# the epilogue and the decoder; early (-DPIPE_PIECES_LATE=0) = between the matrix instructions of the k-steps.
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped     "; tools/experiments/r4_train_line.sh 1 A=1
  echo -n "early       "; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_early.so
done
