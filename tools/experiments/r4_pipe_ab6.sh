#!/bin/bash
# Round 4, pipelined backward on the phase stash: where the phases are decoded.  shipped = split (the data-gradient waves form the
# cosines behind their epilogue, the weight-gradient waves 5 - 7 the sines right behind the barrier, while the data waves issue
# matrix instructions) against nosplit (-DPIPE_SPLIT_DECODE=0: the data-gradient waves decode both behind their epilogue).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "split    "; tools/experiments/r4_train_line.sh 1 A=1
  echo -n "nosplit  "; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_nosplit.so
done
