#!/bin/bash
# Round 4, pipelined backward: cache policy of the read-once HBM streams (phases, top dZ) -- do they evict the hand-off rings from
# the L2?  shipped = "nt"; sysnt = "sc0 sc1 nt", sc1nt = "sc1 nt", sc0nt = "sc0 nt", plain = no bits (-DPIPE_STREAM_BITS=...).
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  for v in sysnt sc1nt sc0nt plain; do printf "%-9s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so; done
done
