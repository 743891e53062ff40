#!/bin/bash
# Round 4, pipelined backward: shipped (paired epilogue / decoder + MODE.FP16_OVFL) against noovfl (-DPIPE_FP16_OVFL=0: 16 v_med3 per
# chunk) and staged (-DPIPE_DECODE_STAGED=1: the decoder as five stages over all 16 phases instead of element pairs).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  for v in noovfl staged; do printf "%-9s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so; done
done
