#!/bin/bash
# Round 4, pipelined backward: fewer vector instructions on the data-gradient waves (the workgroup's critical path).
# shipped = epilogue / decoder on explicit register pairs (-DPIPE_TRIM=1: no v_mov for hipcc's own pairing, phase scaling as 8 packed
# multiplies); notrim = -DPIPE_TRIM=0; ovfl = trim + MODE.FP16_OVFL instead of 16 v_med3 per chunk (-DPIPE_FP16_OVFL=1).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  for v in notrim ovfl; do printf "%-9s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so; done
done
