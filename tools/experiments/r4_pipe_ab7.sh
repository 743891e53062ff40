#!/bin/bash
# Round 4, pipelined backward: slack of the hand-off.  Every stage spends ~15 % of the launch in single-round-trip spins (the
# counters it polled 4 iterations earlier do not yet allow what the partner has long done).  Variants: ring of 32 / 64 slots
# (-DPIPE_RING), counters polled 2 / 1 iterations ahead (-DPIPE_POLL_LAG).
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  echo -n "shipped     "; tools/experiments/r4_train_line.sh 1 A=1
  for v in ring32 ring64 lag2 lag1 ring32lag2; do
    printf "%-12s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so
  done
done
