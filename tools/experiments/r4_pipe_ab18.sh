#!/bin/bash
# Round 4, final pipelined backward (pieces on the weight-gradient waves, both roles balanced): static priority again --
# prio1 = s_setprio 1 on the data-gradient waves, prio2 = on the weight-gradient waves (-DPIPE_PRIO).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  for v in prio1 prio2; do printf "%-9s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so; done
done
