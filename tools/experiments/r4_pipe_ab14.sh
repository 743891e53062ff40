#!/bin/bash
# Round 4, pipelined backward: the data-gradient waves' accumulator tile (and the accumulators of the in-layer stage and of the
# prologue) in architectural VGPRs (vform = -mllvm -amdgpu-mfma-vgpr-form=1 for bwd_pipe.hip: no v_accvgpr_read in front of the
# epilogue) against AGPRs (shipped); daccv = -DPIPE_DACC_VGPR=1: only the data-gradient waves' tile, through asm matrix instructions with a VGPR destination.
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo -n "shipped  "; tools/experiments/r4_train_line.sh 1 A=1
  for v in vform daccv; do [ -f build_var/libsunerf_hip_$v.so ] && { printf "%-9s" $v; tools/experiments/r4_train_line.sh 1 SUNERF_HIP_LIB=$PWD/build_var/libsunerf_hip_$v.so; }; done
done
