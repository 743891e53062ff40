#!/bin/bash
# Builds libsunerf_hip.so (gfx950) in-tree.  Usage: ./build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=${SUNERF_BUILD_OUT:-../libsunerf_hip.so}     # variants for A/B runs: SUNERF_BUILD_OUT=... SUNERF_BUILD_OBJ=... ./build.sh -DFLAG
OBJ=${SUNERF_BUILD_OBJ:-../build_obj}
mkdir -p "$OBJ"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
# -amdgpu-mfma-vgpr-form: MFMA accumulators in architectural VGPRs.  The render / dgrad kernels keep their activation
# fragments in the AGPR half of the register file (see render_fwd.hip), so this removes a v_accvgpr_read per accumulator
# element from every tile epilogue.  wgrad.hip holds 256 accumulator registers per lane and wants them in AGPRs.
pids=()
for f in pack sampler rays render_fwd render_bwd dt train_step bwd_exact; do
  hipcc $COMMON -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c -o "$OBJ/$f.o" "$f.hip" &
  pids+=($!)
done
for f in wgrad bwd_pipe; do     # accumulators in AGPRs (bwd_pipe.hip: 128 per weight-gradient wave, W^T fragments per data-gradient wave)
  hipcc $COMMON "$@" -c -o "$OBJ/$f.o" "$f.hip" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done     # set -e: a failed compile fails the build
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/pack.o "$OBJ"/sampler.o "$OBJ"/rays.o "$OBJ"/render_fwd.o "$OBJ"/render_bwd.o "$OBJ"/dt.o "$OBJ"/train_step.o "$OBJ"/wgrad.o "$OBJ"/bwd_pipe.o "$OBJ"/bwd_exact.o
echo "built $OUT"
