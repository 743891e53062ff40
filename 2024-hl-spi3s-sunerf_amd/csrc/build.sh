#!/bin/bash
# Builds libsunerf_hip.so (gfx950) in-tree.  Usage: ./build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libsunerf_hip.so
# -amdgpu-mfma-vgpr-form: MFMA accumulators in architectural VGPRs (the AGPR half holds the activation fragments,
# see render_fwd.hip), which removes a v_accvgpr_read per accumulator element from every tile epilogue
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result \
  -mllvm -amdgpu-mfma-vgpr-form=1 \
  "$@" -o "$OUT" pack.hip sampler.hip render_fwd.hip
echo "built $OUT"
