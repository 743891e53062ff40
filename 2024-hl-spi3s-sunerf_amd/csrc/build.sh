#!/bin/bash
# Builds libsunerf_hip.so (gfx950) in-tree.  Usage: ./build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libsunerf_hip.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result \
  "$@" -o "$OUT" pack.hip sampler.hip render_fwd.hip
echo "built $OUT"
