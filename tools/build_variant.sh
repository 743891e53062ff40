#!/bin/bash
# Builds build_var/libsunerf_hip_<name>.so with extra hipcc flags: tools/build_variant.sh <name> [-DFLAG ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build_var/obj_$name
SUNERF_BUILD_OUT=$R/build_var/libsunerf_hip_$name.so SUNERF_BUILD_OBJ=$R/build_var/obj_$name bash $R/2024-hl-spi3s-sunerf_amd/csrc/build.sh "$@"
