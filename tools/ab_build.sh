#!/bin/bash
# A/B helper: builds a variant of librm_hip.so under build/variants/ (travels to the GPU box).
# usage: tools/ab_build.sh NAME [extra hipcc flags / -D switches]
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -fvisibility=hidden -Iinclude -Iray-marching_amd/csrc"
name=$1; shift
hipcc $FLAGS "$@" -o build/variants/librm_hip_$name.so ray-marching_amd/csrc/rm_abi.hip
echo built build/variants/librm_hip_$name.so
