#!/bin/bash
# Builds tuning variants of libsplat_hip.so into gpurun_variants/ (not part of the product):
#   tools/build_variants.sh name1:"-DX=1 -DY=2" name2:"..."
set -e
cd "$(dirname "$0")/../splatter_a_video_amd/csrc"
mkdir -p ../../variants build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -fno-slp-vectorize"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc $FLAGS $defs -c blend.hip -o build/blend_$name.o &
done
wait
for spec in "$@"; do
  name="${spec%%:*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libsplat_$name.so build/runtime.o build/pointwise.o build/binning.o build/dynamics.o build/preprocess.o build/densify.o build/knn.o build/optim.o build/frames.o build/arap.o build/blend_$name.o
  echo built variants/libsplat_$name.so
done
