#!/bin/bash
# tuning variants of preprocess.hip into variants/ (not part of the product): tools/build_pre_variants.sh name:"-DX=.." ...
set -e
cd "$(dirname "$0")/../splatter_a_video_amd/csrc"
mkdir -p ../../variants build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I../../include -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function"
for spec in "$@"; do
  n="${spec%%:*}"; d="${spec#*:}"
  /opt/rocm/bin/hipcc $F $d -c preprocess.hip -o build/pre_$n.o &
done
wait
for spec in "$@"; do
  n="${spec%%:*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libsplat_$n.so build/runtime.o build/pointwise.o build/binning.o build/dynamics.o build/pre_$n.o build/densify.o build/knn.o build/optim.o build/frames.o build/arap.o build/blend.o
  echo built variants/libsplat_$n.so
done
