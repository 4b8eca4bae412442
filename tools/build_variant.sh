#!/bin/bash
# variants/libsplat_<name>.so: the library with blend.hip (or another unit: UNIT=binning) recompiled under extra -D flags, for
# A/B timing on the GPU box (SPLAT_LIB_PATH=variants/libsplat_<name>.so python bench.py ...).  The other objects are the in-tree build's.
#   tools/build_variant.sh <name> -DBLEND_Q_SB=64 -DBLEND_Q_CAP=48
set -e
cd "$(dirname "$0")/../splatter_a_video_amd/csrc"
name=$1; shift
unit=${UNIT:-blend}
mkdir -p build/var ../../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function"
[ "$unit" = blend ] && FLAGS="$FLAGS -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $unit.hip -o build/var/${unit}_$name.o
objs=""
for o in runtime pointwise binning blend dynamics preprocess densify knn optim frames arap; do
  if [ "$o" = "$unit" ]; then objs="$objs build/var/${unit}_$name.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libsplat_$name.so $objs
echo built variants/libsplat_$name.so
