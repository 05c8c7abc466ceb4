#!/bin/bash
# A/B builds of one kernel file: scripts/build_variant.sh <suffix> <file.hip> <extra hipcc flags...>
# -> pim_amd/libpimgpu_<suffix>.so (that file recompiled with the flags, the other objects reused); select it with PIMGPU_LIB.
set -e
cd "$(dirname "$0")/../pim_amd/csrc"
make -s
suffix=$1; src=$2; shift; shift
stem=${src%.hip}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-fast-math -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o build/${stem}_$suffix.var.o
objs=$(ls build/*.o | grep -v "\.var\.o" | grep -v "build/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpimgpu_$suffix.so $objs build/${stem}_$suffix.var.o
echo "built pim_amd/libpimgpu_$suffix.so"
