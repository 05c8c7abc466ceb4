#!/bin/bash
# A/B builds of the traversal kernel: scripts/build_variant.sh <suffix> <extra hipcc flags...>
# -> pim_amd/libpimgpu_<suffix>.so (trace.hip recompiled with the flags, the other objects reused); select it with PIMGPU_LIB.
set -e
cd "$(dirname "$0")/../pim_amd/csrc"
make -s
suffix=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-fast-math -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c trace.hip -o build/trace_$suffix.o
objs=$(ls build/*.o | grep -v "build/trace" )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpimgpu_$suffix.so $objs build/trace_$suffix.o
echo "built pim_amd/libpimgpu_$suffix.so"
