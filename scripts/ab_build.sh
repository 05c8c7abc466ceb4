#!/bin/bash
# A/B of BVH builder variants in one call: build time at 20 M and 1 M triangles and the tree's quality (nodes / triangles per incoherent ray)
# usage: bash scripts/ab_build.sh <outdir> base <suffix> ...
out=$1; shift; mkdir -p $out
for lib in "$@"; do
  L=$PWD/pim_amd/libpimgpu.so; [ "$lib" != "base" ] && L=$PWD/pim_amd/libpimgpu_$lib.so
  echo "== $lib"
  PIMGPU_LIB=$L timeout -k 10 300 python -u scripts/build_prof.py 20000000 4 2>&1 | tail -1 || exit 1
  PIMGPU_LIB=$L timeout -k 10 200 python -u scripts/raybench.py 1000000 4194304 2>&1 | grep -E "^tris|n= 4194304|checksum" || exit 1
done
