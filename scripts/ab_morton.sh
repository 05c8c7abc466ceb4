#!/bin/bash
# Morton-code width: build time at 20 M triangles, tree quality at 1 M and 20 M (nodes / triangles per incoherent ray)
for lib in "$@"; do
  L=$PWD/pim_amd/libpimgpu.so; [ "$lib" != "base" ] && L=$PWD/pim_amd/libpimgpu_$lib.so
  echo "== $lib"
  PIMGPU_LIB=$L timeout -k 10 300 python -u scripts/build_prof.py 20000000 4 2>&1 | tail -1 || exit 1
  PIMGPU_LIB=$L PIMGPU_TRACE_DIAG=1 timeout -k 10 200 python -u scripts/raybench.py 1000000 4194304 2>&1 | grep -E "n= 4194304|checksum" || exit 1
  PIMGPU_LIB=$L PIMGPU_TRACE_DIAG=1 timeout -k 10 300 python -u scripts/raybench.py 20000000 4194304 2>&1 | grep -E "n= 4194304|checksum" || exit 1
done
