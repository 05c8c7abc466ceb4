#!/bin/bash
# traversal A/B sweep: variant libraries (scripts/build_variant.sh) through the incoherent-ray microbenchmark,
# each run under its own time limit:  scripts/rb_sweep.sh base "" v16 v32 ...
set -e
for lib in "$@"; do
  name=${lib:+_$lib}
  echo "== libpimgpu$name"
  PIMGPU_LIB=$PWD/pim_amd/libpimgpu$name.so timeout -k 10 150 python -u scripts/raybench.py 1000000 4194304 2>/dev/null | grep "n= 4194304\|checksum"
done
