#!/bin/bash
# traversal tuning sweep: variant libraries x blocks per CU (each run under its own time limit)
set -e
for lib in "" _r1 _r8 _r32 _l2 _l8; do
  echo "== lib$lib"
  PIMGPU_LIB=$PWD/pim_amd/libpimgpu$lib.so timeout -k 10 120 python -u scripts/raybench.py 1000000 4194304 2>/dev/null | grep "n= 4194304"
done
for b in 4 6 12 16; do
  echo "== blocks/CU $b"
  PIMGPU_TRACE_BLOCKS_PER_CU=$b timeout -k 10 120 python -u scripts/raybench.py 1000000 4194304 2>/dev/null | grep "n= 4194304"
done
