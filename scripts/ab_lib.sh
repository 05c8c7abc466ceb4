#!/bin/bash
# A/B of variant libraries (scripts/build_variant.sh) on both bench workloads: scripts/ab_lib.sh base <suffix> ...
for lib in "$@"; do
  L=$PWD/pim_amd/libpimgpu.so; [ "$lib" != "base" ] && L=$PWD/pim_amd/libpimgpu_$lib.so
  for wl in cornell_1080p synth1m_1080p; do
    PIMGPU_LIB=$L timeout -k 10 200 python bench.py --workload $wl --steps 32 --warmup 4 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$lib', '$wl', round(j['value'],1), round(j['ms_per_step'],3))" || exit 1
  done
done
