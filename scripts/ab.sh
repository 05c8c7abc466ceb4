#!/bin/bash
# A/B runner: scripts/ab.sh <outfile> then lines "lib-suffix|ENV=.. ENV=.." on stdin; both bench workloads per line
set -e
mkdir -p gpurun_out
out=gpurun_out/$1; : > $out
while IFS='|' read -r lib envs; do
  for wl in cornell_1080p synth1m_1080p; do
    L=$PWD/pim_amd/libpimgpu.so; [ "$lib" != "base" ] && L=$PWD/pim_amd/libpimgpu_$lib.so
    r=$(env PIMGPU_LIB=$L $envs timeout -k 10 150 python bench.py --workload $wl --steps 32 --warmup 4 --no-cpu-baseline \
      | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'])")
    echo "$wl lib=$lib $envs : $r" >> $out
  done
done
sort $out
