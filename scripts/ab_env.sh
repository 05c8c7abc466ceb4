#!/bin/bash
# A/B of one environment knob on both bench workloads: scripts/ab_env.sh NAME v1 v2 ...
name=$1; shift
for v in "$@"; do
  for wl in cornell_1080p synth1m_1080p; do
    env $name=$v timeout -k 10 200 python bench.py --workload $wl --steps 32 --warmup 4 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$name=$v', '$wl', round(j['value'],1), round(j['ms_per_step'],3))" || exit 1
  done
done
