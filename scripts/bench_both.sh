#!/bin/bash
# both bench workloads, short form: value and ms per step
for wl in cornell_1080p synth1m_1080p; do
  timeout -k 10 200 python bench.py --workload $wl --steps 32 --warmup 4 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$wl', round(j['value'],1), round(j['ms_per_step'],3))" || exit 1
done
