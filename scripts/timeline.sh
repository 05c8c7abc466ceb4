#!/bin/bash
# per-dispatch timeline of one bench step: bash scripts/timeline.sh <workload> <tag>   (environment knobs are inherited)
R=$PWD; wl=${1:-cornell_1080p}; tag=${2:-tl}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/tl_${tag}_$wl; mkdir -p $out
timeout -s KILL 170 rocprofv3 --kernel-trace -d $out/kt -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sub --workload $wl > $out/kt.log 2>&1 || { echo "kernel trace failed"; exit 1; }
cd $R
python3 scripts/step_timeline.py $(find $out/kt -name "*.db" | head -1) > $out/timeline.txt
