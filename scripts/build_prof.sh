#!/bin/bash
# kernel-level breakdown of the BVH build: bash scripts/build_prof.sh <tris>
R=$PWD; n=${1:-4000000}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/buildprof_$n; mkdir -p $out
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $out/kt -o b -- python3 $R/scripts/build_prof.py $n 3 > $out/log.txt 2>&1 || { echo failed; tail -5 $out/log.txt; exit 1; }
cd $R
python3 scripts/rocpd_summary.py $(find $out/kt -name "*.db" | head -1) > $out/kernel_stats.txt
tail -2 $out/log.txt; head -40 $out/kernel_stats.txt | cut -c1-150
