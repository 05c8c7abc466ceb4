#!/bin/bash
# PMC passes over the traversal kernel (1M-triangle scene, 4M incoherent surface rays): one rocprofv3 run per counter
# group given as an argument.   usage: bash scripts/pmc_groups.sh <tag> "<group 1 counters>" "<group 2 counters>" ...
R=$PWD; tag=$1; shift
i=0
for c in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_g$i -o pmc -- python3 $R/scripts/prof_target.py 1000000 4194304 3 > $R/gpurun_out/pmc_${tag}_g$i.log 2>&1 || { echo "pass $c failed"; tail -5 $R/gpurun_out/pmc_${tag}_g$i.log; exit 1; }
done
python3 scripts/pmc_summary.py $R/gpurun_out/pmc_${tag}_g* | grep k_trace
