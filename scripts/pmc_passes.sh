#!/bin/bash
# PMC passes over the traversal kernel (1M-triangle scene, 4M incoherent surface rays), one counter group per run.
# usage: bash scripts/pmc_passes.sh <tag>
R=$PWD; tag=${1:-x}
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-28)
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -o pmc -- python3 $R/scripts/prof_target.py 1000000 4194304 3 > $R/gpurun_out/pmc_${tag}_$n.log 2>&1 || { echo "pass $c failed"; exit 1; }
done
python3 scripts/pmc_summary.py $R/gpurun_out/pmc_${tag}_* | grep k_trace
