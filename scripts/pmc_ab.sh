#!/bin/bash
# PMC passes over the traversal kernel (1M-triangle scene, 4M incoherent surface rays) under several environments, one counter group per run.
# usage: bash scripts/pmc_ab.sh <outdir> "<ENV=.. ENV=..>|<label>" ...
R=$PWD; out=$R/$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  envs=${spec%%|*}; label=${spec##*|}
  export $envs
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-24)
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/${label}_$n -o pmc -- python3 $R/scripts/prof_target.py 1000000 4194304 3 > $out/${label}_$n.log 2>&1 || { echo "pass $c failed for $label"; tail -5 $out/${label}_$n.log; }
  done
  echo "== $label"; python3 $R/scripts/pmc_summary.py --min-grid 300000 $out/${label}_* | grep "k_trace"
done
