#!/bin/bash
# PMC passes over the traversal kernel (1M-triangle scene, 4M incoherent surface rays) for several variant libraries, one counter group per run.
# usage: bash scripts/pmc_ab.sh <outdir> base <suffix> ...
R=$PWD; out=$R/$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  L=$R/pim_amd/libpimgpu.so; [ "$lib" != "base" ] && L=$R/pim_amd/libpimgpu_$lib.so
  export PIMGPU_LIB=$L
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-24)
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/${lib}_$n -o pmc -- python3 $R/scripts/prof_target.py 1000000 4194304 3 > $out/${lib}_$n.log 2>&1 || { echo "pass $c failed for $lib"; tail -5 $out/${lib}_$n.log; }
  done
  echo "== $lib"; python3 $R/scripts/pmc_summary.py $out/${lib}_* | grep "k_trace"
done
