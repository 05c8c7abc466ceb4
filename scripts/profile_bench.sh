#!/bin/bash
# rocprofv3 evidence for one bench workload: kernel trace + four PMC passes (one counter group per run, as the
# MI355X guide prescribes; FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage (on the GPU box, from the repo root): bash scripts/profile_bench.sh <workload> <tag>
R=$PWD; wl=${1:-synth1m_4k}; tag=${2:-r02}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_${tag}_$wl; mkdir -p $out
timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d $out/kt -o bench -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sub --workload $wl > $out/kt.log 2>&1 || { echo "kernel trace failed"; exit 1; }
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$n -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sub --workload $wl > $out/pmc_$n.log 2>&1 || { echo "pass $c failed"; exit 1; }
done
cd $R
python3 scripts/rocpd_summary.py $(find $out/kt -name "*.db" | head -1) 8 2 > $out/kernel_stats.txt
python3 scripts/pmc_summary.py --min-grid 4096 $out/pmc_* | grep -E "k_trace|k_shade|k_tail|k_init_image|k_resolve_image" > $out/pmc_summary.txt      # the kernels of the timed steps
grep -h '"metric"' $out/kt.log | tail -1 > $out/bench_line.json
echo done $wl
