#!/bin/bash
# rocprofv3 evidence for one bench workload: kernel trace + PMC passes (one counter group per run, as the MI355X guide
# prescribes; FETCH_SIZE and WRITE_SIZE cannot share a pass).  Every pass runs `bench.py --plain`: warm-up + timed steps and
# nothing else, so that counter totals divide by frames x units per step (scripts/pmc_per_unit.py).
# usage (on the GPU box, from the repo root): bash scripts/profile_bench.sh <workload|microbench> <tag>
R=$PWD; wl=${1:-synth1m_4k}; tag=${2:-r03}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_${tag}_$wl; mkdir -p $out
if [ "$wl" = "incoherent_closest_hit_1m" ]; then target="--microbench-only"; kt="--microbench-only"; T=170
elif [ "$wl" = "incoherent_closest_hit_20m" ]; then target="--microbench-only --workload synth20m_1080p"; kt="$target"; T=280
elif [ "$wl" = "synth20m_1080p" ]; then target="--plain --steps 3 --warmup 1 --workload $wl"; kt="--plain --steps 4 --warmup 1 --workload $wl"; T=280
else target="--plain --steps 4 --warmup 1 --workload $wl"; kt="--plain --steps 8 --warmup 2 --workload $wl"; T=170; fi
timeout -s KILL $T rocprofv3 --kernel-trace --stats -d $out/kt -o bench -- python3 $R/bench.py $kt > $out/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $out/kt.log; exit 1; }
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout -s KILL $T rocprofv3 --pmc $c --output-format csv -d $out/pmc_$n -o pmc -- python3 $R/bench.py $target > $out/pmc_$n.log 2>&1 || { echo "pass $c failed"; tail -5 $out/pmc_$n.log; exit 1; }
done
cd $R
steps=8; warm=2; [ "$wl" = "synth20m_1080p" ] && { steps=4; warm=1; }
python3 scripts/rocpd_summary.py $(find $out/kt -name "*.db" | head -1) $steps $warm > $out/kernel_stats.txt
python3 scripts/pmc_summary.py $out/pmc_* | grep -E "k_trace|k_shade|k_tail|k_init_image|k_resolve_image" > $out/pmc_summary.txt
grep -h -E '"metric"|"workload"' $out/kt.log | tail -1 > $out/bench_line.json
grep -h -E '"metric"|"workload"' $out/pmc_FETCH_SIZE.log | tail -1 > $out/pmc_line.json
# the counter CSVs and the trace database stay on the box (gpurun_out/ travels back up to 64 MiB): their figures per unit of work now
python3 scripts/pmc_per_unit.py $tag --on-box $wl > $out/per_unit.log 2>&1
rm -rf $out/kt $out/pmc_*/
echo done $wl
