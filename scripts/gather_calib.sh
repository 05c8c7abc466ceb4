#!/bin/bash
# scripts/micro/gather under the FETCH_SIZE and TCC request counters (one pass each): known bytes beside counted bytes
R=$PWD; cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/gather; mkdir -p $out
timeout -k 10 120 $R/scripts/micro/gather > $out/times.txt 2>&1 || { echo "plain run failed"; tail -3 $out/times.txt; exit 1; }
cat $out/times.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- $R/scripts/micro/gather > $out/pmc_fetch.log 2>&1 || { echo "FETCH_SIZE pass failed"; tail -5 $out/pmc_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d $out/pmc_req -o pmc -- $R/scripts/micro/gather > $out/pmc_req.log 2>&1 || { echo "request pass failed"; tail -5 $out/pmc_req.log; }
cd $R
python3 scripts/pmc_summary.py $out/pmc_* > $out/pmc_summary.txt 2>&1
cat $out/pmc_summary.txt | cut -c1-200
rm -rf $out/pmc_fetch $out/pmc_req
