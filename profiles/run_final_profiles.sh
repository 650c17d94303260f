#!/bin/bash
# The measurement pass behind profiles/r03_*: the driver's bench command, the same command under rocprofv3 --kernel-trace
# --stats, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE).  Run on the GPU box from the repository root:
#   bash profiles/run_final_profiles.sh   ->  gpurun_out/fin_*
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
ARGS="--gpus 1 --steps 20 --warmup 5"
python bench.py $ARGS > $OUT/fin_bench.json 2> $OUT/fin_bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fin_stats /tmp/fin_fetch /tmp/fin_write
rocprofv3 --kernel-trace --stats -d /tmp/fin_stats -o bench -- python3 $R/bench.py $ARGS --no-cpu-baseline > $OUT/fin_bench_under_rocprof.json 2> $OUT/fin_stats.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d /tmp/fin_fetch -o pmc -- python3 $R/bench.py $ARGS --no-cpu-baseline > $OUT/fin_fetch.json 2> $OUT/fin_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d /tmp/fin_write -o pmc -- python3 $R/bench.py $ARGS --no-cpu-baseline > $OUT/fin_write.json 2> $OUT/fin_write.err
echo "write pass done"
cd $R
S=$(find /tmp/fin_stats -name "*results.db" | head -1)
F=$(find /tmp/fin_fetch -name "*results.db" | head -1)
W=$(find /tmp/fin_write -name "*results.db" | head -1)
python profiles/summarize_pmc.py stats $S $OUT/fin_kernel_stats.csv replan_regions:20 k_replan_quad:4 > $OUT/fin_stats_summary.txt
python profiles/summarize_pmc.py pmc $F $W $OUT/fin_pmc_summary.json 4096,1000000,full,20,5 > $OUT/fin_pmc_print.txt
echo "summaries done"
