#!/bin/bash
# SQ instruction / cycle counters of the replanning kernels on the final build at the headline size (two rocprofv3 --pmc
# passes of 8 SQ counters each, no tracing beside them), for k_replan alone (TS_QUAD=0) and for the default (k_replan_quad on the
# replanning waves with k_replan beside it, k_replan on the other ticks).
# Run on the GPU box from the repository root:  bash profiles/run_sq_profile.sh  ->  gpurun_out/r3_sq_*.json
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="bench.py --warmup 0 --steps 6 --no-cpu-baseline --no-secondary"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-plain quad}; do
  if [ $mode = quad ]; then unset TS_QUAD; else export TS_QUAD=0; fi     # (quad = the default: k_replan_quad on the big queues)
  rm -rf /tmp/sq_${mode}_1 /tmp/sq_${mode}_2
  rocprofv3 --pmc $P1 -d /tmp/sq_${mode}_1 -o pmc -- python3 $R/$CMD > $OUT/r3_sq_${mode}_bench1.json 2> $OUT/r3_sq_${mode}_1.err
  echo "$mode pass 1 done"
  rocprofv3 --pmc $P2 -d /tmp/sq_${mode}_2 -o pmc -- python3 $R/$CMD > $OUT/r3_sq_${mode}_bench2.json 2> $OUT/r3_sq_${mode}_2.err
  echo "$mode pass 2 done"
  A=$(find /tmp/sq_${mode}_1 -name "*results.db" | head -1)
  B=$(find /tmp/sq_${mode}_2 -name "*results.db" | head -1)
  python3 $R/profiles/summarize_sq.py $OUT/r3_sq_${mode}.json $OUT/r3_sq_${mode}_bench1.json "python3 $CMD ($mode; 4096^2 / 10^6 vehicles, default policy, ticks 1-6 incl. the first replanning wave)" $A $B > $OUT/r3_sq_${mode}_print.txt
  cat $OUT/r3_sq_${mode}_print.txt
done
