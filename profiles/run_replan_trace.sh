#!/bin/bash
# Per-search timeline + per-segment cycle profile of k_replan on the bench workload (4096^2 / 10^6 vehicles, default policy).
# Builds two profiling variants of the library (never the product build), runs the bench's first eleven ticks with each:
#   -DTS_TRACE_REPLAN -> gpurun_out/rtrace_tick*.bin -> profiles/replan_trace.py -> gpurun_out/replan_trace.json
#   -DTS_KPROF        -> gpurun_out/kprof.log (clock64 per segment of astar_loop, of the search that finished last in every tick)
#   gpurun --timeout 600 -- 'bash profiles/run_replan_trace.sh'
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 -fPIC -pthread -Wno-unused-function --offload-arch=gfx950 -shared"
mkdir -p gpurun_out
( cd trafficsimulation_amd/csrc && /opt/rocm/bin/hipcc -DTS_TRACE_REPLAN $F -o /tmp/libts_trace.so engine.hip && /opt/rocm/bin/hipcc -DTS_KPROF $F -o /tmp/libts_kprof.so engine.hip )
rm -f gpurun_out/rtrace_tick*.bin
TS_HIP_LIB=/tmp/libts_trace.so timeout -k 10 300 python3 bench.py --warmup 5 --steps 6 --no-cpu-baseline > gpurun_out/rtrace_bench.log 2> gpurun_out/rtrace_bench.err
python3 profiles/replan_trace.py gpurun_out/rtrace_tick*.bin > gpurun_out/replan_trace.json
TS_KPROF=1 TS_HIP_LIB=/tmp/libts_kprof.so timeout -k 10 300 python3 profiles/kprof_run.py > gpurun_out/kprof.log 2>&1
tail -8 gpurun_out/kprof.log
