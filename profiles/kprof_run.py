"""Per-segment cycle profile of the search that finishes last in each of the bench workload's first ticks (a -DTS_KPROF build,
see profiles/run_replan_trace.sh): ts_debug_read prints the clock64 totals of astar_loop's segments for that search."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.getcwd())
import bench
from trafficsimulation_amd._lib import new_engine
tables, routes, _ = bench.make_workload(4096, 1_000_000, 1)
api = new_engine()
bench.setup(api, tables, routes, 1, policy="full")
buf = (C.c_int32 * 8)()
f = api.lib.ts_debug_read
f.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
for t in range(5):
    t0 = time.time(); api.step(1); dt = time.time() - t0
    print("tick", t, "%.1f ms" % (dt * 1e3), flush=True)
    f(api.h, buf)
