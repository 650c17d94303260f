"""What ONE rank of an N-rank sharded run does in a replanning wave, measured on one GPU: the engine is put into the sharded mode as
rank 0 of N with a collective that returns nothing from the other ranks, so it plans every N-th entry of the queue and imports no
one else's results (the simulation is NOT the reference's from then on - this is a timing probe, not a parity run).  Prints the wall
time of the bench workload's ticks 1-6; tick 6 is the first replanning wave.
    TS_QUAD_MIN=... python profiles/emulate_share.py N"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from trafficsimulation_amd._lib import new_engine
from trafficsimulation_amd.dist import ShardedReplans

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def lonely_all_gather(out, t):          # rank 0's contribution; from the others an empty exchange header (176 zero bytes: no records)
    for i, o in enumerate(out):
        if i == 0:
            o.copy_(t)
        elif t.dtype == torch.int64:
            o.fill_(176)
        else:
            o.zero_()


tables, routes, _ = bench.make_workload(4096, 1_000_000, 1)
api = new_engine()
bench.setup(api, tables, routes, 1, policy="full")
if N > 1:
    ShardedReplans(all_gather=lonely_all_gather, rank=0, world=N).attach(api)
for t in range(6):
    c0 = api.counters()
    t0 = time.perf_counter()
    api.step(1)
    dt = time.perf_counter() - t0
    c1 = api.counters()
    print(f"tick {t + 1}: {dt * 1e3:.1f} ms, {c1.astar_calls - c0.astar_calls} searches, {c1.astar_expansions - c0.astar_expansions} expansions", flush=True)
