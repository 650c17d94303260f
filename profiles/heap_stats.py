#!/usr/bin/env python3
"""What the A* heaps of the bench workload look like (CPU oracle built with -DTSO_STATS; test infrastructure).

    g++ -O2 -std=c++17 -fPIC -ffp-contract=off -DTSO_STATS -shared -o /tmp/libtso_stats.so oracle/tso.cpp
    python profiles/heap_stats.py --size 1024 --vehicles 60000 --ticks 8

Prints, over all searches of those ticks: heap size at the pops (buckets of 64), levels a pop's hole sinks, levels a
push rises, relaxations per expansion, searches by bit length of their expansions."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from trafficsimulation_amd._capi import CApi

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--vehicles", type=int, default=60000)
ap.add_argument("--ticks", type=int, default=8)
ap.add_argument("--lib", default="/tmp/libtso_stats.so")
ap.add_argument("--out", default="")
a = ap.parse_args()
lib = ctypes.CDLL(a.lib)
api = CApi(lib, "tso_")
tables, routes, _ = bench.make_workload(a.size, a.vehicles, 1)
bench.setup(api, tables, routes, 1, extra={"eager_density": 1}, policy="full")
buf = (ctypes.c_longlong * (191 + 32 * 5 + 4 * 4))()
res = []
for t in range(a.ticks):
    api.step(1)
    lib.tso_stats_read(buf)
    v = np.array(buf[:], dtype=np.int64)
    c = api.counters()
    r = dict(tick=t + 1, calls=int(c.astar_calls), exp=int(c.astar_expansions), pop_size64=v[:64].tolist(), sink=v[64:104].tolist(),
             rise=v[104:144].tolist(), relax_n=v[144:149].tolist(), search_exp_bits=v[149:189].tolist(), pops=int(v[189]), stale=int(v[190]),
             ext_n=v[191:223].tolist(), ext_exp=v[223:255].tolist(), empty_n=v[255:259].tolist(), empty_exp=v[259:263].tolist(),
             kind_n=v[263:267].tolist(), kind_exp=v[267:271].tolist(), maxheap_n=v[271:303].tolist(), maxheap_exp=v[303:335].tolist(), fspread=v[335:367].tolist())
    res.append(r)
    print(json.dumps(r), flush=True)
if a.out:
    json.dump(res, open(a.out, "w"))
