#!/usr/bin/env python3
"""Single-search latency probe for the GPU A* (ts_astar = one wave, nothing else on the chip):
expansions per second of lone searches on a synthetic city, the figure that bounds a tick's longest replan."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--queries", type=int, default=12)
    ap.add_argument("--occupancy", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from trafficsimulation_amd import citygen
    from trafficsimulation_amd._lib import new_engine
    from trafficsimulation_amd.world import build_engine
    tb = citygen.generate(args.size, args.size, seed=args.seed)
    api = new_engine()
    build_engine(api, tb, defaults={"RAIN_ENABLED": False}, global_seed=1, sched_seed=1)
    rng = np.random.default_rng(args.seed)
    road = (np.asarray(tb["is_road_map"]) == 1) & (np.asarray(tb["intersection_map"]) == 0)
    ys, xs = np.nonzero(road)
    occ = np.zeros_like(tb["is_road_map"], dtype=np.int8)
    pick = rng.choice(len(xs), size=int(len(xs) * args.occupancy), replace=False)
    occ[ys[pick], xs[pick]] = 1
    api.debug_set_occupancy(occ)
    free = np.nonzero(road & (occ == 0))
    tot_exp = tot_t = 0
    for q in range(args.queries):
        a, b = rng.integers(len(free[0]), size=2)
        sx, sy, gx, gy = int(free[1][a]), int(free[0][a]), int(free[1][b]), int(free[0][b])
        for soft in (False, True):
            c0 = api.counters()
            t0 = time.perf_counter()
            path = api.astar(sx, sy, gx, gy, soft, False, 0x7FFFFFFF)
            dt = time.perf_counter() - t0
            c1 = api.counters()
            ex = c1.astar_expansions - c0.astar_expansions
            dbg = (ctypes.c_int32 * 8)()
            api.lib.ts_debug_read(api.h, dbg)
            cyc = (dbg[0] & 0xFFFFFFFF) | (dbg[1] << 32)
            wall = ((dbg[2] & 0xFFFFFFFF) | (dbg[3] << 32)) * 10.0   # ns
            print(f"q{q} soft={int(soft)} md={abs(sx-gx)+abs(sy-gy)} len={len(path)} exp={ex} relax={c1.astar_relaxations - c0.astar_relaxations} "
                  f"{dt*1e3:.2f} ms  {dt/max(ex,1)*1e9:.0f} ns/exp  kernel {wall/1e6:.2f} ms, {cyc/max(ex,1):.0f} cycles/exp, {cyc/max(wall,1):.2f} GHz, max heap {dbg[4]}")
            if ex > 2000:
                tot_exp += ex
                tot_t += dt
    print(f"TOTAL {tot_exp} expansions in {tot_t*1e3:.1f} ms -> {tot_t/max(tot_exp,1)*1e9:.0f} ns/expansion, {tot_exp/max(tot_t,1e-9)/1e6:.2f} M exp/s per wave")
    api.close()


if __name__ == "__main__":
    main()
