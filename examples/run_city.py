"""Headless run of the reference's city on the MI355X engine: `python examples/run_city.py --size 200 --seed 1 --ticks 500`.

Everything comes from (size, seed): worldgen builds the city the reference builds after `random.seed(seed)`, the engine's
traffic generator schedules day 0, then `model.step()` is CityModel.step().  Prints the traffic generator's statistics
every `--every` ticks, like the reference's console output.  Needs the HIP library and a GPU (no CPU fallback)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# config.py's values for what DynamicTrafficAgent, the service fleet and the city blocks read (config.py:238-253, 333-335)
TRAFFIC = dict(P_int=10000, P_thr=2400, dt=6, start_offset=6 * 3600, service_food=50, service_waste=50,
               max_load_food=50.0, max_load_waste=250.0, load_time=20, gradual=True, food_capacity_per_cell=2.0,
               waste_capacity_per_cell=1.5, food_consumption_ticks=50, waste_production_ticks=100)

def run(size, seed, ticks, every=50, engine=None, traffic=None, defaults=None, out=print, **world_kwargs):
    from trafficsimulation_amd.mesa_api import CityModel
    t0 = time.time()
    m = CityModel(size, size, seed=seed, defaults=defaults or {}, traffic=dict(TRAFFIC, **(traffic or {})), engine=engine,
                  **world_kwargs)
    out(f"city {m.width}x{m.height} seed {seed}: {len(m.intersection_light_groups)} light groups, {len(m.city_blocks)} blocks, "
        f"{len(m.get_start_blocks())} entries, built in {time.time() - t0:.1f} s")
    t0 = time.time()
    for t in range(1, ticks + 1):
        m.step()
        if t % every == 0 or t == ticks:
            s = m.dynamic_traffic_generator.cached_stats
            out(f"tick {t:6d}  live {len(m.active_vehicle_agents):6d}  internal {s['live_internal']:5d}  through {s['live_through']:5d}  "
                f"service {s['live_service_food'] + s['live_service_waste']:3d}  parked {s['parked']:4d}  stuck {s['stuck']:4d}  "
                f"clouds {len(m.rains)}  {1e3 * (time.time() - t0) / t:.2f} ms/tick")
    return m


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ticks", type=int, default=500)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--carve", action="store_true", help="carve_subblock_roads=True (BASELINE config 5 style)")
    a = ap.parse_args()
    kw = dict(carve_subblock_roads=True) if a.carve else {}
    run(a.size, a.seed, a.ticks, a.every, **kw)
