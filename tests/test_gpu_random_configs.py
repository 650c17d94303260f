"""Randomised differential test: HIP engine vs CPU oracle on small synthetic worlds with parameter sets drawn at
random (light algorithms, speed ranges with power-of-two and non-power-of-two spans, awareness range, replanning
thresholds, contraflow switches, stranding chances, penalties, rain with a manager, transition timers, stuck despawn,
field-of-view masking, fractional road-type penalties, the non-batched step path).  Every
tick is compared state for state; the draws are seeded, so a failure names its case."""
import os

import numpy as np
import pytest

from trafficsimulation_amd import _capi as capi
from trafficsimulation_amd import citygen
from trafficsimulation_amd.world import build_engine

pytestmark = pytest.mark.gpu


def random_case(case: int):
    rng = np.random.default_rng(1000 + case)
    size = int(rng.choice([64, 96, 128, 160]))
    vmin = int(rng.integers(1, 3))
    vmax = int(vmin + rng.choice([0, 1, 2, 3, 4, 6]))
    d = {
        "TRAFFIC_LIGHT_AGENT_ALGORITHM": str(rng.choice(["DISABLED", "FIXED_TIME", "QUEUE_ACTUATED", "NEIGHBOR_PRESSURE_CONTROL",
                                                         "NEIGHBOR_GREEN_WAVE", "PRESSURE_CONTROL"])),
        "VEHICLE_MIN_SPEED": vmin, "VEHICLE_MAX_SPEED": vmax,
        "VEHICLE_AWARENESS_RANGE": int(rng.choice([3, 6, 10, 14, 16])),
        "PATHFINDING_COOLDOWN": int(rng.choice([0, 2, 5, 10 ** 9])),
        "VEHICLE_STUCK_RECOMPUTE_THRESHOLD": int(rng.choice([3, 10, 30, 10 ** 9])),
        "VEHICLE_STUCK_RECOMPUTE_THRESHOLD_INTERSECTION": int(rng.choice([1, 4, 10 ** 9])),
        "VEHICLE_CONTRAFLOW_OVERTAKE_ACTIVE": bool(rng.integers(2)),
        "VEHICLE_STUCK_CONTRAFLOW_ENABLED": bool(rng.integers(2)),
        "VEHICLE_STUCK_CONTRAFLOW_THRESHOLD": int(rng.choice([5, 20, 60])),
        "VEHICLE_STUCK_CONTRAFLOW_THRESHOLD_INTERSECTION": int(rng.choice([2, 10])),
        "VEHICLE_MALFUNCTION_CHANCE": float(rng.choice([0.0, 1e-7, 0.003, 0.02])),
        "VEHICLE_MALFUNCTION_DURATION": int(rng.integers(3, 40)),
        "VEHICLE_SIDESWIPE_COLLISION_ACTIVE": bool(rng.integers(4) > 0),
        "VEHICLE_SIDESWIPE_COLLISION_CHANCE": float(rng.choice([0.0, 1e-9, 0.05, 0.5])),
        "VEHICLE_SIDESWIPE_COLLISION_DURATION": int(rng.integers(3, 40)),
        "VEHICLE_TURN_PENALTY_ENABLED": bool(rng.integers(2)),
        "VEHICLE_ROAD_TYPES_PENALTIES_ENABLED": bool(rng.integers(2)),
        "VEHICLE_DYNAMIC_PENALTIES_ENABLED": bool(rng.integers(2)),
        "TRAFFIC_LIGHT_TRANSITION_DURATION_ENABLED": bool(rng.integers(2)),
        "TRAFFIC_LIGHT_TRANSITION_CLEARANCE_ENABLED": bool(rng.integers(2)),
        "TRAFFIC_LIGHT_ALL_RED_DURATION": int(rng.integers(1, 5)),
        "TRAFFIC_LIGHT_GREEN_DURATION": int(rng.integers(4, 30)),
        "RAIN_ENABLED": bool(rng.integers(2)),
        "RAIN_RADIUS_MIN": 6, "RAIN_RADIUS_MAX": 20, "RAIN_SPAWN_CHANCE": 0.3, "RAIN_SPEED_REDUCTION": int(rng.integers(1, 4)),
        # round 2: _despawn_check, field-of-view masking, and penalties that are not multiples of 0.5 (the A*'s double path
        # instead of its integer half-unit path)
        "VEHICLE_STUCK_DESPAWN_ENABLED": bool(rng.integers(4) == 0),
        "VEHICLE_STUCK_DESPAWN_THRESHOLD": int(rng.choice([6, 20, 3600])),
        "VEHICLE_STUCK_DESPAWN_THRESHOLD_INTERSECTION": int(rng.choice([2, 20])),
        "VEHICLE_RESPECT_AWARENESS": bool(rng.integers(4) == 0),
        "VEHICLE_ROAD_TYPES_PENALTY_R2": float(rng.choice([5.0, 5.25, 7.5])),
        "VEHICLE_TURN_PENALTY": int(rng.choice([10, 3])),
    }
    vehicles = int(rng.integers(40, 700))
    # round 3: the non-batched step path for every fifth case or so (a stream of its own: the other draws of a case stay what they were)
    if os.environ.get("TS_HUNT_NOBATCH") == "1" or np.random.default_rng(9000 + case).integers(5) == 0:
        d["PATHFINDING_BATCHING"] = False
        vehicles = min(vehicles, 250)
    with_manager = d["RAIN_ENABLED"]
    return size, vehicles, d, with_manager, int(rng.integers(1, 10 ** 6))


def run_case(case, make_engines, ticks=45):
    size, vehicles, d, with_manager, seed = random_case(case)
    tb = dict(citygen.generate(size, size, seed=seed % 97 + 1))
    if with_manager:   # RainManager scheduled behind the clock agent (city_model.py:198-204 order: manager, then DTA)
        kinds = list(np.asarray(tb["schedule_kinds0"]))
        tb["schedule_kinds0"] = np.asarray(kinds[:-1] + [2] + kinds[-1:], dtype=np.int8)
    s, g, off, dirs = citygen.make_routes(tb, vehicles, seed=seed + 1, min_len=10, max_len=90)
    # trips that end where they start (they despawn inside the decide phase and the next vehicle of the decide order loses
    # its turn): some among the self-planning half, some with an (empty) route of their own on cells that other vehicles
    # occupy too, some added in the middle of the run; a separate stream so that the other draws of a case stay what they were
    rng2 = np.random.default_rng(5000 + case)
    n_sg = int(rng2.choice([0, 0, 1, 3, 8]))
    h = len(s) // 2
    g = np.array(g, copy=True)
    for i in rng2.integers(h, len(s), size=n_sg):
        g[i] = s[i]
    sg_cells = np.asarray(s)[rng2.integers(0, len(s), size=n_sg)].reshape(-1, 2)
    late_cells = np.asarray(s)[rng2.integers(0, len(s), size=n_sg)].reshape(-1, 2)
    apis = make_engines()
    for e in apis:
        build_engine(e, tb, defaults=d, global_seed=seed, sched_seed=seed + 7)
        # half the vehicles bring their route, the other half plan it themselves (spawn-time planner, path cache)
        e.add_vehicles_dirs(s[:h], g[:h], np.full(h, capi.POP["through"], np.int32), off[:h + 1], dirs[:off[h]])
        e.add_vehicles(s[h:], g[h:], np.full(len(s) - h, capi.POP["internal"], np.int32))
        if n_sg:
            e.add_vehicles_dirs(sg_cells, sg_cells, np.full(n_sg, capi.POP["through"], np.int32), np.zeros(n_sg + 1, np.int64),
                                np.zeros(0, np.uint8))
    a, b = apis
    ctx0 = f"case {case} ({size}x{size}, {len(s)} vehicles, {d['TRAFFIC_LIGHT_AGENT_ALGORITHM']})"
    for t in range(ticks):
        if n_sg and t == 12:
            for e in apis:
                e.add_vehicles(late_cells, late_cells, np.full(n_sg, capi.POP["internal"], np.int32))
        a.step(1), b.step(1)
        ctx = f"{ctx0} tick {t}"
        va, vb = a.vehicles(), b.vehicles()
        assert va.shape == vb.shape, f"{ctx}: live vehicles {va.shape} vs {vb.shape}"
        if not np.array_equal(va, vb):
            r, col = np.argwhere(va != vb)[0]
            raise AssertionError(f"{ctx}: vehicle row {r} field {capi.V_FIELDS[col]}: {va[r, col]} vs {vb[r, col]}; params {d}")
        for which in (capi.MAP_OCCUPANCY, capi.MAP_STOP, capi.MAP_STUCK, capi.MAP_RAIN):
            assert np.array_equal(a.map(which), b.map(which)), f"{ctx}: map {which}"
        assert np.array_equal(a.groups(), b.groups()), f"{ctx}: groups"
        assert a.rng_fingerprint(capi.RNG_GLOBAL) == b.rng_fingerprint(capi.RNG_GLOBAL), f"{ctx}: global RNG"
        assert a.rng_fingerprint(capi.RNG_SCHEDULER) == b.rng_fingerprint(capi.RNG_SCHEDULER), f"{ctx}: scheduler RNG"
        assert a.num_scheduled() == b.num_scheduled(), f"{ctx}: schedule size"
    ca, cb = a.counters(), b.counters()
    for f in ("stuck", "collisions", "malfunctions", "overtaking", "in_stuck_detour", "live_internal", "live_through",
              "count_completed_internal", "count_completed_through", "total_distance_internal", "astar_calls", "agent_steps"):
        assert getattr(ca, f) == getattr(cb, f), f"{ctx0}: counter {f}"
    for e in apis:
        e.close()


N_CASES = int(os.environ.get("TS_RANDOM_CASES", "20"))   # more for a bug hunt: TS_RANDOM_CASES=200 TS_RANDOM_TICKS=80
FIRST = int(os.environ.get("TS_RANDOM_FIRST", "0"))
N_TICKS = int(os.environ.get("TS_RANDOM_TICKS", "45"))


@pytest.mark.parametrize("case", range(FIRST, FIRST + N_CASES))
def test_hip_vs_oracle_random_config(case):
    from oracle import pyoracle
    from trafficsimulation_amd._lib import new_engine
    run_case(case, lambda: (new_engine(), pyoracle.load()), ticks=N_TICKS)
