"""GPU parity tests proper: the HIP engine, driven through the C-ABI, against (a) the golden
fixtures captured from the reference and (b) the CPU oracle on the same seeded inputs."""
import os

import numpy as np
import pytest

from trafficsimulation_amd import _capi as capi
from trafficsimulation_amd.world import load_trace
from tests.trace_util import CLOSED_TRACES, DEFAULT_TRACES, DTA_TRACES, RAIN_TRACES, RECT_TRACES, SERVICE_TRACES, VARIANT_TRACES, DESPAWN_TRACES, NOBATCH_TRACES, check_initial, replay_and_compare, setup_from_trace, trace_path

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hip():
    from trafficsimulation_amd._lib import new_engine
    api = new_engine()
    yield api
    api.close()


@pytest.mark.parametrize("name", CLOSED_TRACES + DTA_TRACES + RAIN_TRACES + SERVICE_TRACES + RECT_TRACES + DEFAULT_TRACES + VARIANT_TRACES + DESPAWN_TRACES + NOBATCH_TRACES)
def test_hip_reproduces_reference_trace(hip, name):
    """Every closed-population trace captured from the reference: car-following, the light controllers, the
    full replanning policy (GPU A*, phases 0-4), frequent strandings, sub-block roads, and the traffic generator
    spawning internal / through trips in the middle of the shuffled order (dta_*).  Vehicles are spawned without
    paths, so the spawn-time planner (cache + A*) is part of the check."""
    tr = load_trace(trace_path(name))
    setup_from_trace(hip, tr, explicit_paths=False)
    check_initial(hip, tr)
    assert hip.counters().astar_calls == int(tr["astar_calls_spawn"])
    n = replay_and_compare(hip, tr)
    assert n == len(tr["veh_off"]) - 1
    if "raised_at_tick" not in tr:   # (the tick in which the reference raised ran part of its searches)
        assert hip.counters().astar_calls == int(tr["astar_calls_spawn"]) + int(tr["astar_per_tick"].sum())


@pytest.mark.parametrize("name", ["faults_64_s9", "despawn_96_s25", "startgoal_96_s27", "service_heavy_96_s16", "rain_96_s14",
                                  "fov_96_s26", "carve_96_s10", "dta_96_s13"])
def test_hip_vs_oracle_without_batching(name):
    """PATHFINDING_BATCHING=False on the worlds and populations of the other traces (frequent strandings and sideswipes,
    stuck despawns, trips that end where they start, a busy service fleet, rain, the field-of-view mask, sub-block roads,
    the traffic generator): every vehicle decides at its own turn of the shuffled order.  The oracle's form of that switch
    is pinned by the nobatch_* traces captured from the reference (and a hunt against it); here it is the checker, state for state, every tick."""
    from oracle import pyoracle
    from trafficsimulation_amd._lib import new_engine
    tr = dict(load_trace(trace_path(name)))
    tr["defaults_json"] = {**tr["defaults_json"], "PATHFINDING_BATCHING": False}
    h, c = new_engine(), pyoracle.load()
    setup_from_trace(h, tr, explicit_paths=False)
    setup_from_trace(c, tr, explicit_paths=False)
    ticks = min(len(tr["veh_off"]) - 1, 150)
    for t in range(ticks):
        h.step(1)
        c.step(1)
        for which in (capi.MAP_OCCUPANCY, capi.MAP_STOP, capi.MAP_STUCK):
            assert np.array_equal(h.map(which), c.map(which)), f"tick {t}: map {which}"
        a, b = h.vehicles(), c.vehicles()
        assert a.shape == b.shape, f"tick {t}: live vehicles {a.shape} vs {b.shape}"
        if not np.array_equal(a, b):
            r, col = np.argwhere(a != b)[0]
            raise AssertionError(f"tick {t}: vehicle row {r} field {capi.V_FIELDS[col]}: hip {a[r, col]} cpu {b[r, col]}")
        assert np.array_equal(h.groups(), c.groups()), f"tick {t}: light groups"
        assert h.rng_fingerprint(capi.RNG_GLOBAL) == c.rng_fingerprint(capi.RNG_GLOBAL), f"tick {t}: RNG"
        assert h.rng_fingerprint(capi.RNG_SCHEDULER) == c.rng_fingerprint(capi.RNG_SCHEDULER)
    ch, cc = h.counters(), c.counters()
    for f in ("stuck", "parked", "collisions", "malfunctions", "overtaking", "in_stuck_detour", "live_internal", "live_through",
              "count_completed_internal", "count_completed_through", "errored_internal", "errored_through", "agent_steps",
              "astar_calls", "elapsed", "step_count"):
        assert getattr(ch, f) == getattr(cc, f), f
    h.close()
    c.close()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_astar_kats(hip, golden_dir, tag):
    """The pathfinder operator seam (ts_astar) against the reference's astar_numba on 260 queries per map:
    strict / soft / contraflow / step-limited, failures included."""
    k = np.load(os.path.join(golden_dir, "astar_kats.npz"))
    hip.create(k[f"{tag}_allowed_dirs_map"], k[f"{tag}_is_road_map"], k[f"{tag}_road_type_map"],
               k[f"{tag}_intersection_map"], hip.default_params())
    hip.debug_set_occupancy(k[f"{tag}_occupancy_map"])
    hip.upload_map(capi.MAP_STOP, k[f"{tag}_stop_map"])
    q, off, xy = k[f"{tag}_queries"], k[f"{tag}_path_off"], k[f"{tag}_path_xy"]
    for i, (sx, sy, gx, gy, soft, ign, maxs) in enumerate(q):
        got = hip.astar(int(sx), int(sy), int(gx), int(gy), bool(soft), bool(ign), int(maxs))
        assert np.array_equal(got, xy[off[i]:off[i + 1]]), f"query {i}: {q[i]}"


@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_astar_fov_kats(golden_dir, tag):
    """ts_astar with VEHICLE_RESPECT_AWARENESS (field-of-view masking) against the reference's astar_numba."""
    from tests.test_oracle_kats import run_astar_fov_kats
    from trafficsimulation_amd._lib import new_engine
    run_astar_fov_kats(new_engine, golden_dir, tag)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_pathfinder_operator_signature(golden_dir, tag):
    """trafficsimulation_amd.pathfinding.astar_hip - the reference's `astar(...)` operator signature
    (astar_numba.py:243-256) - bound to the HIP library (its default engine factory), on the reference's A* KATs."""
    from trafficsimulation_amd import pathfinding
    k = np.load(os.path.join(golden_dir, "astar_kats.npz"))
    H, W = k[f"{tag}_is_road_map"].shape
    maps = dict(occupancy_map=k[f"{tag}_occupancy_map"], stop_map=k[f"{tag}_stop_map"], is_road_map=k[f"{tag}_is_road_map"],
                road_type_map=k[f"{tag}_road_type_map"], allowed_dirs_map=k[f"{tag}_allowed_dirs_map"])
    q, off, xy = k[f"{tag}_queries"], k[f"{tag}_path_off"], k[f"{tag}_path_xy"]
    try:
        for i, (sx, sy, gx, gy, soft, ign, maxs) in enumerate(q[::3]):
            got = pathfinding.astar_hip(W, H, int(sx), int(sy), int(gx), int(gy), respect_awareness=False, awareness_range=10,
                                        density_map=k[f"{tag}_density32"], soft_obstacles=bool(soft), ignore_flow=bool(ign),
                                        maximum_steps=int(maxs), **maps)
            j = 3 * i
            assert got == [tuple(p) for p in xy[off[j]:off[j + 1]].tolist()], f"query {j}: {q[j]}"
        assert len(pathfinding._cache) == 1
        pathfinding.astar_hip(W, H, 1, 1, 2, 2, respect_awareness=True, awareness_range=10, density_map=None,
                              soft_obstacles=False, ignore_flow=False, **maps)
        assert len(pathfinding._cache) == 2          # field-of-view masking is an engine parameter: its own instance
    finally:
        pathfinding.release()


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_hip_density_matches_scipy(hip, golden_dir, tag):
    k = np.load(os.path.join(golden_dir, "density_kats.npz"))
    road, occ, want = k[f"{tag}_road"], k[f"{tag}_occ"], k[f"{tag}_density"]
    z = np.zeros_like(road)
    hip.create(z.astype(np.uint8), road, z, z, hip.default_params())
    hip.debug_set_occupancy(occ)
    got = hip.density()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _pair(size, vehicles, seed, policy):
    """HIP engine and CPU oracle on the same synthetic world / routes / seeds."""
    import bench
    from oracle import pyoracle
    from trafficsimulation_amd._lib import new_engine
    tables, routes, _ = bench.make_workload(size, vehicles, seed)
    hip_api, cpu_api = new_engine(), pyoracle.load()
    bench.setup(hip_api, tables, routes, seed, extra=policy)
    bench.setup(cpu_api, tables, routes, seed, extra=policy)
    return hip_api, cpu_api


def _compare(hip_api, cpu_api, ticks, every=1):
    for t in range(ticks):
        hip_api.step(1)
        cpu_api.step(1)
        if (t + 1) % every and t != ticks - 1:
            continue
        for which in (capi.MAP_OCCUPANCY, capi.MAP_STOP, capi.MAP_STUCK):
            assert np.array_equal(hip_api.map(which), cpu_api.map(which)), f"tick {t}: map {which}"
        a, b = hip_api.vehicles(), cpu_api.vehicles()
        assert a.shape == b.shape, f"tick {t}: live vehicles {a.shape} vs {b.shape}"
        if not np.array_equal(a, b):
            r, c = np.argwhere(a != b)[0]
            raise AssertionError(f"tick {t}: vehicle row {r} field {capi.V_FIELDS[c]}: hip {a[r, c]} cpu {b[r, c]}")
        assert np.array_equal(hip_api.groups(), cpu_api.groups()), f"tick {t}: light groups"
        assert hip_api.rng_fingerprint(capi.RNG_GLOBAL) == cpu_api.rng_fingerprint(capi.RNG_GLOBAL), f"tick {t}: RNG"
        assert hip_api.rng_fingerprint(capi.RNG_SCHEDULER) == cpu_api.rng_fingerprint(capi.RNG_SCHEDULER)
    ch, cc = hip_api.counters(), cpu_api.counters()
    for f in ("stuck", "live_through", "count_completed_through", "total_distance_through", "agent_steps",
              "total_duration_through", "elapsed", "step_count"):
        assert getattr(ch, f) == getattr(cc, f), f
    hip_api.close()
    cpu_api.close()


def test_hip_vs_oracle_carfollow_512():
    """Config 2 of BASELINE.json: 512x512, 50k vehicles, lights fixed green (disabled), car-following only."""
    h, c = _pair(512, 50_000, 3, {})
    _compare(h, c, 40)


def test_hip_vs_oracle_lights_512():
    """Same world with QUEUE_ACTUATED lights interleaved in the shuffled order (replans still gated off)."""
    h, c = _pair(512, 20_000, 4, {"TRAFFIC_LIGHT_AGENT_ALGORITHM": "QUEUE_ACTUATED"})
    _compare(h, c, 60)


def test_hip_vs_oracle_full_policy_256():
    """Defaults (queue-actuated lights, replans, contraflow) with strandings made frequent, on a synthetic
    256x256 world: the host-scan fix-up path, blocker-triggered replans and overtakes all fire."""
    pol = {"TRAFFIC_LIGHT_AGENT_ALGORITHM": "QUEUE_ACTUATED", "PATHFINDING_COOLDOWN": 5,
           "VEHICLE_STUCK_RECOMPUTE_THRESHOLD": 30, "VEHICLE_STUCK_RECOMPUTE_THRESHOLD_INTERSECTION": 1,
           "VEHICLE_CONTRAFLOW_OVERTAKE_ACTIVE": True, "VEHICLE_STUCK_CONTRAFLOW_ENABLED": True,
           "VEHICLE_SIDESWIPE_COLLISION_CHANCE": 0.05, "VEHICLE_SIDESWIPE_COLLISION_DURATION": 30,
           "VEHICLE_MALFUNCTION_CHANCE": 0.001, "VEHICLE_MALFUNCTION_DURATION": 25}
    h, c = _pair(256, 2_000, 5, pol)
    hc = None
    for t in range(40):
        h.step(1)
        c.step(1)
        a, b = h.vehicles(), c.vehicles()
        assert a.shape == b.shape, f"tick {t}"
        if not np.array_equal(a, b):
            r, col = np.argwhere(a != b)[0]
            raise AssertionError(f"tick {t}: vehicle row {r} field {capi.V_FIELDS[col]}: hip {a[r, col]} cpu {b[r, col]}")
        assert np.array_equal(h.map(capi.MAP_OCCUPANCY), c.map(capi.MAP_OCCUPANCY)), f"tick {t}"
        assert np.array_equal(h.map(capi.MAP_STOP), c.map(capi.MAP_STOP)), f"tick {t}"
        assert h.rng_fingerprint(capi.RNG_GLOBAL) == c.rng_fingerprint(capi.RNG_GLOBAL), f"tick {t}: RNG"
    hc, cc = h.counters(), c.counters()
    assert hc.rng_fixups > 0 and hc.astar_calls == cc.astar_calls and hc.astar_calls > 100
    # searches whose target the reachability pass proved unreachable are answered [] without flooding
    assert 0 < hc.astar_expansions <= cc.astar_expansions
    assert (hc.overtaking, hc.in_stuck_detour, hc.collisions, hc.malfunctions) == (
        cc.overtaking, cc.in_stuck_detour, cc.collisions, cc.malfunctions)
    h.close()
    c.close()


def _pair_full(size, vehicles, seed, extra=None, carves=False):
    """HIP engine and CPU oracle under the reference's default policy (bench.py --policy full: QUEUE_ACTUATED lights,
    replanning, contraflow, malfunctions / sideswipes) on the same synthetic world / routes / seeds."""
    import bench
    from oracle import pyoracle
    from trafficsimulation_amd._lib import new_engine
    tables, routes, _ = bench.make_workload(size, vehicles, seed, carves=carves)
    hip_api, cpu_api = new_engine(), pyoracle.load()
    bench.setup(hip_api, tables, routes, seed, extra=extra, policy="full")
    bench.setup(cpu_api, tables, routes, seed, extra=extra, policy="full")
    return hip_api, cpu_api


def _compare_full(h, c, ticks, every=1):
    for t in range(ticks):
        h.step(1)
        c.step(1)
        if (t + 1) % every and t != ticks - 1:
            continue
        for which in (capi.MAP_OCCUPANCY, capi.MAP_STOP, capi.MAP_STUCK):
            assert np.array_equal(h.map(which), c.map(which)), f"tick {t}: map {which}"
        a, b = h.vehicles(), c.vehicles()
        assert a.shape == b.shape, f"tick {t}: live vehicles {a.shape} vs {b.shape}"
        if not np.array_equal(a, b):
            r, col = np.argwhere(a != b)[0]
            raise AssertionError(f"tick {t}: vehicle row {r} field {capi.V_FIELDS[col]}: hip {a[r, col]} cpu {b[r, col]}")
        assert np.array_equal(h.groups(), c.groups()), f"tick {t}: light groups"
        assert h.rng_fingerprint(capi.RNG_GLOBAL) == c.rng_fingerprint(capi.RNG_GLOBAL), f"tick {t}: RNG"
        assert h.rng_fingerprint(capi.RNG_SCHEDULER) == c.rng_fingerprint(capi.RNG_SCHEDULER)
        assert h.counters().astar_calls == c.counters().astar_calls, f"tick {t}: A* calls"
    ch, cc = h.counters(), c.counters()
    for f in ("stuck", "live_through", "count_completed_through", "total_distance_through", "agent_steps", "overtaking",
              "in_stuck_detour", "collisions", "malfunctions", "astar_calls", "elapsed", "step_count"):
        assert getattr(ch, f) == getattr(cc, f), f
    h.close()
    c.close()
    return ch


def test_hip_vs_oracle_full_policy_512_with_carved_blocks():
    """The default policy on a synthetic world with sub-block roads / L-shaped carves (BASELINE config 5's kind of world; the
    reference's own carved city is the trace carve_96_s10): the searches route through the one-lane roads inside the blocks,
    through a replanning wave."""
    h, c = _pair_full(512, 12_000, 9, carves=True)
    ch = _compare_full(h, c, 9)
    assert ch.astar_calls > 5_000


def test_hip_vs_oracle_full_policy_512_through_a_replanning_wave():
    """Reference defaults on 512 x 512 / 12 000 vehicles across the tick in which every vehicle's path-retry cooldown
    runs out at once (thousands of searches in one tick: the work queue is far longer than the searcher slots, the
    path pool is garbage-collected / grown under it).  State for state against the oracle, every tick."""
    h, c = _pair_full(512, 12_000, 7)
    ch = _compare_full(h, c, 9)
    assert ch.astar_calls > 5_000 and ch.astar_expansions > 1_000_000


def test_hip_vs_oracle_full_policy_1024():
    """Reference defaults at 1024 x 1024 / 60 000 vehicles (searches of 10^4 - 10^5 expansions: the direct-indexed tables
    of a megacell map, searcher slots in the tens of MB), four ticks, every tick."""
    h, c = _pair_full(1024, 60_000, 1)
    ch = _compare_full(h, c, 4)
    assert ch.astar_calls > 1_000 and ch.astar_expansions > 5_000_000


def test_hip_vs_oracle_replanning_wave_with_a_full_path_pool(monkeypatch):
    """The same wave with (almost) no room reserved in the path pool: planners find it full, their entries go to the retry
    list, the host garbage-collects / grows the pool and the searches run again - state for state against the oracle."""
    monkeypatch.setenv("TS_DEBUG_POOL_PER_ENTRY", "2")
    h, c = _pair_full(512, 12_000, 9)
    ch = _compare_full(h, c, 8)
    assert ch.astar_calls > 5_000


def test_hip_vs_oracle_full_policy_768_through_a_replanning_wave():
    """The same wave one size up (768 x 768 / 30 000 vehicles: tens of thousands of searches of ~10^4 expansions in one
    tick), compared after every second tick."""
    h, c = _pair_full(768, 30_000, 3)
    ch = _compare_full(h, c, 7, every=2)
    assert ch.astar_calls > 20_000


def test_hip_vs_oracle_config3_2048_500k():
    """BASELINE config 3 at its own size: 2048 x 2048, 500 000 vehicles, QUEUE_ACTUATED light groups AND replanning
    (reference defaults), three ticks, compared state for state after the third."""
    h, c = _pair_full(2048, 500_000, 1)
    ch = _compare_full(h, c, 3, every=3)
    assert ch.astar_calls > 1_000


def test_hip_vs_oracle_full_policy_4096_1m_first_ticks():
    """bench.py's default workload itself - 4096 x 4096, 10^6 vehicles, the reference's default policy - for its first four
    ticks (the replans start with the third: some 2.5 x 10^4 searches on 4096 searcher slots with 32 MB tables each),
    state for state after the second and the fourth.  (A
    replanning wave of this size is hours of oracle time; waves are compared at 512^2 .. 1024^2 above.)"""
    h, c = _pair_full(4096, 1_000_000, 1)
    ch = _compare_full(h, c, 4, every=2)
    assert ch.astar_calls > 5_000


def test_hip_vs_oracle_full_size_4096_1m():
    """BASELINE.json's headline size (4096x4096, 10^6 vehicles), 6 ticks, compared state for state."""
    h, c = _pair(4096, 1_000_000, 1, {})
    _compare(h, c, 6, every=3)


def test_hip_vs_oracle_config5_size_8192_4m():
    """BASELINE config 5's size and kind of world on ONE GPU: 8192 x 8192 with sub-block roads / L-shaped carves
    (citygen's form of city_model.py:563-737), 4 x 10^6 vehicles + 3.3 x 10^5 QUEUE_ACTUATED light groups (more
    than 2^22 scheduled agents: 24-bit ranks in the claim words), replans gated off; the decide phase runs in four passes
    over the MT19937 word ring.  Three ticks, compared state for state after the last."""
    import bench
    from oracle import pyoracle
    from trafficsimulation_amd._lib import new_engine
    tables, routes, _ = bench.make_workload(8192, 4_000_000, 1, carves=True)      # (L-shaped one-lane roads inside the large blocks)
    assert int(tables["carved_blocks"]) > 10_000
    h, c = new_engine(), pyoracle.load()
    bench.setup(h, tables, routes, 1, policy="lights")
    bench.setup(c, tables, routes, 1, policy="lights")
    assert h.num_scheduled() > (1 << 22)
    _compare(h, c, 3, every=3)


def test_facade_on_hip(hip):
    """The Mesa-shaped facade (CityModel / VehicleAgent / grid / schedule) over the HIP engine."""
    from tests.test_mesa_facade import run_facade_against_trace
    run_facade_against_trace(hip)


def test_facade_with_generator_on_hip(hip):
    """The same facade with the engine's traffic generator armed: generator-spawned vehicles, service vehicles and
    CityBlock views read from the HIP engine."""
    from tests.test_mesa_facade import run_facade_with_generator
    run_facade_with_generator(hip)


def test_hip_vs_oracle_service_stress():
    """Service vehicles under stress on a reference-generated world whose block types are overridden so that a single
    block needs food: a vehicle that still carries load is sent back to the cell it stands on (start == goal, empty
    path) and runs on_target_reached -> _start_service inside step_decide (k_decide_arrive / seen_parked); a huge
    fleet with tiny loads keeps several vehicles parked at once and contests the ranked service cells.  HIP engine
    against the CPU oracle, state for state, up to and including the duplicate-id error both must report."""
    from oracle import pyoracle
    from trafficsimulation_amd._lib import new_engine
    from trafficsimulation_amd.world import build_engine
    tr = load_trace(trace_path("service_64_s15"))
    tables = dict(tr)
    tables["blk_type"] = np.asarray([0, 2, 0, 0], dtype=np.int32)
    svc = dict(service_food=4000, service_waste=4000, load_time=3, max_load_food=5000.0, max_load_waste=3.0,
               food_consumption_ticks=2, waste_production_ticks=3)
    apis = []
    for api in (new_engine(), pyoracle.load()):
        build_engine(api, tables, defaults={"RAIN_ENABLED": False, "VEHICLE_MALFUNCTION_CHANCE": 0.002,
                                            "VEHICLE_MALFUNCTION_DURATION": 15}, global_seed=77, sched_seed=78)
        api.set_traffic_generator(tables, internal_per_day=2000, passing_per_day=3000, start_offset_seconds=6 * 3600, service=svc)
        apis.append(api)
    h, c = apis
    own_cell_retargets, raised = 0, None
    for t in range(400):
        errs = []
        for api in (h, c):
            try:
                api.step(1)
                errs.append(None)
            except capi.EngineError as ex:
                errs.append(ex.code)
        assert errs[0] == errs[1], f"tick {t}: hip {errs[0]} vs oracle {errs[1]}"
        if errs[0] is not None:
            raised = (t, errs[0])
            break
        a, b = h.vehicles(), c.vehicles()
        assert a.shape == b.shape, f"tick {t}: live vehicles"
        if not np.array_equal(a, b):
            r, col = np.argwhere(a != b)[0]
            raise AssertionError(f"tick {t}: vehicle row {r} field {capi.V_FIELDS[col]}: hip {a[r, col]} cpu {b[r, col]}")
        ma, mb = h.vehicle_meta(), c.vehicle_meta()
        assert np.array_equal(ma, mb), f"tick {t}: vehicle meta"
        own_cell_retargets += int(np.sum((ma[:, 5] == 0) & (ma[:, 2] == a[:, 1]) & (ma[:, 3] == a[:, 2])))
        for x, y in zip(h.service_vehicles(), c.service_vehicles()):
            assert np.array_equal(x, y), f"tick {t}: service vehicle loads / blocks"
        assert np.array_equal(h.blocks(), c.blocks()), f"tick {t}: block stock"
        for which in (capi.MAP_OCCUPANCY, capi.MAP_STOP, capi.MAP_STUCK):
            assert np.array_equal(h.map(which), c.map(which)), f"tick {t}: map {which}"
        assert h.rng_fingerprint(capi.RNG_GLOBAL) == c.rng_fingerprint(capi.RNG_GLOBAL), f"tick {t}: RNG"
        ch, cc = h.counters(), c.counters()
        for f in ("parked", "live_through", "count_completed_through", "total_distance_through", "malfunctions",
                  "created_service_food", "created_service_waste", "live_service_food", "live_service_waste", "astar_calls"):
            assert getattr(ch, f) == getattr(cc, f), f"tick {t}: {f}"
    assert own_cell_retargets > 20, "the scenario is built to exercise arrivals inside the decide phase"
    assert raised is not None and raised[1] == capi.TS_E_UNSUPPORTED
    h.close()
    c.close()


def test_hip_multi_pass_decide_in_a_subprocess():
    """More vehicles than one RNG pass covers (2^20) split the decide phase into several passes, each resuming the
    stream bookkeeping where the previous one stopped.  TS_DEBUG_SEG shrinks the pass size so that a 256x256 world
    walks that path (dozens of passes per tick, with strandings firing inside them); the knob is read once per
    process, hence the subprocess."""
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, '.')\n"
        "from tests.test_gpu_parity import _pair, _compare\n"
        "pol = {'TRAFFIC_LIGHT_AGENT_ALGORITHM': 'QUEUE_ACTUATED', 'VEHICLE_MALFUNCTION_CHANCE': 0.002,\n"
        "       'VEHICLE_MALFUNCTION_DURATION': 12, 'VEHICLE_SIDESWIPE_COLLISION_CHANCE': 0.05,\n"
        "       'VEHICLE_SIDESWIPE_COLLISION_DURATION': 9, 'PATHFINDING_COOLDOWN': 4}\n"
        "h, c = _pair(256, 6000, 11, pol)\n"
        "_compare(h, c, 30)\n"
        "print('multi-pass ok')\n")
    env = dict(os.environ, TS_DEBUG_SEG="777")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "multi-pass ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_hip_vs_oracle_long_run_word_ring_wraps():
    """400 ticks at 512x512 / 50k vehicles consume ~50M words of the global stream and ~20M of the scheduler stream:
    both pre-generated word rings (8M words each) and the device mirror wrap several times, the move-phase claim
    epochs wrap too."""
    h, c = _pair(512, 50_000, 21, {"TRAFFIC_LIGHT_AGENT_ALGORITHM": "QUEUE_ACTUATED"})
    _compare(h, c, 400, every=25)



@pytest.mark.parametrize("name", ["full_96_s8", "default_200_s20", "faults_64_s9", "startgoal_96_s27"])
def test_hip_quad_searcher_reproduces_reference_trace(monkeypatch, name):
    """k_replan_quad (astar_quad.h: sixteen searches per wave; by default only for queues of 262 144 entries and more) with k_replan beside it on the hand-backs,
    forced on for every tick of a captured run: the same per-tick comparison against the reference's recorded state."""
    from trafficsimulation_amd._lib import new_engine
    monkeypatch.setenv("TS_QUAD", "1")
    monkeypatch.setenv("TS_QUAD_MIN", "1")
    api = new_engine()
    try:
        tr = load_trace(trace_path(name))
        setup_from_trace(api, tr, explicit_paths=False)
        n = replay_and_compare(api, tr)
        assert n == len(tr["veh_off"]) - 1
        assert api.counters().astar_calls == int(tr["astar_calls_spawn"]) + int(tr["astar_per_tick"].sum())
    finally:
        api.close()


def test_hip_quad_searcher_vs_oracle_512_through_a_replanning_wave(monkeypatch):
    """... and against the oracle on 512 x 512 / 12 000 vehicles through the first replanning wave (heaps beyond the quads'
    LDS share, hand-backs to k_replan while both kernels run, the direction window moving), state for state every tick."""
    monkeypatch.setenv("TS_QUAD", "1")
    monkeypatch.setenv("TS_QUAD_MIN", "1")
    h, c = _pair_full(512, 12_000, 7)
    ch = _compare_full(h, c, 9)
    assert ch.astar_calls > 5_000 and ch.astar_expansions > 1_000_000


def test_hip_quad_and_wave_searchers_share_the_table_arena(monkeypatch):
    """TS_QUAD=1 as it runs by default thresholds, scaled down: the replanning wave of a 768 x 768 world goes to the quads
    (TS_QUAD_MIN=5000), the ticks before and after it to k_replan on all its slots - and the quads' tables alias k_replan's
    (1024 quad slots fit behind the side waves' slots here), so the arena is cleared and changes hands twice inside the run."""
    monkeypatch.setenv("TS_QUAD", "1")
    monkeypatch.setenv("TS_QUAD_MIN", "5000")
    monkeypatch.setenv("TS_QUAD_SLOTS", "1024")
    h, c = _pair_full(768, 30_000, 3)
    ch = _compare_full(h, c, 9, every=1)
    assert ch.astar_calls > 20_000


@pytest.mark.parametrize("name", ["dta_64_s12", "dta_96_s13", "config1_64_s11"])
def test_hip_cached_stats_match_the_reference(hip, name):
    """DynamicTrafficAgent._update_cached_stats on the device (k_live_stats at the generator's place in the shuffled order,
    every STATISTICS_UPDATE_INTERVAL ticks) + the host-side daily keys: every key of the reference's dict, every tick."""
    from tests.trace_util import replay_and_compare_cached_stats
    tr = load_trace(trace_path(name))
    setup_from_trace(hip, tr, explicit_paths=False)
    assert replay_and_compare_cached_stats(hip, tr, name) >= 9


def test_hip_quad_searcher_with_a_full_path_pool(monkeypatch):
    """The quads' commit finds the path pool full (TS_DEBUG_POOL_PER_ENTRY=2): their vehicles go to the retry list like k_replan's,
    the host makes room and k_replan runs them again - state for state against the oracle."""
    monkeypatch.setenv("TS_QUAD", "1")
    monkeypatch.setenv("TS_QUAD_MIN", "1")
    monkeypatch.setenv("TS_DEBUG_POOL_PER_ENTRY", "2")
    h, c = _pair_full(512, 12_000, 9)
    ch = _compare_full(h, c, 8)
    assert ch.astar_calls > 5_000
