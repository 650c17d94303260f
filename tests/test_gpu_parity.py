"""GPU parity tests proper: the HIP engine, driven through the C-ABI, against (a) the golden
fixtures captured from the reference and (b) the CPU oracle on the same seeded inputs."""
import os

import numpy as np
import pytest

from trafficsimulation_amd import _capi as capi
from trafficsimulation_amd.world import load_trace
from tests.trace_util import NO_ASTAR_TRACES, check_initial, replay_and_compare, setup_from_trace, trace_path

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hip():
    from trafficsimulation_amd._lib import new_engine
    api = new_engine()
    yield api
    api.close()


@pytest.mark.parametrize("name", NO_ASTAR_TRACES)
def test_hip_reproduces_reference_trace(hip, name):
    tr = load_trace(trace_path(name))
    setup_from_trace(hip, tr, explicit_paths=True)
    check_initial(hip, tr)
    n = replay_and_compare(hip, tr)
    assert n == len(tr["veh_off"]) - 1


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


def test_hip_faults_request_replan_loudly():
    """Malfunctions and sideswipes made frequent: the host scan finds events (fix-up path), and the first
    vehicle blocked by a stranded one asks for a replan - which must surface as TS_E_UNSUPPORTED until the
    GPU A* exists, never as a silent CPU fallback."""
    h, c = _pair(256, 3_000, 5, {"VEHICLE_SIDESWIPE_COLLISION_CHANCE": 0.3, "VEHICLE_SIDESWIPE_COLLISION_DURATION": 5,
                                  "VEHICLE_MALFUNCTION_CHANCE": 0.002, "VEHICLE_MALFUNCTION_DURATION": 4,
                                  "PATHFINDING_COOLDOWN": 10 ** 9})
    with pytest.raises(capi.EngineError) as ei:
        h.step(30)
    assert ei.value.code == capi.TS_E_UNSUPPORTED
    assert h.counters().rng_fixups > 0
    h.close()
    c.close()


def test_hip_vs_oracle_full_size_4096_1m():
    """BASELINE.json's headline size (4096x4096, 10^6 vehicles), 6 ticks, compared state for state."""
    h, c = _pair(4096, 1_000_000, 1, {})
    _compare(h, c, 6, every=3)
