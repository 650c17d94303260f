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
