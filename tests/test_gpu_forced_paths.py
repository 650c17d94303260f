"""The A* paths an ordinary workload only reaches by accident, forced.

`libtrafficsim_hip_smallheap.so` is the engine built with -DTS_LDS_HEAP=128 -DTS_DEBUG_STAMP_MAX=300u (csrc/Makefile): with
128 heap slots in LDS every search of a few hundred expansions runs the HBM-spill form of the loop (astar_loop<true, ...>)
and the hand-overs between the two forms (astar.h: AL_SWITCH), and a searcher's table epoch wraps - table cleared by the
wave, astar.h next_epoch - after 300 searches instead of 262 143.  The same fixtures and oracle comparisons as
tests/test_gpu_parity.py run on it; one more test runs a replanning wave with a single searcher slot."""
import ctypes
import os

import numpy as np
import pytest

from trafficsimulation_amd import _capi as capi
from trafficsimulation_amd import _lib
from tests import test_gpu_parity as P

pytestmark = pytest.mark.gpu

SMALL = os.path.join(os.path.dirname(_lib.LIB_PATH), "libtrafficsim_hip_smallheap.so")


def small_engine():
    if not os.path.exists(SMALL):
        raise _lib.EngineUnavailable(f"{SMALL} is missing - `make -C trafficsimulation_amd/csrc` builds it")
    return capi.CApi(ctypes.CDLL(SMALL), "ts_")


@pytest.fixture()
def hip_small():
    api = small_engine()
    yield api
    api.close()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_smallheap_astar_kats(hip_small, golden_dir, tag):
    """520 reference queries through ts_astar: one searcher slot, so its epoch wraps (and the table is cleared) inside the run."""
    P.test_hip_astar_kats(hip_small, golden_dir, tag)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_smallheap_astar_fov_kats(golden_dir, tag):
    from tests.test_oracle_kats import run_astar_fov_kats
    run_astar_fov_kats(small_engine, golden_dir, tag)


@pytest.mark.parametrize("name", ["full_96_s8", "default_200_s20"])
def test_smallheap_reproduces_reference_trace(hip_small, name):
    P.test_hip_reproduces_reference_trace(hip_small, name)


def test_smallheap_vs_oracle_512_through_a_replanning_wave(monkeypatch):
    """A replanning wave (thousands of searches, heaps of several hundred entries) almost entirely in the spill form."""
    monkeypatch.setattr(_lib, "new_engine", small_engine)
    h, c = P._pair_full(512, 12_000, 7)
    ch = P._compare_full(h, c, 9)
    assert ch.astar_calls > 5_000 and ch.astar_expansions > 1_000_000


def test_one_searcher_slot_serves_a_whole_wave(monkeypatch):
    """TS_ASTAR_SLOTS=1: the whole queue of a replanning wave on one searcher (one table, one epoch counter, every class
    list drained by a single wave)."""
    monkeypatch.setenv("TS_ASTAR_SLOTS", "1")
    h, c = P._pair_full(512, 12_000, 7)
    ch = P._compare_full(h, c, 7)
    assert ch.astar_calls > 3_000
