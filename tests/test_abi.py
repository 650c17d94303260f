"""The C-ABI libraries load and export every symbol include/trafficsim.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "trafficsim.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("ts_create", "ts_destroy", "ts_set_lights", "ts_schedule_add", "ts_seed", "ts_seed_int",
                 "ts_add_vehicles", "ts_upload_map", "ts_step", "ts_download_map", "ts_download_vehicles",
                 "ts_download_groups", "ts_counters", "ts_astar", "ts_last_error", "ts_default_params"):
        assert must in syms


def test_hip_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build_hip()
    from trafficsimulation_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} missing from libtrafficsim_hip.so"


def test_oracle_library_exports_the_same_surface():
    from oracle import pyoracle
    lib = ctypes.CDLL(pyoracle.build())
    for s in declared_symbols():
        assert hasattr(lib, "tso_" + s[3:]), f"tso_{s[3:]} missing from libtso.so"


def test_params_struct_layout_matches_header_defaults():
    """ctypes TsParams mirrors the C struct: defaults written by C read back correctly in Python."""
    from oracle import pyoracle
    api = pyoracle.load()
    p = api.default_params()
    assert (p.vehicle_min_speed, p.vehicle_max_speed, p.vehicle_awareness_range) == (1, 5, 10)
    assert p.malfunction_chance == 1e-7 and p.sideswipe_chance == 1e-9
    assert (p.road_type_penalty_r1, p.road_type_penalty_r2, p.road_type_penalty_r3) == (0.5, 5.0, 50.0)
    assert p.light_algorithm == 2 and p.qa_gap == 3 and p.time_per_step_seconds == 6
    assert p.contraflow_penalty == 5000 and p.dynamic_penalty_scale == 4.0


def test_product_fails_loudly_without_gpu():
    import numpy as np
    from trafficsimulation_amd import _capi
    from trafficsimulation_amd._lib import new_engine
    e = new_engine()
    z = np.zeros((8, 8), np.int8)
    try:
        e.create(z.astype(np.uint8), z, z, z, e.default_params())
    except _capi.EngineError as ex:
        assert ex.code == _capi.TS_E_DEVICE
    else:
        e.close()
        pytest.skip("a GPU is present: the engine came up")
