"""ctypes binding of the C-ABI declared in include/trafficsim.h.

`CApi(lib, prefix)` wraps one shared library exporting `<prefix>create`, `<prefix>step`, ...
The product uses `prefix="ts_"` with trafficsimulation_amd/csrc/libtrafficsim_hip.so (see
`_lib.py`); tests additionally instantiate it over the CPU oracle (`tso_`, oracle/libtso.so) so
that both sides of a parity check are driven through identical host code.
"""
from __future__ import annotations

import ctypes as C
import zlib
from typing import Dict, Optional, Sequence

import numpy as np

# ---- enums (mirror include/trafficsim.h) ----------------------------------------------------
TS_OK, TS_E_INVALID, TS_E_STATE, TS_E_DEVICE, TS_E_UNSUPPORTED, TS_E_CAPACITY = 0, -1, -2, -3, -4, -5
LIGHT_ALGORITHMS = {
    "DISABLED": 0, "FIXED_TIME": 1, "QUEUE_ACTUATED": 2, "PRESSURE_CONTROL": 3,
    "NEIGHBOR_PRESSURE_CONTROL": 4, "NEIGHBOR_GREEN_WAVE": 5,
}
M_FIELDS = ["spawn_idx", "population", "target_x", "target_y", "vehicle_type", "service_phase"]
TRIP_SERVICE_FOOD, TRIP_SERVICE_WASTE = 3, 4
AGENT_LIGHT_GROUP, AGENT_NOOP, AGENT_RAIN_MANAGER, AGENT_CLOCK, AGENT_CITY_BLOCK = 0, 1, 2, 3, 6
MAP_OCCUPANCY, MAP_STOP, MAP_STUCK, MAP_RAIN = 0, 1, 2, 3
RNG_GLOBAL, RNG_SCHEDULER = 0, 1
POP = {"undefined": 0, "internal": 1, "through": 2}
V_FIELDS = ["spawn_idx", "x", "y", "base_speed", "current_speed", "max_steps", "direction", "stuck_ticks",
            "cooldown", "flags", "stranded_left", "steps_traveled", "path_len", "path_crc", "overtake_dur",
            "detour_dur"]
G_FIELDS = ["current_phase", "pending_phase", "queue_timer", "gap_timer", "last_arrival", "fixed_time_timer",
            "ft_phase", "ns_pressure", "ew_pressure"]
F_EARLY_EXIT, F_STUCK, F_PARKED, F_COLLISION, F_MALFUNCTION, F_OVERTAKING, F_DETOUR, F_BLOCKED, F_HAS_PREV = (
    1 << i for i in range(9))


class TsParams(C.Structure):
    _fields_ = [
        ("vehicle_min_speed", C.c_int32), ("vehicle_max_speed", C.c_int32),
        ("vehicle_awareness_range", C.c_int32), ("rain_enabled", C.c_int32),
        ("rain_speed_reduction", C.c_int32), ("pathfinding_cooldown", C.c_int32),
        ("pathfinding_cache", C.c_int32), ("stuck_recompute_threshold", C.c_int32),
        ("stuck_recompute_threshold_intersection", C.c_int32), ("contraflow_overtake_active", C.c_int32),
        ("max_contraflow_overtake_steps", C.c_int32), ("contraflow_overtake_duration", C.c_int32),
        ("stuck_contraflow_enabled", C.c_int32), ("stuck_contraflow_threshold", C.c_int32),
        ("stuck_contraflow_threshold_intersection", C.c_int32), ("max_contraflow_stuck_detour_steps", C.c_int32),
        ("contraflow_stuck_detour_duration", C.c_int32), ("malfunction_active", C.c_int32),
        ("malfunction_duration", C.c_int32), ("sideswipe_active", C.c_int32), ("sideswipe_duration", C.c_int32),
        ("malfunction_chance", C.c_double), ("sideswipe_chance", C.c_double),
        ("contraflow_penalty", C.c_int32), ("obstacle_penalty_vehicle", C.c_int32),
        ("obstacle_penalty_stop", C.c_int32), ("road_type_penalties_enabled", C.c_int32),
        ("turn_penalty_enabled", C.c_int32), ("turn_penalty", C.c_int32),
        ("dynamic_penalties_enabled", C.c_int32), ("_pad0", C.c_int32),
        ("road_type_penalty_r1", C.c_double), ("road_type_penalty_r2", C.c_double),
        ("road_type_penalty_r3", C.c_double), ("dynamic_penalty_scale", C.c_double),
        ("light_algorithm", C.c_int32), ("transition_duration_enabled", C.c_int32),
        ("transition_clearance_enabled", C.c_int32), ("all_red_duration", C.c_int32),
        ("green_duration", C.c_int32), ("qa_min_green", C.c_int32), ("qa_max_green", C.c_int32),
        ("qa_gap", C.c_int32), ("enable_traffic", C.c_int32), ("time_per_step_seconds", C.c_int32),
        ("eager_density", C.c_int32), ("rain_radius_min", C.c_int32), ("rain_radius_max", C.c_int32),
        ("rain_occurrences_max", C.c_int32), ("rain_cooldown", C.c_int32), ("rain_spawn_offset", C.c_int32),
        ("rain_spawn_chance", C.c_double),
        ("stuck_despawn_enabled", C.c_int32), ("stuck_despawn_threshold", C.c_int32),
        ("stuck_despawn_threshold_intersection", C.c_int32), ("respect_awareness", C.c_int32),
        ("pathfinding_batching", C.c_int32),
    ]


class TsWorld(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("allowed_dirs_map", C.c_void_p),
                ("is_road_map", C.c_void_p), ("road_type_map", C.c_void_p), ("intersection_map", C.c_void_p)]


_LT_PTRS = ["g_light_off", "light_xy", "light_ctrl_off", "light_ctrl_xy", "g_ns_off", "g_ns", "g_ew_off", "g_ew",
            "g_icell_off", "g_icell_xy", "g_ns_in_off", "g_ns_in_xy", "g_ns_out_off", "g_ns_out_xy",
            "g_ew_in_off", "g_ew_in_xy", "g_ew_out_off", "g_ew_out_xy", "g_neighbors", "g_neighbors_ctor"]


class TsLightTables(C.Structure):
    _fields_ = [("n_groups", C.c_int32), ("n_lights", C.c_int32)] + [(n, C.c_void_p) for n in _LT_PTRS]


class TsRainInfo(C.Structure):
    _fields_ = [("has_manager", C.c_int32), ("n_rains", C.c_int32), ("cooldown", C.c_int32), ("counter", C.c_int32)]


class TsCounters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "stuck", "collisions", "malfunctions", "overtaking", "in_stuck_detour", "parked", "live_internal",
        "live_through", "count_completed_internal", "count_completed_through", "total_distance_internal",
        "total_distance_through", "errored_internal", "errored_through")] + [
        ("total_duration_internal", C.c_double), ("total_duration_through", C.c_double), ("elapsed", C.c_double)] + [
        (n, C.c_int64) for n in ("step_count", "agent_steps", "astar_calls", "astar_expansions",
                                 "astar_relaxations", "move_rounds", "rng_fixups", "created_internal",
                                 "created_through", "created_service_food", "created_service_waste",
                                 "live_service_food", "live_service_waste")]


class TsTrafficZone(C.Structure):
    _fields_ = [("start_hour", C.c_int32), ("end_hour", C.c_int32), ("through_distribution", C.c_double),
                ("n_internal", C.c_int32), ("origin_type", C.c_int32 * 8), ("dest_type", C.c_int32 * 8),
                ("fraction", C.c_double * 8)]


class TsTrafficTables(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("blk_type", C.c_void_p), ("blk_entr_off", C.c_void_p),
                ("blk_entr_xy", C.c_void_p), ("n_highway_entrances", C.c_int32), ("highway_entrances_xy", C.c_void_p),
                ("n_highway_exits", C.c_int32), ("highway_exits_xy", C.c_void_p),
                ("internal_population_per_day", C.c_int32), ("passing_population_per_day", C.c_int32),
                ("start_offset_seconds", C.c_int32), ("n_zones", C.c_int32), ("zones", TsTrafficZone * 8),
                ("total_service_vehicles_food", C.c_int32), ("total_service_vehicles_waste", C.c_int32),
                ("service_load_time", C.c_int32), ("gradual_city_block_resources", C.c_int32),
                ("food_consumption_ticks", C.c_int32), ("waste_production_ticks", C.c_int32),
                ("needs_food_type_mask", C.c_int32), ("produces_waste_type_mask", C.c_int32),
                ("service_max_load_food", C.c_double), ("service_max_load_waste", C.c_double),
                ("food_capacity_per_cell", C.c_double), ("waste_capacity_per_cell", C.c_double),
                ("blk_inner_cells", C.c_void_p), ("blk_service_off", C.c_void_p), ("blk_service_xy", C.c_void_p),
                ("statistics_update_interval", C.c_int32)]


class TsCachedStats(C.Structure):
    """include/trafficsim.h: DynamicTrafficAgent._update_cached_stats' raw figures (index 0..3 = internal, through,
    service_food, service_waste)."""
    _fields_ = [("valid", C.c_int32), ("pad_", C.c_int32), ("update_step", C.c_int64),
                ("dur_live", C.c_double * 2), ("dist_live", C.c_int64 * 2), ("n_live", C.c_int64 * 2),
                ("stuck_ticks_sum", C.c_int64), ("stuck_ticks_max", C.c_int64),
                ("stuck", C.c_int64), ("collisions", C.c_int64), ("malfunctions", C.c_int64), ("parked", C.c_int64),
                ("overtaking", C.c_int64), ("in_stuck_detour", C.c_int64),
                ("live_internal", C.c_int64), ("live_through", C.c_int64), ("live_service_food", C.c_int64), ("live_service_waste", C.c_int64),
                ("count_completed", C.c_int64 * 2), ("total_distance", C.c_int64 * 2), ("total_duration", C.c_double * 2),
                ("daily_total", C.c_int64 * 4), ("created", C.c_int64 * 4), ("errored", C.c_int64 * 4),
                ("eta", C.c_double * 4), ("avg_daily_difference", C.c_double)]


# Defaults.TIME_ZONES (config.py:155-236) with block types as indices into AVAILABLE_CITY_BLOCKS
_BT = {"Res": 0, "Off": 1, "Mar": 2, "Lei": 3, "Oth": 4}
DEFAULT_TIME_ZONES = [
    (6, 9, 0.15, [("Res", "Off", 0.05), ("Res", "Mar", 0.05), ("Res", "Lei", 0.02), ("Res", "Oth", 0.03)]),
    (9, 12, 0.20, [("Res", "Mar", 0.10), ("Res", "Oth", 0.04), ("Off", "Oth", 0.06)]),
    (12, 15, 0.15, [("Res", "Mar", 0.07), ("Res", "Oth", 0.03), ("Off", "Oth", 0.05)]),
    (15, 18, 0.15, [("Res", "Mar", 0.03), ("Off", "Oth", 0.05), ("Mar", "Oth", 0.05), ("Lei", "Oth", 0.02)]),
    (18, 21, 0.12, [("Res", "Oth", 0.02), ("Res", "Lei", 0.02), ("Off", "Lei", 0.02), ("Mar", "Lei", 0.02),
                    ("Oth", "Lei", 0.02), ("Mar", "Oth", 0.01), ("Lei", "Oth", 0.01)]),
    (21, 24, 0.10, [("Off", "Res", 0.03), ("Mar", "Res", 0.03), ("Lei", "Res", 0.02), ("Oth", "Res", 0.02)]),
    (0, 3, 0.08, [("Off", "Res", 0.02), ("Lei", "Res", 0.04), ("Oth", "Res", 0.01), ("Res", "Lei", 0.01)]),
    (3, 6, 0.05, [("Res", "Mar", 0.02), ("Res", "Lei", 0.02), ("Res", "Oth", 0.01)]),
]


# Defaults attribute name -> TsParams field (config.py)
DEFAULTS_TO_PARAMS = {
    "VEHICLE_MIN_SPEED": "vehicle_min_speed", "VEHICLE_MAX_SPEED": "vehicle_max_speed",
    "VEHICLE_AWARENESS_RANGE": "vehicle_awareness_range", "RAIN_ENABLED": "rain_enabled",
    "RAIN_SPEED_REDUCTION": "rain_speed_reduction", "PATHFINDING_COOLDOWN": "pathfinding_cooldown",
    "PATHFINDING_CACHE": "pathfinding_cache", "VEHICLE_STUCK_RECOMPUTE_THRESHOLD": "stuck_recompute_threshold",
    "VEHICLE_STUCK_RECOMPUTE_THRESHOLD_INTERSECTION": "stuck_recompute_threshold_intersection",
    "VEHICLE_CONTRAFLOW_OVERTAKE_ACTIVE": "contraflow_overtake_active",
    "VEHICLE_MAX_CONTRAFLOW_OVERTAKE_STEPS": "max_contraflow_overtake_steps",
    "VEHICLE_CONTRAFLOW_OVERTAKE_DURATION": "contraflow_overtake_duration",
    "VEHICLE_STUCK_CONTRAFLOW_ENABLED": "stuck_contraflow_enabled",
    "VEHICLE_STUCK_CONTRAFLOW_THRESHOLD": "stuck_contraflow_threshold",
    "VEHICLE_STUCK_CONTRAFLOW_THRESHOLD_INTERSECTION": "stuck_contraflow_threshold_intersection",
    "VEHICLE_MAX_CONTRAFLOW_STUCK_DETOUR_STEPS": "max_contraflow_stuck_detour_steps",
    "VEHICLE_CONTRAFLOW_STUCK_DETOUR_DURATION": "contraflow_stuck_detour_duration",
    "VEHICLE_MALFUNCTION_ACTIVE": "malfunction_active", "VEHICLE_MALFUNCTION_CHANCE": "malfunction_chance",
    "VEHICLE_MALFUNCTION_DURATION": "malfunction_duration",
    "VEHICLE_SIDESWIPE_COLLISION_ACTIVE": "sideswipe_active",
    "VEHICLE_SIDESWIPE_COLLISION_CHANCE": "sideswipe_chance",
    "VEHICLE_SIDESWIPE_COLLISION_DURATION": "sideswipe_duration",
    "VEHICLE_CONTRAFLOW_PENALTY": "contraflow_penalty",
    "VEHICLE_OBSTACLE_PENALTY_VEHICLE": "obstacle_penalty_vehicle",
    "VEHICLE_OBSTACLE_PENALTY_STOP": "obstacle_penalty_stop",
    "VEHICLE_ROAD_TYPES_PENALTIES_ENABLED": "road_type_penalties_enabled",
    "VEHICLE_ROAD_TYPES_PENALTY_R1": "road_type_penalty_r1", "VEHICLE_ROAD_TYPES_PENALTY_R2": "road_type_penalty_r2",
    "VEHICLE_ROAD_TYPES_PENALTY_R3": "road_type_penalty_r3",
    "VEHICLE_TURN_PENALTY_ENABLED": "turn_penalty_enabled", "VEHICLE_TURN_PENALTY": "turn_penalty",
    "VEHICLE_DYNAMIC_PENALTIES_ENABLED": "dynamic_penalties_enabled",
    "VEHICLE_DYNAMIC_PENALTY_SCALE": "dynamic_penalty_scale",
    "TRAFFIC_LIGHT_AGENT_ALGORITHM": "light_algorithm",
    "TRAFFIC_LIGHT_TRANSITION_DURATION_ENABLED": "transition_duration_enabled",
    "TRAFFIC_LIGHT_TRANSITION_CLEARANCE_ENABLED": "transition_clearance_enabled",
    "TRAFFIC_LIGHT_ALL_RED_DURATION": "all_red_duration", "TRAFFIC_LIGHT_GREEN_DURATION": "green_duration",
    "TRAFFIC_LIGHT_QUEUE_ACTUATED_MIN_GREEN": "qa_min_green",
    "TRAFFIC_LIGHT_QUEUE_ACTUATED_MAX_GREEN": "qa_max_green", "TRAFFIC_LIGHT_QUEUE_ACTUATED_GAP": "qa_gap",
    "ENABLE_TRAFFIC": "enable_traffic", "TIME_PER_STEP_IN_SECONDS": "time_per_step_seconds",
    "RAIN_RADIUS_MIN": "rain_radius_min", "RAIN_RADIUS_MAX": "rain_radius_max",
    "RAIN_OCCURRENCES_MAX": "rain_occurrences_max", "RAIN_COOLDOWN": "rain_cooldown",
    "RAIN_SPAWN_OFFSET": "rain_spawn_offset", "RAIN_SPAWN_CHANCE": "rain_spawn_chance",
    "VEHICLE_STUCK_DESPAWN_ENABLED": "stuck_despawn_enabled", "VEHICLE_STUCK_DESPAWN_THRESHOLD": "stuck_despawn_threshold",
    "VEHICLE_STUCK_DESPAWN_THRESHOLD_INTERSECTION": "stuck_despawn_threshold_intersection",
    "VEHICLE_RESPECT_AWARENESS": "respect_awareness",
    "PATHFINDING_BATCHING": "pathfinding_batching",
}
# switches whose non-default value selects a code path this build does not carry: (unsupported value, why).
# params_from_defaults refuses them loudly instead of running the default behaviour.
UNSUPPORTED_DEFAULTS = {
}


# ts_exchange_fn (include/trafficsim.h): all-gather of variable-size host byte buffers
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_int64))


class EngineError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"trafficsim error {code}: {msg}")
        self.code = code


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def path_crc(xy) -> int:
    """crc32 of a path as int32 (x, y) pairs, 0 if empty - the TS_V_PATH_CRC convention."""
    a = _i32(xy).reshape(-1, 2)
    if a.size == 0:
        return 0
    return zlib.crc32(a.tobytes()) & 0xFFFFFFFF


class CApi:
    """One engine instance behind the C-ABI (`prefix` = "ts_" for the HIP engine)."""

    def __init__(self, lib: C.CDLL, prefix: str):
        self.lib, self.prefix = lib, prefix
        self.h = C.c_void_p()
        self._keep = []
        f = self._f
        f("default_params").restype = None
        f("last_error").restype = C.c_char_p
        f("last_error").argtypes = [C.c_void_p]
        for name in ("destroy", "num_vehicles", "num_groups", "num_scheduled", "num_blocks", "num_spawned", "rain_spawn"):
            f(name).argtypes = [C.c_void_p]
        f("create").argtypes = [C.POINTER(TsWorld), C.POINTER(TsParams), C.POINTER(C.c_void_p)]
        f("set_lights").argtypes = [C.c_void_p, C.POINTER(TsLightTables)]
        f("schedule_add").argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        f("set_traffic_generator").argtypes = [C.c_void_p, C.POINTER(TsTrafficTables)]
        f("cached_stats").argtypes = [C.c_void_p, C.POINTER(TsCachedStats)]
        f("seed").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32]
        f("seed_int").argtypes = [C.c_void_p, C.c_int32, C.c_uint64]
        f("rng_state").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_uint32)]
        f("add_vehicles").argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
        f("add_vehicles_dirs").argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
        f("remove_vehicle").argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        f("upload_map").argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        f("step").argtypes = [C.c_void_p, C.c_int32]
        f("download_map").argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        f("download_density").argtypes = [C.c_void_p, C.c_void_p]
        f("download_vehicles").argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        f("download_path").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        f("download_groups").argtypes = [C.c_void_p, C.c_void_p]
        f("download_blocks").argtypes = [C.c_void_p, C.c_void_p]
        f("rain_info").argtypes = [C.c_void_p, C.POINTER(TsRainInfo)]
        f("add_service_vehicle").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        f("group_links").argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        f("download_vehicle_meta").argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        f("download_service_vehicles").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        f("counters").argtypes = [C.c_void_p, C.POINTER(TsCounters)]
        f("astar").argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p, C.c_int32]
        f("debug_set_occupancy").argtypes = [C.c_void_p, C.c_void_p]
        f("set_device").argtypes = [C.c_int32]
        f("set_replan_sharding").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        if True:
            f("set_replan_sharding_device").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        f("profile_enable").argtypes = [C.c_void_p, C.c_int32]
        f("profile_name").restype = C.c_char_p
        f("profile_name").argtypes = [C.c_int32]
        f("profile_get").argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64)]

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def _chk(self, rc: int) -> int:
        if rc < 0:
            msg = self._f("last_error")(self.h) if self.h else b""
            raise EngineError(rc, (msg or b"").decode())
        return rc

    # ---- construction ----------------------------------------------------------------------
    def default_params(self) -> TsParams:
        p = TsParams()
        self._f("default_params")(C.byref(p))
        return p

    def params_from_defaults(self, overrides: Optional[dict] = None) -> TsParams:
        """TsParams from config.py defaults plus `Defaults`-style overrides (UPPER_CASE keys)."""
        p = self.default_params()
        for k, v in (overrides or {}).items():
            if k in UNSUPPORTED_DEFAULTS and bool(v) == UNSUPPORTED_DEFAULTS[k][0]:
                raise EngineError(TS_E_UNSUPPORTED, f"{k}={v!r}: {UNSUPPORTED_DEFAULTS[k][1]}")
            if k in DEFAULTS_TO_PARAMS:
                field = DEFAULTS_TO_PARAMS[k]
                if field == "light_algorithm":
                    if v not in LIGHT_ALGORITHMS:
                        raise EngineError(TS_E_UNSUPPORTED, f"light algorithm {v!r} is out of scope (RL variants)")
                    v = LIGHT_ALGORITHMS[v]
                setattr(p, field, type(getattr(p, field))(v))
        return p

    def create(self, allowed_dirs, is_road, road_type, intersection, params: TsParams):
        a = np.ascontiguousarray(allowed_dirs, dtype=np.uint8)
        r = np.ascontiguousarray(is_road, dtype=np.int8)
        t = np.ascontiguousarray(road_type, dtype=np.int8)
        i = np.ascontiguousarray(intersection, dtype=np.int8)
        H, W = a.shape
        assert r.shape == t.shape == i.shape == (H, W)
        self.W, self.H = W, H
        w = TsWorld(W, H, a.ctypes.data, r.ctypes.data, t.ctypes.data, i.ctypes.data)
        rc = self._f("create")(C.byref(w), C.byref(params), C.byref(self.h))
        if rc < 0:
            raise EngineError(rc, "create failed")
        return self

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lights(self, tables: dict):
        """`tables`: the G1 light tables (keys as in tests/golden world tables / citygen)."""
        t = TsLightTables()
        arrs = {
            "g_light_off": tables["g_light_off"], "light_xy": tables["light_xy"],
            "light_ctrl_off": tables["light_ctrl_off"], "light_ctrl_xy": tables["light_ctrl_xy"],
            "g_ns_off": tables["g_ns_lights_off"], "g_ns": tables["g_ns_lights"],
            "g_ew_off": tables["g_ew_lights_off"], "g_ew": tables["g_ew_lights"],
            "g_icell_off": tables["g_icell_off"], "g_icell_xy": tables["g_icell_xy"],
            "g_ns_in_off": tables["g_ns_in_off"], "g_ns_in_xy": tables["g_ns_in_xy"],
            "g_ns_out_off": tables["g_ns_out_off"], "g_ns_out_xy": tables["g_ns_out_xy"],
            "g_ew_in_off": tables["g_ew_in_off"], "g_ew_in_xy": tables["g_ew_in_xy"],
            "g_ew_out_off": tables["g_ew_out_off"], "g_ew_out_xy": tables["g_ew_out_xy"],
            "g_neighbors": tables["g_neighbors"],
            "g_neighbors_ctor": tables.get("g_neighbors_ctor", tables["g_neighbors"]),
        }
        keep = {k: _i32(v) for k, v in arrs.items()}
        t.n_groups = len(keep["g_light_off"]) - 1
        t.n_lights = len(keep["light_ctrl_off"]) - 1
        for k, v in keep.items():
            setattr(t, k, v.ctypes.data)
        self._chk(self._f("set_lights")(self.h, C.byref(t)))
        self.n_groups = t.n_groups

    def cached_stats(self) -> Dict[str, object]:
        """DynamicTrafficAgent.cached_stats as the statistics panel reads it (ui_modules/traffic_statistics.py): the
        reference's keys and quotients (dynamic_traffic_generator.py:562-648) formed from ts_cached_stats' raw figures;
        {} until the generator's first update, like the reference's dict."""
        c = TsCachedStats()
        self._chk(self._f("cached_stats")(self.h, C.byref(c)))
        if not c.valid:
            return {}

        def safe(a, b):
            return a / b if b else 0.0
        out: Dict[str, object] = {}
        names = ("internal", "through")
        for what, num_c, num_l, den_c, den_l in (
                ("avg_duration", c.total_duration, c.dur_live, c.count_completed, c.n_live),
                ("avg_time_per_unit", c.total_duration, c.dur_live, c.total_distance, c.dist_live)):
            for k, nm in enumerate(names):
                out[f"{what}_{nm}_completed"] = safe(num_c[k], den_c[k])
            for k, nm in enumerate(names):
                out[f"{what}_{nm}_live"] = safe(num_l[k], den_l[k])
            for k, nm in enumerate(names):
                out[f"{what}_{nm}_total"] = safe(num_c[k] + num_l[k], den_c[k] + den_l[k])
        for what in ("avg_duration", "avg_time_per_unit"):
            for nm in names:
                out[f"{what}_{nm}"] = out[f"{what}_{nm}_total"]
        out["avg_daily_difference"] = c.avg_daily_difference
        out["count_completed_internal"] = int(c.count_completed[0])
        for f in ("live_internal", "live_through", "live_service_food", "live_service_waste", "collisions", "malfunctions", "parked",
                  "overtaking", "stuck"):
            out[f] = int(getattr(c, f))
        out["live_average_stuck_duration"] = (c.stuck_ticks_sum / c.stuck) if c.stuck > 0 else 0.0
        out["live_max_stuck_duration"] = int(c.stuck_ticks_max)
        out["in_stuck_detour"] = int(c.in_stuck_detour)
        for k, kind in enumerate(("internal", "through", "service_food", "service_waste")):
            total, created = int(c.daily_total[k]), int(c.created[k])
            out[f"daily_total_{kind}"] = total
            out[f"created_{kind}"] = created
            out[f"remaining_{kind}"] = total - created
            out[f"percentage_created_{kind}"] = (created / total * 100) if total else 0.0
            out[f"errored_{kind}"] = int(c.errored[k]) if k < 2 else 0.0      # (getattr(self, "errored_service_*", 0.0): no such attribute)
            out[f"eta_{kind}"] = None if c.eta[k] != c.eta[k] else float(c.eta[k])
        return out

    def set_traffic_generator(self, tables: dict, internal_per_day=10000, passing_per_day=2400,
                              start_offset_seconds=6 * 3600, zones=None, service: Optional[dict] = None,
                              statistics_update_interval: int = 20):
        """DynamicTrafficAgent.__init__: `tables` carries blk_type / blk_entr_off / blk_entr_xy /
        highway_entrances_xy / highway_exits_xy (golden world-table keys).  Generates day 0 (global stream)."""
        t = TsTrafficTables()
        keep = dict(bt=_i32(tables["blk_type"]), bo=_i32(tables["blk_entr_off"]), bx=_i32(tables["blk_entr_xy"]),
                    hi=_i32(tables["highway_entrances_xy"]), ho=_i32(tables["highway_exits_xy"]))
        t.n_blocks = len(keep["bt"])
        t.blk_type, t.blk_entr_off, t.blk_entr_xy = keep["bt"].ctypes.data, keep["bo"].ctypes.data, keep["bx"].ctypes.data
        t.n_highway_entrances = len(keep["hi"].reshape(-1, 2))
        t.highway_entrances_xy = keep["hi"].ctypes.data
        t.n_highway_exits = len(keep["ho"].reshape(-1, 2))
        t.highway_exits_xy = keep["ho"].ctypes.data
        t.internal_population_per_day, t.passing_population_per_day = int(internal_per_day), int(passing_per_day)
        t.start_offset_seconds = int(start_offset_seconds)
        t.statistics_update_interval = int(statistics_update_interval)
        zs = zones if zones is not None else DEFAULT_TIME_ZONES
        t.n_zones = len(zs)
        for i, (h0, h1, thr, pairs) in enumerate(zs):
            z = t.zones[i]
            z.start_hour, z.end_hour, z.through_distribution, z.n_internal = h0, h1, thr, len(pairs)
            for k, (o, dd, fr) in enumerate(pairs):
                z.origin_type[k], z.dest_type[k], z.fraction[k] = _BT[o], _BT[dd], fr
        if "blk_inner_cells" in tables:
            service = service or {}
            keep["bi"] = _i32(tables["blk_inner_cells"])
            keep["so"], keep["sx"] = _i32(tables["blk_service_off"]), _i32(tables["blk_service_xy"])
            t.blk_inner_cells, t.blk_service_off, t.blk_service_xy = (
                keep["bi"].ctypes.data, keep["so"].ctypes.data, keep["sx"].ctypes.data)
            t.total_service_vehicles_food = int(service.get("service_food", 0))
            t.total_service_vehicles_waste = int(service.get("service_waste", 0))
            t.service_load_time = int(service.get("load_time", 20))
            t.gradual_city_block_resources = int(service.get("gradual", True))
            t.food_consumption_ticks = int(service.get("food_consumption_ticks", 50))
            t.waste_production_ticks = int(service.get("waste_production_ticks", 100))
            t.needs_food_type_mask = int(service.get("needs_food_type_mask", 0b01100))
            t.produces_waste_type_mask = int(service.get("produces_waste_type_mask", 0b11111))
            t.service_max_load_food = float(service.get("max_load_food", 50.0))
            t.service_max_load_waste = float(service.get("max_load_waste", 250.0))
            t.food_capacity_per_cell = float(service.get("food_capacity_per_cell", 2.0))
            t.waste_capacity_per_cell = float(service.get("waste_capacity_per_cell", 1.5))
        self._chk(self._f("set_traffic_generator")(self.h, C.byref(t)))

    def schedule_add(self, kind: int, count: int = 1):
        self._chk(self._f("schedule_add")(self.h, kind, count))

    def seed_state(self, stream: int, state):
        """`state` = random.getstate() (or just its [1] tuple of 625 ints)."""
        tup = state[1] if (isinstance(state, tuple) and len(state) == 3) else state
        arr = np.asarray(tup, dtype=np.uint64)
        assert arr.size == 625
        mt = np.ascontiguousarray(arr[:624].astype(np.uint32))
        self._chk(self._f("seed")(self.h, stream, mt.ctypes.data, int(arr[624])))

    def seed_int(self, stream: int, seed: int):
        self._chk(self._f("seed_int")(self.h, stream, seed))

    def rng_state(self, stream: int):
        mt = np.zeros(624, dtype=np.uint32)
        idx = C.c_uint32()
        self._chk(self._f("rng_state")(self.h, stream, mt.ctypes.data, C.byref(idx)))
        return mt, idx.value

    def rng_fingerprint(self, stream: int):
        mt, idx = self.rng_state(stream)
        return zlib.crc32(mt.tobytes()) & 0xFFFFFFFF, idx

    def add_vehicles(self, start_xy, goal_xy, population_type=None, path_off=None, path_xy=None):
        s, g = _i32(start_xy).reshape(-1, 2), _i32(goal_xy).reshape(-1, 2)
        n = len(s)
        pt = _i32(population_type if population_type is not None else np.zeros(n))
        po = px = None
        if path_off is not None:
            po, px = _i32(path_off), _i32(path_xy).reshape(-1, 2)
            assert len(po) == n + 1
        self._chk(self._f("add_vehicles")(
            self.h, n, s.ctypes.data, g.ctypes.data, pt.ctypes.data,
            po.ctypes.data if po is not None else None, px.ctypes.data if px is not None else None))

    def add_vehicles_dirs(self, start_xy, goal_xy, population_type, path_off, path_dirs):
        """Routes as direction codes (N0 E1 S2 W3), one byte per step; path_off is int64."""
        s, g = _i32(start_xy).reshape(-1, 2), _i32(goal_xy).reshape(-1, 2)
        n = len(s)
        pt = _i32(population_type if population_type is not None else np.zeros(n))
        po = np.ascontiguousarray(path_off, dtype=np.int64)
        pd = np.ascontiguousarray(path_dirs, dtype=np.uint8)
        assert len(po) == n + 1 and len(pd) == po[-1]
        self._chk(self._f("add_vehicles_dirs")(self.h, n, s.ctypes.data, g.ctypes.data, pt.ctypes.data,
                                                po.ctypes.data, pd.ctypes.data))

    def remove_vehicle(self, spawn_idx: int, population_type: int = 0):
        """CityModel.remove_vehicle between ticks; `spawn_idx` = column 0 of vehicles().  `population_type` (POP[...]) is the
        reference's argument of that name: the live counter it names drops by one, 'undefined' (the default) touches none."""
        self._chk(self._f("remove_vehicle")(self.h, int(spawn_idx), int(population_type)))

    def upload_map(self, which: int, arr):
        a = np.ascontiguousarray(arr, dtype=np.int8)
        assert a.shape == (self.H, self.W)
        self._chk(self._f("upload_map")(self.h, which, a.ctypes.data))

    def set_device(self, device: int):
        rc = self._f("set_device")(device)
        if rc < 0:
            raise EngineError(rc, f"cannot select HIP device {device}")

    def profile_enable(self, on: bool = True):
        self._chk(self._f("profile_enable")(self.h, int(on)))

    def profile(self) -> dict:
        """{kernel name: (total_ms, launches, items)} measured with HIP events on the engine's stream."""
        out = {}
        for k in range(self._f("profile_count")()):
            ms, n, it = C.c_double(), C.c_int64(), C.c_int64()
            self._chk(self._f("profile_get")(self.h, k, C.byref(ms), C.byref(n), C.byref(it)))
            out[self._f("profile_name")(k).decode()] = (ms.value, n.value, it.value)
        return out

    def set_replan_sharding(self, rank: int, world: int, callback, device_buffers: bool = False):
        """ts_set_replan_sharding (host buffers) / ts_set_replan_sharding_device (device buffers): `callback` is an
        EXCHANGE_FN instance (kept alive here) or None for world == 1."""
        self._exchange_cb = callback
        ptr = C.cast(callback, C.c_void_p) if callback is not None else None
        name = "set_replan_sharding_device" if device_buffers else "set_replan_sharding"
        self._chk(self._f(name)(self.h, int(rank), int(world), ptr, None))

    def debug_set_occupancy(self, arr):
        """Test hook: overwrite occupancy_map without placing vehicles (A*/density KATs)."""
        a = np.ascontiguousarray(arr, dtype=np.int8)
        assert a.shape == (self.H, self.W)
        self._chk(self._f("debug_set_occupancy")(self.h, a.ctypes.data))

    # ---- stepping and read-back ---------------------------------------------------------------
    def step(self, n: int = 1):
        self._chk(self._f("step")(self.h, n))

    def num_vehicles(self) -> int:
        return self._chk(self._f("num_vehicles")(self.h))

    def num_scheduled(self) -> int:
        return self._chk(self._f("num_scheduled")(self.h))

    def map(self, which: int) -> np.ndarray:
        out = np.zeros((self.H, self.W), dtype=np.int8)
        self._chk(self._f("download_map")(self.h, which, out.ctypes.data))
        return out

    def density(self) -> np.ndarray:
        out = np.zeros((self.H, self.W), dtype=np.float32)
        self._chk(self._f("download_density")(self.h, out.ctypes.data))
        return out

    def vehicles(self) -> np.ndarray:
        n = self.num_vehicles()
        out = np.zeros((max(n, 1), len(V_FIELDS)), dtype=np.int32)
        got = self._chk(self._f("download_vehicles")(self.h, out.ctypes.data, max(n, 1)))
        return out[:got]

    def path(self, active_pos: int) -> np.ndarray:
        n = self._chk(self._f("download_path")(self.h, active_pos, None, 0))
        out = np.zeros((max(n, 1), 2), dtype=np.int32)
        self._chk(self._f("download_path")(self.h, active_pos, out.ctypes.data, max(n, 1)))
        return out[:n]

    def groups(self) -> np.ndarray:
        n = self._chk(self._f("num_groups")(self.h))
        out = np.zeros((max(n, 1), len(G_FIELDS)), dtype=np.int32)
        self._chk(self._f("download_groups")(self.h, out.ctypes.data))
        return out[:n]

    def num_spawned(self) -> int:
        return self._chk(self._f("num_spawned")(self.h))

    def vehicle_meta(self) -> np.ndarray:
        """[n][M_FIELDS] rows in the order of vehicles(): population, target, vehicle type, service phase."""
        n = self.num_vehicles()
        out = np.zeros((max(n, 1), len(M_FIELDS)), dtype=np.int32)
        got = self._chk(self._f("download_vehicle_meta")(self.h, out.ctypes.data, max(n, 1)))
        return out[:got]

    def service_vehicles(self):
        """(spawn_idx [n], loads [n][2] = current / max, block [n]) of the live service vehicles, by spawn index."""
        cap = max(self.num_vehicles(), 1)
        idx = np.zeros(cap, dtype=np.int32)
        loads = np.zeros((cap, 2), dtype=np.float64)
        blk = np.zeros(cap, dtype=np.int32)
        n = self._chk(self._f("download_service_vehicles")(self.h, idx.ctypes.data, loads.ctypes.data, blk.ctypes.data, cap))
        return idx[:n], loads[:n], blk[:n]

    def group_links(self, group: int, repopulate: bool = False) -> bool:
        """get_opposite_traffic_lights()'s side effect / state (see ts_group_links)."""
        return bool(self._chk(self._f("group_links")(self.h, int(group), int(bool(repopulate)))))

    def add_service_vehicle(self, x: int, y: int, service_type: int):
        """ServiceVehicleAgent(vid, model, entrance, sv_type) from the UI (vehicle_control.py:182-206)."""
        self._chk(self._f("add_service_vehicle")(self.h, int(x), int(y), int(service_type)))

    def rain_info(self) -> TsRainInfo:
        r = TsRainInfo()
        self._chk(self._f("rain_info")(self.h, C.byref(r)))
        return r

    def rain_spawn(self):
        """RainManager.add_random_rain() between ticks (the /spawn_rain handler)."""
        self._chk(self._f("rain_spawn")(self.h))

    def num_blocks(self) -> int:
        return self._chk(self._f("num_blocks")(self.h))

    def blocks(self) -> np.ndarray:
        """(food_units, waste_units) per CityBlock, city_blocks order."""
        n = self._chk(self._f("num_blocks")(self.h))
        out = np.zeros((max(n, 1), 2), dtype=np.float64)
        self._chk(self._f("download_blocks")(self.h, out.ctypes.data))
        return out[:n]

    def counters(self) -> TsCounters:
        c = TsCounters()
        self._chk(self._f("counters")(self.h, C.byref(c)))
        return c

    def astar(self, sx, sy, gx, gy, soft_obstacles=False, ignore_flow=False, maximum_steps=0x7FFFFFFF):
        cap = self.W * self.H
        out = np.zeros((cap, 2), dtype=np.int32)
        n = self._chk(self._f("astar")(self.h, sx, sy, gx, gy, int(soft_obstacles), int(ignore_flow),
                                       int(maximum_steps), out.ctypes.data, cap))
        return out[:n].copy()
