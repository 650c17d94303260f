"""The reference's pathfinder operator seam on the device engine.

`Simulation/agents/vehicles/vehicle_base.py:11-16` binds a module-level `astar` chosen by
`Defaults.PATHFINDING_METHOD`; every back-end has the signature of `astar_numba` (astar_numba.py:243-256) and returns
the path as (x, y) tuples without the start cell.  `astar_hip` is that callable for this build (SURVEY.md §8(b),
seam 2): the search itself is `ts_astar` (the wave-cooperative kernel of csrc/astar.h, quirks of `astar_core`
included), the maps the caller passes are handed to an engine instance that is created once per set of static maps
and reused.

    from trafficsimulation_amd.pathfinding import astar_hip as astar     # in vehicle_base.py, next to the other choices

Differences a caller can observe:
  * `density_map` is not read: the engine derives the density from `occupancy_map` with the reference's own recipe
    (`_update_density_map`, bit-exact).  That equals the array the reference passes whenever its density is current,
    i.e. everywhere inside the decide phase; a search started mid-move with a stale `density_map` (soft mode only)
    would see the fresh one here.
  * `respect_awareness=True` (field-of-view masking, off by default: config.py:278) is carried: the engine for that flag
    is created with `TsParams.respect_awareness`, `awareness_range` being its `vehicle_awareness_range`.
  * the library must be present: there is no CPU fall-back.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

from . import _capi as capi

_cache: dict = {}     # (W, H, id(is_road), id(road_type), id(allowed)) -> (engine, the three arrays, awareness_range)


def _engine_for(width, height, is_road_map, road_type_map, allowed_dirs_map, awareness_range, factory: Optional[Callable],
                respect_awareness: bool = False):
    key = (int(width), int(height), id(is_road_map), id(road_type_map), id(allowed_dirs_map), int(awareness_range), bool(respect_awareness))
    hit = _cache.get(key)
    # the arrays are kept alive by the cache entry, so an equal id() means the very same objects
    if hit is not None and hit[1] is is_road_map and hit[2] is road_type_map and hit[3] is allowed_dirs_map:
        return hit[0]
    if factory is None:
        from ._lib import new_engine
        factory = new_engine          # raises when the HIP library is missing
    api = factory()
    p = api.default_params()
    p.vehicle_awareness_range = int(awareness_range)      # window of the density map (config.py:279) and of the field of view
    p.respect_awareness = 1 if respect_awareness else 0
    a = np.asarray(allowed_dirs_map)
    if a.shape != (height, width):
        raise ValueError(f"maps must be (height, width) = ({height}, {width}) arrays, got {a.shape}")
    api.create(a, is_road_map, road_type_map, np.zeros((height, width), np.int8), p)   # A* reads no intersection map
    _cache[key] = (api, is_road_map, road_type_map, allowed_dirs_map)
    return api


def astar_hip(width: int, height: int, start_x: int, start_y: int, goal_x: int, goal_y: int,
              occupancy_map: np.ndarray, stop_map: np.ndarray, is_road_map: np.ndarray, road_type_map: np.ndarray,
              allowed_dirs_map: np.ndarray, respect_awareness: bool = False, awareness_range: int = 10,
              density_map: Optional[np.ndarray] = None, soft_obstacles: bool = False, ignore_flow: bool = False,
              maximum_steps: int = 0x7FFFFFFF, _engine_factory: Optional[Callable] = None) -> List[Tuple[int, int]]:
    """astar_numba(width, height, sx, sy, gx, gy, occupancy_map, stop_map, is_road_map, road_type_map,
    allowed_dirs_map, respect_awareness, awareness_range, density_map, soft_obstacles, ignore_flow, maximum_steps).

    One limit the reference's operator does not have: a `maximum_steps` that can actually bind must not exceed 4094 (the
    device table keeps the step count of a search node in 12 bits).  A limit of width * height or more never binds (no chain
    of relaxations revisits a cell) and is accepted, as is the default 0x7FFFFFFF; a value in between raises EngineError
    (TS_E_UNSUPPORTED) - it is never silently clipped.  The reference's own callers pass 6, 20 or nothing."""
    api = _engine_for(width, height, is_road_map, road_type_map, allowed_dirs_map, awareness_range, _engine_factory, respect_awareness)
    api.debug_set_occupancy(occupancy_map)          # the dynamic planes as the caller sees them right now
    api.upload_map(capi.MAP_STOP, stop_map)
    xy = api.astar(int(start_x), int(start_y), int(goal_x), int(goal_y), bool(soft_obstacles), bool(ignore_flow),
                   int(min(maximum_steps, 0x7FFFFFFF)))
    return [(int(x), int(y)) for x, y in xy]


def release():
    """Destroy the cached engine instances (device memory of every set of static maps seen so far)."""
    for entry in _cache.values():
        entry[0].close()
    _cache.clear()
