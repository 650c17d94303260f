"""Host-side assembly of an engine instance from world tables (the output of world-gen that the
hot path consumes: static maps + light-group tables + schedule layout)."""
from __future__ import annotations

import json
from typing import Optional

import numpy as np

from . import _capi as capi

# schedule_kinds0 codes written by tests/golden/make_golden.py::world_tables
_KIND_TO_AGENT = {0: capi.AGENT_LIGHT_GROUP, 1: capi.AGENT_CITY_BLOCK, 2: capi.AGENT_RAIN_MANAGER, 3: capi.AGENT_CLOCK,
                  4: capi.AGENT_NOOP}


def build_engine(api: capi.CApi, tables, defaults: Optional[dict] = None, params=None,
                 global_state=None, sched_state=None, global_seed: Optional[int] = None,
                 sched_seed: Optional[int] = None) -> capi.CApi:
    """CityModel.__init__ minus world-gen: maps -> light groups -> schedule -> RNG streams."""
    p = params if params is not None else api.params_from_defaults(defaults)
    api.create(tables["allowed_dirs_map"], tables["is_road_map"], tables["road_type_map"],
               tables["intersection_map"], p)
    api.set_lights(tables)
    kinds = np.asarray(tables["schedule_kinds0"]).astype(int)
    # run-length encode consecutive kinds into schedule_add calls (insertion order is preserved)
    i = 0
    while i < len(kinds):
        j = i
        while j < len(kinds) and kinds[j] == kinds[i]:
            j += 1
        api.schedule_add(_KIND_TO_AGENT[int(kinds[i])], j - i)
        i = j
    if global_state is not None:
        api.seed_state(capi.RNG_GLOBAL, global_state)
    elif global_seed is not None:
        api.seed_int(capi.RNG_GLOBAL, global_seed)
    if sched_state is not None:
        api.seed_state(capi.RNG_SCHEDULER, sched_state)
    elif sched_seed is not None:
        api.seed_int(capi.RNG_SCHEDULER, sched_seed)
    return api


def load_trace(path: str) -> dict:
    """A tests/golden/trace_*.npz fixture as a plain dict (JSON fields decoded)."""
    z = np.load(path, allow_pickle=False)
    d = {k: z[k] for k in z.files}
    for k in ("scenario", "defaults_json", "veh_fields", "grp_fields", "cnt_fields"):
        if k in d:
            d[k] = json.loads(str(d[k]))
    return d
