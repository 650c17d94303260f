"""Mesa-shaped facade over the engine: the API surface that the reference's server, portrayal code
and Tornado handlers use (SURVEY.md §8(b) "API surface the Python facade must keep").

    model = CityModel(width=200, height=200, seed=1)        # the reference's world for that seed (worldgen), or
    model = CityModel(width=4096, height=4096, seed=1, world="synthetic")   # fast vectorised look-alike (citygen), or
    model = CityModel.from_tables(tables, seed=1)           # world tables captured elsewhere
    VehicleAgent("veh_1", model, start_cell, target_cell, population_type="through")
    model.step()                                            # CityModel.step (city_model.py:1831-1860)
    model.grid[x, y], model.schedule.agents, model.occupancy_map, agent.get_portrayal() ...

All state lives on the device; the Python objects are *views* that read a per-tick snapshot
(downloaded lazily, once per tick, the first time anything asks).  Host-side writes that the
reference's UI performs directly on numpy maps (cell.py:241-251: set_light_stop/go) are collected in
the host copy of stop_map and uploaded before the next tick.

Differences from the reference that a caller can observe:
  * `CityModel(width, height, seed=..., **ctor_kwargs)` builds the same world as the reference does after
    `random.seed(seed)` (trafficsimulation_amd.worldgen; the constructor keywords are the reference's) and hands
    the global stream on to the engine in the state world-gen left it in; `world="synthetic"` selects citygen.
  * individual `agent.step()` calls are not meaningful: agents are stepped on the device in the
    shuffled order.  `model.schedule.step()` therefore runs the decide + move phases of one tick.
  * with `traffic={...}` the engine's traffic generator spawns internal / through / service vehicles itself;
    they show up as VehicleAgent / ServiceVehicleAgent views (ids "V_<spawn index>" / "SV_<spawn index>": the
    reference's id strings are not kept on the device), `model.city_blocks` holds CityBlock views.
  * RL light controllers are out of scope; rain clouds are not exposed as agents (only
    `model.rain_map`).
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Optional

import numpy as np

from . import _capi as capi
from .world import build_engine

DIR_NAMES = ["N", "E", "S", "W"]
DIRECTION_ICONS = {"N": "↑", "S": "↓", "E": "→", "W": "←"}


def str_to_unique_int(s: str) -> int:
    """utilities/general.py:12-14."""
    return int(hashlib.md5(s.encode("utf-8")).hexdigest(), 16)


class Defaults:
    """The subset of config.py `Defaults` the facade reads; engine parameters are set through
    `CityModel(defaults={...})` with the reference's attribute names."""
    VEHICLE_BASE_COLOR = "black"
    VEHICLE_PARKED_COLOR = "aliceblue"
    VEHICLE_MALFUNCTION_COLOR = "yellow"
    VEHICLE_COLLISION_COLOR = "red"
    VEHICLE_CONTRAFLOW_OVERTAKE_COLOR = "orange"
    SERVICE_VEHICLE_BASE_COLOR = "darkolivegreen"
    AVAILABLE_CITY_BLOCKS = ["Residential", "Office", "Market", "Leisure", "Other"]
    AGENT_PORTRAYAL_LEVEL = 2
    ZONE_COLORS = {"Road": "saddlebrown", "Intersection": "yellow", "Nothing": "white", "TrafficLight": "lime",
                   "TrafficLightStop": "red", "ControlledRoad": "thistle", "ControlledRoadStop": "salmon",
                   # config.py:98-120, for worlds whose tables carry `cell_type_map` (worldgen)
                   "Residential": "cadetblue", "Office": "orange", "Market": "green", "Leisure": "palevioletred",
                   "Other": "darkkhaki", "Empty": "papayawhip", "Sidewalk": "grey", "Wall": "black", "R1": "dodgerblue",
                   "R2": "saddlebrown", "R3": "darkgreen", "IntersectionPending": "darkkhaki", "HighwayEntrance": "blue",
                   "HighwayExit": "royalblue", "BlockEntrance": "magenta"}


from .worldgen import CELL_TYPE_NAMES as _CELL_TYPE_NAMES


class CellAgent:
    """View of one grid cell (cell.py:11-60): type and directions come from the static maps."""

    def __init__(self, model: "CityModel", x: int, y: int):
        self.model = self.city_model = model
        self.position = self.pos = (x, y)
        self.id = f"Cell_{x}_{y}"
        self.unique_id = str_to_unique_int(self.id)
        self.light = None
        self.intersection_group = None
        self.controlled_blocks: List["CellAgent"] = []

    @property
    def cell_type(self) -> str:
        x, y = self.position
        m = self.model
        if m._cell_type_map is not None:          # the reference's own type strings (worldgen tables)
            return _CELL_TYPE_NAMES[int(m._cell_type_map[y, x])]
        if (x, y) in m._light_index:
            return "TrafficLight"
        if m.intersection_map[y, x]:
            return "Intersection"
        if (x, y) in m._controlled_cells:
            return "ControlledRoad"
        return "Road" if m.is_road_map[y, x] else "Nothing"

    @property
    def block_id(self):
        """CellAgent.block_id of block cells and block entrances (None elsewhere, or without a `block_id_map` table)."""
        m = self.model
        b = int(m._block_id_map[self.position[1], self.position[0]]) if m._block_id_map is not None else 0
        return b or None

    @property
    def block_type(self):
        b = self.block_id
        blk = self.model.city_blocks.get(b) if b else None
        return blk.block_type if blk is not None else None

    @property
    def road_type(self):
        ct = self.cell_type
        return ct if ct in ("R1", "R2", "R3") else None

    @property
    def directions(self) -> List[str]:
        x, y = self.position
        bits = int(self.model.allowed_dirs_map[y, x])
        return [DIR_NAMES[k] for k in range(4) if bits & (1 << k)]

    def get_position(self):
        return self.position

    # ---- names and predicates the UI's drop-downs use (cell.py:160-195) -------------------------------------------------
    def get_display_name(self):
        ct = self.cell_type
        x, y = self.position
        m = self.model
        if ct == "Intersection":
            return f"Intersection_{x}_{y}"                      # the custom_id place_cell gave it (city_model.py:240)
        if ct == "BlockEntrance":
            return f"BlockEntrance_{self.block_id}"
        if ct in ("HighwayEntrance", "HighwayExit"):            # _format_highway_label (cell.py:77-156)
            if y == 0:
                cardinal = "South"
            elif y == m.height - 1:
                cardinal = "North"
            elif x == 0:
                cardinal = "West"
            elif x == m.width - 1:
                cardinal = "East"
            else:
                cardinal = "Center"
            horizontal = cardinal in ("South", "North")
            # highway_id is "highway_horizontal" / "highway_vertical" (city_model.py:1413-1416): one id per orientation,
            # so the label's group index is always 1
            same = m.highway_entrances if ct == "HighwayEntrance" else m.highway_exits
            if horizontal:
                coll = sorted((c for c in same if c.position[1] == y), key=lambda c: c.position[0])
            else:
                coll = sorted((c for c in same if c.position[0] == x), key=lambda c: c.position[1])
            kind = "Entrance" if ct == "HighwayEntrance" else "Exit"
            return f"{'Horizontal' if horizontal else 'Vertical'}_1_{cardinal}_{kind}_{coll.index(self) + 1}"
        if ct == "TrafficLight" and self.intersection_group is not None:
            g = self.intersection_group
            return f"TrafficLight_I{g.id}_#{g.traffic_lights.index(self)}"
        return self.position

    def is_block_entrance(self):
        return self.cell_type == "BlockEntrance"

    def is_highway_entrance(self):
        return self.cell_type == "HighwayEntrance"

    def is_highway_exit(self):
        return self.cell_type == "HighwayExit"

    def is_controlled_road(self):
        return self.cell_type == "ControlledRoad"

    def set_light_stop(self):   # cell.py:241-245
        self.model._write_stop([self.position] + [c.position for c in self.controlled_blocks], 1)

    def set_light_go(self):     # cell.py:247-251
        self.model._write_stop([self.position] + [c.position for c in self.controlled_blocks], 0)

    def is_traffic_light(self):
        return self.cell_type == "TrafficLight"

    def get_portrayal(self):
        x, y = self.position
        ct = self.cell_type
        is_stop = self.model.stop_map[y, x] == 1
        color = Defaults.ZONE_COLORS.get(ct, "white")
        if ct == "TrafficLight" and is_stop:
            color = Defaults.ZONE_COLORS["TrafficLightStop"]
        if ct == "ControlledRoad" and is_stop:
            color = Defaults.ZONE_COLORS["ControlledRoadStop"]
        p = {"Shape": "rect", "w": 1.0, "h": 1.0, "Filled": True, "Layer": 0, "Color": color, "Position": self.position}
        if ct == "ControlledRoad":
            p["Control State"] = "Stop" if is_stop else "Go"
        if self.block_id is not None:               # cell.py:316-317
            p["Block ID"] = self.block_id
        arrows = [DIRECTION_ICONS[d] for d in self.directions]
        if arrows:
            p["Directions"] = " ".join(arrows)
        return p


class VehicleAgent:
    """VehicleAgent(custom_id, model, start_cell, target_cell, population_type, vehicle_type)
    (vehicle_base.py:29-89).  Construction places the vehicle and plans its path on the device."""

    def __init__(self, custom_id, model: "CityModel", start_cell: CellAgent, target_cell: CellAgent,
                 population_type: str = "undefined", vehicle_type: str = "undefined", path=None):
        self.id = custom_id
        self.unique_id = str_to_unique_int(custom_id)
        self.model = self.city_model = model
        self.start_cell, self.target = start_cell, target_cell
        self.population_type, self.vehicle_type = population_type, vehicle_type
        self.base_color = Defaults.VEHICLE_BASE_COLOR
        if self.unique_id in model._vehicle_by_uid:
            raise Exception(f"Agent with unique id {self.unique_id!r} already added to scheduler")
        model._flush_host_writes()
        pop = [capi.POP.get(population_type, 0)]
        spawn_idx = model.engine.num_spawned()   # shared with the vehicles the engine's generator creates
        if path is None:
            model.engine.add_vehicles([start_cell.position], [target_cell.position], pop)
        else:
            model.engine.add_vehicles([start_cell.position], [target_cell.position], pop, [0, len(path)], path)
        self._spawn_idx = spawn_idx
        model._n_spawned = spawn_idx + 1
        model._vehicles[self._spawn_idx] = self
        model._vehicle_by_uid[self.unique_id] = self
        model._invalidate()

    @classmethod
    def _view(cls, model: "CityModel", spawn_idx: int, meta) -> "VehicleAgent":
        """A vehicle the engine's traffic generator created (dynamic_traffic_generator.py:398-430)."""
        vt = int(meta[capi.M_FIELDS.index("vehicle_type")])
        klass = ServiceVehicleAgent if vt else cls
        v = object.__new__(klass)
        v.id = f"{'SV' if vt else 'V'}_{spawn_idx}"
        v.unique_id = str_to_unique_int(v.id)
        v.model = v.city_model = model
        v.start_cell = None
        v._static_target = None if vt else model.cell(int(meta[2]), int(meta[3]))
        v.population_type = {1: "internal", 2: "through"}.get(int(meta[1]), "undefined")
        v.vehicle_type = {capi.TRIP_SERVICE_FOOD: "food", capi.TRIP_SERVICE_WASTE: "waste"}.get(vt, "undefined")
        v.base_color = Defaults.VEHICLE_BASE_COLOR
        v._spawn_idx = spawn_idx
        model._vehicles[spawn_idx] = v
        model._vehicle_by_uid[v.unique_id] = v
        return v

    @property
    def target(self):
        t = self.__dict__.get("_static_target")
        if t is not None:
            return t
        m = self.model._meta_row(self._spawn_idx)     # service vehicles change their target
        return None if m is None else self.model.cell(int(m[2]), int(m[3]))

    @target.setter
    def target(self, cell):
        self.__dict__["_static_target"] = cell

    # ---- live state (one row of the per-tick snapshot) --------------------------------------------
    def _row(self):
        return self.model._vehicle_row(self._spawn_idx)

    def _field(self, name):
        r = self._row()
        return None if r is None else int(r[capi.V_FIELDS.index(name)])

    @property
    def pos(self):
        r = self._row()
        return None if r is None else (int(r[1]), int(r[2]))   # None once removed, like MultiGrid.remove_agent

    @property
    def direction(self):
        d = self._field("direction")
        return None if d is None or d < 0 else DIR_NAMES[d]

    current_speed = property(lambda s: s._field("current_speed"))
    base_speed = property(lambda s: s._field("base_speed"))
    max_steps = property(lambda s: s._field("max_steps"))
    stuck_ticks = property(lambda s: s._field("stuck_ticks"))
    steps_traveled = property(lambda s: s._field("steps_traveled"))
    path_retry_cooldown = property(lambda s: s._field("cooldown"))

    def _flag(self, bit):
        f = self._field("flags")
        return bool(f & bit) if f is not None else False

    is_stuck = property(lambda s: s._flag(capi.F_STUCK))
    is_parked = property(lambda s: s._flag(capi.F_PARKED))
    is_in_collision = property(lambda s: s._flag(capi.F_COLLISION))
    is_in_malfunction = property(lambda s: s._flag(capi.F_MALFUNCTION))
    is_overtaking = property(lambda s: s._flag(capi.F_OVERTAKING))
    is_in_stuck_detour = property(lambda s: s._flag(capi.F_DETOUR))
    blocked_by_vehicle = property(lambda s: s._flag(capi.F_BLOCKED))

    @property
    def path(self):
        i = self.model._active_index(self._spawn_idx)
        return [] if i is None else [tuple(c) for c in self.model.engine.path(i).tolist()]

    def is_stranded(self):
        return self.is_in_collision or self.is_in_malfunction

    def get_current_direction(self):
        return self.direction

    def get_current_path(self):
        return self.path

    def get_target(self):
        return self.target.get_position()

    def get_vehicle_type_name(self):
        return "Vehicle"

    def get_description(self):
        return f"{self.population_type} Citizen"

    def step(self):
        raise NotImplementedError("vehicles are stepped on the device, in the shuffled order, by CityModel.step()")

    def get_portrayal(self):   # vehicle_base.py:817-865
        p = {"Shape": "circle", "r": 0.66, "Filled": True, "Layer": 1, "Color": self.base_color}
        base = Defaults.VEHICLE_CONTRAFLOW_OVERTAKE_COLOR if (self.is_overtaking or self.is_in_stuck_detour) else self.base_color
        flash_on = (self.model.step_count % 2) == 0
        if self.is_in_collision:
            p["Color"] = base if flash_on else Defaults.VEHICLE_COLLISION_COLOR
        elif self.is_in_malfunction:
            p["Color"] = base if flash_on else Defaults.VEHICLE_MALFUNCTION_COLOR
        elif self.is_parked:
            p["Color"] = base if flash_on else Defaults.VEHICLE_PARKED_COLOR
        else:
            p["Color"] = base
        p["Type"] = self.get_vehicle_type_name()
        p["Identifier"] = self.unique_id
        p["Description"] = self.get_description()
        p["Position"] = self.pos
        p["Direction"] = DIRECTION_ICONS.get(self.direction, "?")
        p["Speed"] = self.current_speed
        p["Current Destination"] = self.target.id if self.target else "None"
        flags = []
        if self.is_in_stuck_detour: flags.append("Detouring (Stuck)")
        if self.is_overtaking: flags.append("Overtaking")
        if self.is_in_malfunction: flags.append("Malfunctioning")
        if self.is_in_collision: flags.append("InCollision")
        if self.is_parked: flags.append("Parked")
        if (self.stuck_ticks or 0) > 0: flags.append(f"Stuck ({self.stuck_ticks})")
        p["Status"] = ", ".join(flags) if flags else "Ok"
        return p


class ServiceVehicleAgent(VehicleAgent):
    """View of a ServiceVehicleAgent (vehicle_service.py:13-158); created by the engine's traffic generator."""

    def __init__(self, custom_id, model: "CityModel", start_cell: CellAgent, service_type: str, max_load=None):
        """ServiceVehicleAgent(vid, model, entrance, sv_type) as the UI's CreateServiceVehicleHandler calls it
        (vehicle_control.py:182-206); the generator's own service vehicles arrive as views (VehicleAgent._view)."""
        if max_load is not None:
            raise NotImplementedError("max_load comes from Defaults (the handler never passes it)")
        self.id = custom_id
        self.unique_id = str_to_unique_int(custom_id)
        if self.unique_id in model._vehicle_by_uid:
            raise Exception(f"Agent with unique id {self.unique_id!r} already added to scheduler")
        self.model = self.city_model = model
        self.start_cell = start_cell
        self.population_type = "through"
        self.vehicle_type = service_type.lower()
        self.base_color = Defaults.VEHICLE_BASE_COLOR
        model._flush_host_writes()
        self._spawn_idx = model.engine.num_spawned()
        model.engine.add_service_vehicle(start_cell.position[0], start_cell.position[1],
                                         capi.TRIP_SERVICE_FOOD if self.vehicle_type == "food" else capi.TRIP_SERVICE_WASTE)
        model._n_spawned = self._spawn_idx + 1
        model._vehicles[self._spawn_idx] = self
        model._vehicle_by_uid[self.unique_id] = self
        model._invalidate()

    service_type = property(lambda s: "Food" if s.vehicle_type == "food" else "Waste")

    def _svc(self):
        return self.model._service_row(self._spawn_idx)

    @property
    def current_load(self):
        r = self._svc()
        return None if r is None else float(r[0])

    @property
    def max_load(self):
        r = self._svc()
        return None if r is None else float(r[1])

    @property
    def current_block(self):
        r = self._svc()
        blocks = list(self.model.city_blocks.values())
        return None if r is None or int(r[2]) < 0 or int(r[2]) >= len(blocks) else blocks[int(r[2])]

    @property
    def phase(self):
        m = self.model._meta_row(self._spawn_idx)
        return None if m is None else {0: "to_block", 1: "servicing", 2: "to_exit"}.get(int(m[5]))

    def _get_base_color(self):
        return Defaults.SERVICE_VEHICLE_BASE_COLOR

    def get_vehicle_type_name(self):
        return f"{self.service_type}ServiceVehicle"

    def get_description(self):
        return f"{self.service_type}-Service Vehicle ({self.current_load:.1f}/{self.max_load})"

    def get_portrayal(self):
        p = super().get_portrayal()
        b = self.current_block
        p["Destination Block"] = f"{b.id}, {b.block_type}" if b else "None"
        return p


class CityBlock:
    """View of a CityBlock (city_block.py:14-150): stock lives in the engine."""

    def __init__(self, model: "CityModel", index: int, block_type: str, inner_cells: int, entrances, block_id=None):
        self.model = self.city_model = model
        self.index = index
        self.block_id = int(block_id) if block_id is not None else index + 1   # key in model.city_blocks (city_model.py:1743)
        self.id = f"CB_{self.block_id}"                                          # custom_id of _spawn_city_block (1729)
        self.unique_id = str_to_unique_int(self.id)
        self.block_type = block_type
        self._n_inner = int(inner_cells)
        self._entrances = list(entrances)
        sv = model._service_cfg
        self.max_food_units = self._n_inner * sv.get("food_capacity_per_cell", 2.0)
        self.max_waste_units = self._n_inner * sv.get("waste_capacity_per_cell", 1.5)

    def get_entrances(self):
        return self._entrances

    def get_food_units(self):
        return float(self.model._block_rows()[self.index, 0])

    def get_waste_units(self):
        return float(self.model._block_rows()[self.index, 1])

    def needs_food(self):
        return self.block_type in ("Market", "Leisure")        # Defaults.CITY_BLOCK_THAT_NEED_FOOD

    def produces_waste(self):
        return True                                             # Defaults.CITY_BLOCK_THAT_PRODUCE_WASTE = all types

    def food_shortage(self):
        return self.max_food_units - self.get_food_units()

    def waste_surplus(self):
        return self.get_waste_units()

    def step(self):
        raise NotImplementedError("city blocks are stepped by CityModel.step()")

    def __repr__(self):
        return (f"<CityBlock {self.id} | {self.block_type} | Food {self.get_food_units():.1f}/{self.max_food_units}, "
                f"Waste {self.get_waste_units():.1f}/{self.max_waste_units}>")


class IntersectionLightGroup:
    """View of one light group (intersection_light_group.py:30-116)."""

    def __init__(self, model: "CityModel", index: int, lights: List[CellAgent], cells: List[CellAgent]):
        self.model = self.city_model = model
        self.index = index
        self.id = f"Intersection_{index + 1}"
        self.unique_id = str_to_unique_int(self.id)
        self.traffic_lights = lights
        self.intersection_cells = cells

    def _f(self, name):
        return int(self.model._group_rows()[self.index, capi.G_FIELDS.index(name)])

    @property
    def current_phase(self):
        v = self._f("current_phase")
        return None if v < 0 else v

    @property
    def pending_phase(self):
        v = self._f("pending_phase")
        return None if v < 0 else v

    queue_timer = property(lambda s: s._f("queue_timer"))
    gap_timer = property(lambda s: s._f("gap_timer"))
    fixed_time_timer = property(lambda s: s._f("fixed_time_timer"))
    ns_pressure = property(lambda s: s._f("ns_pressure"))
    ew_pressure = property(lambda s: s._f("ew_pressure"))

    # ---- links (intersection_light_group.py:293-307) --------------------------------------------------------------
    def get_opposite_traffic_lights(self):
        """{"N-S": [...], "W-E": [...]}; like the reference, the first call re-runs populate_links() for the group."""
        m = self.model
        m.engine.group_links(self.index, True)
        t = m.tables
        ns = np.asarray(t["g_ns_lights"])[int(t["g_ns_lights_off"][self.index]):int(t["g_ns_lights_off"][self.index + 1])]
        ew = np.asarray(t["g_ew_lights"])[int(t["g_ew_lights_off"][self.index]):int(t["g_ew_lights_off"][self.index + 1])]
        return {"N-S": [m.traffic_lights[int(i)] for i in ns], "W-E": [m.traffic_lights[int(i)] for i in ew]}

    def get_neighbor_groups(self):
        """{direction: group}: the table as the constructor left it until the links are re-populated."""
        m = self.model
        key = "g_neighbors" if m.engine.group_links(self.index, False) else "g_neighbors_ctor"
        nb = np.asarray(m.tables.get(key, m.tables["g_neighbors"])).reshape(-1, 4, 2)[self.index]
        return {DIR_NAMES[int(d)]: m.intersection_light_groups[int(g)] for d, g in nb if d >= 0 and g >= 0}

    def get_intermediate_groups(self):
        """Groups lying between this one and a neighbour without closing all lanes (the reference keeps a set; a list in
        ascending group order here).  Needs the `g_intermediate*` tables (worldgen and make_golden's `worlds` job write
        them; trace fixtures captured earlier do not hold them)."""
        m = self.model
        key = "g_intermediate" if m.engine.group_links(self.index, False) else "g_intermediate_ctor"
        if key not in m.tables:
            raise NotImplementedError("these world tables carry no intermediate-group lists")
        off = np.asarray(m.tables[key + "_off"])
        return [m.intersection_light_groups[int(g)] for g in np.asarray(m.tables[key])[int(off[self.index]):int(off[self.index + 1])]]

    def set_all_go_with_neighbors_and_intermediate(self):     # 332-337
        self.set_all_go_with_neighbors()
        for g in self.get_intermediate_groups():
            g.set_all_go()

    def set_all_stop_with_neighbors_and_intermediate(self):   # 339-344
        self.set_all_stop_with_neighbors()
        for g in self.get_intermediate_groups():
            g.set_all_stop()

    def set_all_go_with_neighbors(self):      # 322-325
        self.set_all_go()
        for g in self.get_neighbor_groups().values():
            g.set_all_go()

    def set_all_stop_with_neighbors(self):    # 327-330
        self.set_all_stop()
        for g in self.get_neighbor_groups().values():
            g.set_all_stop()

    def set_all_stop(self):     # intersection_light_group.py:316-317
        for tl in self.traffic_lights:
            tl.set_light_stop()

    def set_all_go(self):
        for tl in self.traffic_lights:
            tl.set_light_go()

    def step(self):
        raise NotImplementedError("light groups are stepped on the device by CityModel.step()")


class _TrafficStats:
    """dynamic_traffic_generator.py:102-131 counters, read from the engine."""
    _FIELDS = ("stuck", "collisions", "malfunctions", "overtaking", "in_stuck_detour", "parked", "live_internal",
               "live_through", "count_completed_internal", "count_completed_through", "total_distance_internal",
               "total_distance_through", "errored_internal", "errored_through", "total_duration_internal",
               "total_duration_through", "elapsed", "created_internal", "created_through", "created_service_food",
               "created_service_waste", "live_service_food", "live_service_waste")

    def __init__(self, model):
        self._model = model
        self.unique_id = "DTA"

    def __getattr__(self, name):
        if name in _TrafficStats._FIELDS:
            return getattr(self._model._counters(), name)
        raise AttributeError(name)

    @property
    def cached_stats(self) -> Dict[str, object]:
        """DynamicTrafficAgent.cached_stats (dynamic_traffic_generator.py:525-648): refreshed inside the generator's own step
        every STATISTICS_UPDATE_INTERVAL ticks, read by ui_modules/traffic_statistics.py between updates; empty before the
        first update, like the reference's.  A generator without any population to spawn is not stepped as an agent of
        its own by the engine (nothing it does could be seen): its counters are served as they are then."""
        cs = self._model.engine.cached_stats()
        if cs or self._model._traffic_armed:
            return cs
        c = self._model._counters()
        return {f: getattr(c, f) for f in _TrafficStats._FIELDS}


class _RainManager:
    """What the RainControl card and the /spawn_rain handler use of RainManager (rain_control.py:22-73)."""

    def __init__(self, model):
        self._model = model
        self.unique_id = "RainManager"

    cooldown = property(lambda s: int(s._model.engine.rain_info().cooldown))
    counter = property(lambda s: int(s._model.engine.rain_info().counter))

    def add_random_rain(self):
        self._model._flush_host_writes()
        self._model.engine.rain_spawn()
        self._model._invalidate()

    def step(self):
        raise NotImplementedError("the rain manager is stepped by CityModel.step()")


class _Grid:
    """MultiGrid view: grid[x, y] -> [CellAgent, vehicles in arrival order...]."""

    def __init__(self, model):
        self.model, self.width, self.height = model, model.width, model.height

    def __getitem__(self, xy):
        x, y = xy
        return [self.model.cell(x, y)] + self.model._vehicles_at(x, y)

    def coord_iter(self):
        for x in range(self.width):
            for y in range(self.height):
                yield self[x, y], (x, y)

    def out_of_bounds(self, pos):
        return not self.model.in_bounds(*pos)


class _Schedule:
    """RandomActivation view: .agents in insertion order, .step() = one device tick."""

    def __init__(self, model):
        self.model = model
        self.steps = 0

    @property
    def agents(self):
        m = self.model
        return list(m.intersection_light_groups) + list(m.city_blocks.values()) \
            + ([m.dynamic_traffic_generator] if m.dynamic_traffic_generator else []) + m.active_vehicle_agents

    def get_agent_count(self):
        return self.model.engine.num_scheduled()

    def step(self):
        m = self.model
        m._flush_host_writes()
        m.engine.step(1)
        m._invalidate()
        self.steps += 1


class CityModel:
    """CityModel (city_model.py:26-205) on the device engine."""

    def __init__(self, width=200, height=200, seed=None, defaults: Optional[dict] = None, tables: Optional[dict] = None,
                 engine: Optional[capi.CApi] = None, global_seed: Optional[int] = None, global_state=None,
                 sched_state=None, traffic: Optional[dict] = None, world: str = "reference", **world_kwargs):
        import random as _random
        self._seed = seed if seed is not None else _random.random()
        if traffic is not None and "gradual_city_block_resources" in world_kwargs:   # ctor kwarg -> CityBlock mode (city_model.py:50, 1735)
            traffic = dict(traffic, gradual=bool(world_kwargs["gradual_city_block_resources"]))
        if tables is None and world == "synthetic":
            from . import citygen
            tables = citygen.generate(width, height, seed=seed if seed is not None else 1, **world_kwargs)
        elif tables is None:
            # CityModel.__init__'s build sequence (city_model.py:124-148) draws from the global `random` stream; the
            # engine's global stream continues from where that leaves it, as the reference's agents would
            from . import worldgen
            d = defaults or {}
            tables = worldgen.generate_world(width, height, seed=global_seed if global_seed is not None else self._seed,
                                             rain_enabled=bool(d.get("RAIN_ENABLED", True)), enable_traffic=traffic is not None,
                                             block_entrance_road_level=int(d.get("BLOCK_ENTRANCE_ROAD_LEVEL", 0)), **world_kwargs)
            if global_state is None:
                global_state = tables["global_rng_state"]
        self.width, self.height = int(tables["width"]), int(tables["height"])
        if engine is None:
            from ._lib import new_engine
            engine = new_engine()                   # the HIP engine; raises loudly when it is unavailable
        self.engine = engine
        self.tables = tables
        sched_seed = self._seed if isinstance(self._seed, int) else int(self._seed * 2 ** 53)
        gseed = global_seed if global_seed is not None else sched_seed
        # random.setstate()-style states take precedence over integer seeds (exact replays of a reference run)
        build_engine(engine, tables, defaults=defaults or {}, global_seed=gseed, sched_seed=sched_seed,
                     global_state=global_state, sched_state=sched_state)
        # DynamicTrafficAgent("DTA", self) (city_model.py:203-204): traffic = {"P_int", "P_thr", "start_offset",
        # "service_food", "service_waste", ...} arms the engine's generator; it draws day 0 from the global stream now
        self._service_cfg = dict(traffic or {})
        self._traffic_armed = traffic is not None
        if traffic is not None:
            engine.set_traffic_generator(tables, internal_per_day=traffic.get("P_int", 10000),
                                         passing_per_day=traffic.get("P_thr", 2400),
                                         start_offset_seconds=traffic.get("start_offset", 6 * 3600), service=traffic)
        self.allowed_dirs_map = np.asarray(tables["allowed_dirs_map"], dtype=np.uint8)
        self.is_road_map = np.asarray(tables["is_road_map"], dtype=np.int8)
        self.road_type_map = np.asarray(tables["road_type_map"], dtype=np.int8)
        self.intersection_map = np.asarray(tables["intersection_map"], dtype=np.int8)
        self._cell_type_map = np.asarray(tables["cell_type_map"]) if "cell_type_map" in tables else None
        self._block_id_map = np.asarray(tables["block_id_map"]) if "block_id_map" in tables else None
        self._stop_host = np.zeros((self.height, self.width), dtype=np.int8)
        self._stop_dirty = False
        self.step_count = 0
        self._n_spawned = 0
        self._vehicles: Dict[int, VehicleAgent] = {}
        self._vehicle_by_uid: Dict[int, VehicleAgent] = {}
        self._cells: Dict[tuple, CellAgent] = {}
        self._snap = None
        self.user_selected_traffic_light = self.user_selected_intersection = self.user_selected_opposite = None
        # lights / groups
        lxy = np.asarray(tables["light_xy"]).reshape(-1, 2)
        self._light_index = {(int(x), int(y)): i for i, (x, y) in enumerate(lxy)}
        self.traffic_lights = [self.cell(int(x), int(y)) for x, y in lxy]
        coff, cxy = np.asarray(tables["light_ctrl_off"]), np.asarray(tables["light_ctrl_xy"]).reshape(-1, 2)
        self._controlled_cells = set()
        self.controlled_roads = []
        for i, tl in enumerate(self.traffic_lights):
            for x, y in cxy[coff[i]:coff[i + 1]]:
                c = self.cell(int(x), int(y))
                c.light = tl
                tl.controlled_blocks.append(c)
                if (int(x), int(y)) not in self._controlled_cells:
                    self._controlled_cells.add((int(x), int(y)))
                    self.controlled_roads.append(c)
        goff = np.asarray(tables["g_light_off"])
        ioff, ixy = np.asarray(tables["g_icell_off"]), np.asarray(tables["g_icell_xy"]).reshape(-1, 2)
        self.intersection_light_groups = []
        for g in range(len(goff) - 1):
            lights = self.traffic_lights[goff[g]:goff[g + 1]]
            cells = [self.cell(int(x), int(y)) for x, y in ixy[ioff[g]:ioff[g + 1]]]
            grp = IntersectionLightGroup(self, g, lights, cells)
            for c in lights + cells:
                c.intersection_group = grp
            self.intersection_light_groups.append(grp)
        self.dynamic_traffic_generator = _TrafficStats(self) if 3 in np.asarray(tables["schedule_kinds0"]) else None
        self.grid = _Grid(self)
        self.schedule = _Schedule(self)
        self.city_blocks = {}
        if 2 in np.asarray(tables["schedule_kinds0"]):
            self.rain_manager = _RainManager(self)
        if "blk_type" in tables and 1 in np.asarray(tables["schedule_kinds0"]):
            eoff, exy = np.asarray(tables["blk_entr_off"]), np.asarray(tables["blk_entr_xy"]).reshape(-1, 2)
            inner = np.asarray(tables.get("blk_inner_cells", np.zeros(len(eoff) - 1)))
            ids = np.asarray(tables["blk_id"]) if "blk_id" in tables else np.arange(1, len(eoff))
            for b, t in enumerate(np.asarray(tables["blk_type"])):
                cb = CityBlock(self, b, Defaults.AVAILABLE_CITY_BLOCKS[int(t)], int(inner[b]),
                               [self.cell(int(x), int(y)) for x, y in exy[eoff[b]:eoff[b + 1]]], block_id=int(ids[b]))
                self.city_blocks[cb.block_id] = cb       # keyed by block_id like the reference's dict
        self.block_entrances = [self.cell(int(x), int(y)) for x, y in np.asarray(tables.get("block_entrances_xy", np.zeros((0, 2)))).reshape(-1, 2)]
        self.highway_entrances = [self.cell(int(x), int(y)) for x, y in np.asarray(tables.get("highway_entrances_xy", np.zeros((0, 2)))).reshape(-1, 2)]
        self.highway_exits = [self.cell(int(x), int(y)) for x, y in np.asarray(tables.get("highway_exits_xy", np.zeros((0, 2)))).reshape(-1, 2)]

    @classmethod
    def from_tables(cls, tables: dict, seed=None, **kw):
        return cls(tables=tables, seed=seed, **kw)

    # ---- stepping ------------------------------------------------------------------------------
    def step(self):
        """city_model.py:1831-1860: density -> decide -> schedule.step() -> step_count += 1."""
        self.schedule.step()
        self.step_count += 1

    def run_parallel_decide(self):   # fused into the engine tick (city_model.py:1811-1829)
        pass

    def _update_density_map(self):   # materialised on demand by the engine (city_model.py:1764-1778)
        pass

    @property
    def density_map(self):
        self._flush_host_writes()
        return self.engine.density()

    # ---- snapshot plumbing -----------------------------------------------------------------------
    def _invalidate(self):
        self._snap = None

    def _snapshot(self):
        if self._snap is None:
            rows = self.engine.vehicles()
            by_spawn = {int(r[0]): i for i, r in enumerate(rows)}
            cells: Dict[tuple, list] = {}
            for r in rows:
                cells.setdefault((int(r[1]), int(r[2])), []).append(int(r[0]))
            self._snap = dict(rows=rows, by_spawn=by_spawn, cells=cells, maps={}, groups=None, counters=None,
                              meta=None, service=None, blocks=None)
            self._n_spawned = self.engine.num_spawned()
        return self._snap

    def _meta_row(self, spawn_idx):
        s = self._snapshot()
        if s["meta"] is None:
            s["meta"] = self.engine.vehicle_meta()
        i = s["by_spawn"].get(spawn_idx)
        return None if i is None else s["meta"][i]

    def _service_row(self, spawn_idx):
        s = self._snapshot()
        if s["service"] is None:
            idx, loads, blk = self.engine.service_vehicles()
            s["service"] = {int(i): (loads[k, 0], loads[k, 1], int(blk[k])) for k, i in enumerate(idx)}
        return s["service"].get(spawn_idx)

    def _block_rows(self):
        s = self._snapshot()
        if s["blocks"] is None:
            s["blocks"] = self.engine.blocks()
        return s["blocks"]

    rain_map = property(lambda s: s._map(capi.MAP_RAIN))

    @property
    def rains(self):
        """city_model.rains as far as its callers go: they only take its length (rain_control.py:33, 67;
        traffic_statistics.py:216)."""
        return [None] * int(self.engine.rain_info().n_rains)

    def _vehicle_row(self, spawn_idx):
        s = self._snapshot()
        i = s["by_spawn"].get(spawn_idx)
        return None if i is None else s["rows"][i]

    def _active_index(self, spawn_idx):
        return self._snapshot()["by_spawn"].get(spawn_idx)

    def _vehicle_view(self, spawn_idx):
        v = self._vehicles.get(spawn_idx)
        if v is None:   # spawned by the engine's traffic generator
            v = VehicleAgent._view(self, spawn_idx, self._meta_row(spawn_idx))
        return v

    def _vehicles_at(self, x, y):
        return [self._vehicle_view(i) for i in self._snapshot()["cells"].get((x, y), [])]

    def _group_rows(self):
        s = self._snapshot()
        if s["groups"] is None:
            s["groups"] = self.engine.groups()
        return s["groups"]

    def _counters(self):
        s = self._snapshot()
        if s["counters"] is None:
            s["counters"] = self.engine.counters()
        return s["counters"]

    def _map(self, which):
        s = self._snapshot()
        if which not in s["maps"]:
            s["maps"][which] = self.engine.map(which)
        return s["maps"][which]

    occupancy_map = property(lambda s: s._map(capi.MAP_OCCUPANCY))
    stuck_map = property(lambda s: s._map(capi.MAP_STUCK))

    @property
    def stop_map(self):
        return self._stop_host if self._stop_dirty else self._map(capi.MAP_STOP)

    def _write_stop(self, cells, value):
        if not self._stop_dirty:
            self._stop_host = self._map(capi.MAP_STOP).copy()
            self._stop_dirty = True
        for (x, y) in cells:
            self._stop_host[y, x] = value

    def _flush_host_writes(self):
        if self._stop_dirty:
            self.engine.upload_map(capi.MAP_STOP, self._stop_host)
            self._stop_dirty = False
            self._invalidate()

    # ---- getters used by the UI (city_model.py:1965-2149) ------------------------------------------
    @property
    def active_vehicle_agents(self):
        rows = self._snapshot()["rows"]
        return [self._vehicle_view(int(r[0])) for r in rows]

    def cell(self, x, y) -> CellAgent:
        c = self._cells.get((x, y))
        if c is None:
            c = self._cells[(x, y)] = CellAgent(self, x, y)
        return c

    def remove_vehicle(self, vehicle, population_type: str = "undefined", vehicle_type: str = "undefined"):
        """city_model.py:1920-1941 - between ticks: the vehicle leaves the grid, the schedule and the decide order
        As in the reference the live counters follow `population_type` as the caller passes it ('internal' / 'through'; the
        default 'undefined' leaves them alone); `vehicle_type` only concerns service vehicles, which the host cannot remove."""
        self.engine.remove_vehicle(vehicle._spawn_idx, capi.POP.get(population_type, capi.POP["undefined"]))
        self._vehicles.pop(vehicle._spawn_idx, None)
        self._invalidate()

    def get_cell_contents(self, x, y):
        return self.grid[x, y] if self.in_bounds(x, y) else []

    def in_bounds(self, x, y):
        return 0 <= x < self.width and 0 <= y < self.height

    def get_width(self):
        return self.width

    def get_height(self):
        return self.height

    def get_traffic_lights(self):
        return self.traffic_lights

    def get_controlled_roads(self):
        return self.controlled_roads

    def get_intersection_light_groups(self):
        return self.intersection_light_groups

    def get_block_entrances(self):
        return self.block_entrances

    def get_highway_entrances(self):
        return self.highway_entrances

    def get_highway_exits(self):
        return self.highway_exits

    def get_start_blocks(self):
        return list(self.block_entrances) + list(self.highway_entrances)

    def get_exit_blocks(self):
        return list(self.block_entrances) + list(self.highway_exits)

    def get_valid_exits(self, entry_cell):
        """city_model.py:2118-2149.  For a HighwayEntrance the reference filters with `not _are_adjacent(...)`, and
        `_are_adjacent` returns a Manhattan distance (2092-2099): nothing passes, the list is empty."""
        if entry_cell in self.block_entrances:
            return [be for be in self.block_entrances if be is not entry_cell] + list(self.highway_exits)
        return []

    # ---- city-block queries (city_model.py:2017-2087): views over the engine's block stock -------------------------
    @staticmethod
    def _sort_blocks(blocks, by):
        if by == "food":
            return sorted(blocks, key=lambda b: b.get_food_units())
        if by == "waste":
            return sorted(blocks, key=lambda b: -b.get_waste_units())
        return blocks

    def get_all_city_blocks(self, sort_by="unsorted"):
        return self._sort_blocks(list(self.city_blocks.values()), sort_by)

    def get_blocks_needing_food(self, sort_by="unsorted"):
        return self._sort_blocks([b for b in self.city_blocks.values() if b.needs_food()], sort_by)

    def get_blocks_producing_waste(self, sort_by="unsorted"):
        return self._sort_blocks([b for b in self.city_blocks.values() if b.produces_waste()], sort_by)

    def get_city_blocks_by_types(self, block_types, sort_by="unsorted"):
        if block_types is None:
            subset = list(self.city_blocks.values())
        else:
            wanted = {block_types} if isinstance(block_types, str) else set(block_types)
            subset = [b for b in self.city_blocks.values() if b.block_type in wanted]
        return self._sort_blocks(subset, sort_by)

    def get_city_blocks_by_type(self, block_type, sort_by="unsorted"):
        return self.get_city_blocks_by_types([block_type], sort_by)

    def get_residential_city_blocks(self, sort_by="unsorted"):
        return self.get_city_blocks_by_types("Residential", sort_by)

    def get_office_city_blocks(self, sort_by="unsorted"):
        return self.get_city_blocks_by_types("Office", sort_by)

    def get_market_city_blocks(self, sort_by="unsorted"):
        return self.get_city_blocks_by_types("Market", sort_by)

    def get_leisure_city_blocks(self, sort_by="unsorted"):
        return self.get_city_blocks_by_types("Leisure", sort_by)

    def get_other_city_blocks(self, sort_by="unsorted"):
        return self.get_city_blocks_by_types("Other", sort_by)

    def get_block_most_in_need_of_food(self):
        needy = self.get_blocks_needing_food(sort_by="food")
        return needy[0] if needy else None

    def get_block_most_in_need_of_waste_pickup(self):
        dirty = self.get_blocks_producing_waste(sort_by="waste")
        return dirty[0] if dirty else None

    def is_type(self, x, y, ctype):       # city_model.py:1803-1805
        return self.in_bounds(x, y) and self.cell(x, y).cell_type == ctype

    @staticmethod
    def next_cell_in_direction(x, y, d):  # city_model.py:1015-1024
        dx, dy = {"N": (0, 1), "S": (0, -1), "E": (1, 0), "W": (-1, 0)}.get(d, (0, 0))
        return x + dx, y + dy

    def set_traffic_lights_go(self):      # city_model.py:1995-1997
        for tl in self.traffic_lights:
            tl.set_light_go()

    def set_traffic_lights_stop(self):    # city_model.py:1999-2001
        for tl in self.traffic_lights:
            tl.set_light_stop()

    def close(self):
        self.engine.close()


def agent_portrayal(agent):
    """visualization/agent_portrayal.py:18-52: dispatch on the agent's own get_portrayal()."""
    if agent is None or not hasattr(agent, "get_portrayal"):
        return None
    return agent.get_portrayal()
