"""Seed-compatible world generator: the tables the hot path consumes, from (size, seed, options).

Restates `CityModel.__init__`'s build sequence (reference Simulation/city_model.py:124-148) over flat integer
planes instead of one Mesa `CellAgent` per cell.  Two things tie the output to the reference bit for bit:

* every draw goes to a `random.Random` stream in the order, and through the same `random` methods
  (`gauss`, `random`, `choice`, `randint`, `choices`), as the reference draws from the global module stream after
  `random.seed(seed)`;
* wherever the reference iterates a Python `set` of `(x, y)` tuples, the iteration order is part of the result
  (order of highway entrances, which sidewalk run a block entrance sits on, order of intersection clusters ...).
  Those sets are rebuilt here with the same sequence of `add` / `discard` / `pop` calls, so CPython yields the same
  order.  Sets whose order cannot reach the output are replaced by plane lookups.

Output: `generate_world(...)` returns the dict of tables `world.build_engine` takes (same keys as the
tests/golden/trace_*.npz fixtures), plus `global_rng_state` = the 625-word MT19937 state after generation.
"""
from __future__ import annotations

import random as _random
from typing import Dict, List, Optional, Tuple

import numpy as np

# ---- cell types (reference strings in comments are CellAgent.cell_type values) ----
WALL, SIDEWALK, NOTHING, R1, R2, R3, INTERSECTION, BLOCK_ENTRANCE, HW_ENTRANCE, HW_EXIT, CONTROLLED, LIGHT = range(12)
BLOCK0 = 12                       # Residential, Office, Market, Leisure, Other = BLOCK0 + index (config.py:51)
EMPTY_BLOCK = BLOCK0 + 5          # "Empty"
N_BLOCK_TYPES = 5
BLOCK_WEIGHTS = [0.25, 0.25, 0.2, 0.2, 0.1]   # config.py:53-60 CITY_BLOCK_CHANCE in AVAILABLE_CITY_BLOCKS order

CELL_TYPE_NAMES = ("Wall", "Sidewalk", "Nothing", "R1", "R2", "R3", "Intersection", "BlockEntrance", "HighwayEntrance",
                   "HighwayExit", "ControlledRoad", "TrafficLight", "Residential", "Office", "Market", "Leisure", "Other",
                   "Empty")      # CellAgent.cell_type strings, indexed by the codes above (the `cell_type_map` table)

ROADS = (R1, R2, R3)                                                             # config.py:13
ROAD_LIKE = frozenset((R1, R2, R3, INTERSECTION, HW_ENTRANCE, HW_EXIT, BLOCK_ENTRANCE))          # config.py:68
ROAD_LIKE_NO_INTER = frozenset((R1, R2, R3, HW_ENTRANCE, HW_EXIT, BLOCK_ENTRANCE))               # config.py:69
REMOVABLE_DEAD_END = frozenset((R2, R3, INTERSECTION))                                           # config.py:70
TOUCHES_ROAD = frozenset((R1, R2, R3, INTERSECTION, HW_ENTRANCE, CONTROLLED))   # city_model.py:1792-1794
THICKNESS = {R1: 4, R2: 2, R3: 1}                                                # config.py:45-49
ROAD_BY_NAME = {"R1": R1, "R2": R2, "R3": R3, None: None}

# ---- directions: the value is the bit index of allowed_dirs_map (city_model.py:2190-2196) ----
N, E, S, W = 0, 1, 2, 3
ALL_DIRS = (N, S, E, W)            # config.py:62 AVAILABLE_DIRECTIONS order
VEC = {N: (0, 1), S: (0, -1), W: (-1, 0), E: (1, 0)}
OPP = {N: S, S: N, E: W, W: E}
RIGHT_OF = {N: E, E: S, S: W, W: N}
NB4 = ((1, 0), (-1, 0), (0, 1), (0, -1))

Band = Tuple[int, int, int, Optional[int]]   # (start, end, road type, direction)


class _Light:
    __slots__ = ("pos", "controlled", "incoming", "outgoing")

    def __init__(self, pos):
        self.pos = pos
        self.controlled: List[Tuple[int, int]] = []   # ControlledRoad cells, in assignment order
        self.incoming: List[Tuple[int, int]] = []     # assigned_incoming_road_blocks (duplicates kept)
        self.outgoing: List[Tuple[int, int]] = []


class _Group:
    __slots__ = ("lights", "cells", "neighbors", "ctor_neighbors", "intermediate", "ctor_intermediate", "blocks_cache", "ns_in", "ns_out", "ew_in", "ew_out", "ns_lights", "ew_lights")

    def __init__(self, lights, cells):
        self.lights: List[_Light] = lights
        self.cells: List[Tuple[int, int]] = cells
        self.neighbors: Dict[int, "_Group"] = {}
        self.ctor_neighbors: Dict[int, "_Group"] = {}
        self.intermediate: List["_Group"] = []        # intermediate_groups: groups passed over on the way to a neighbour
        self.ctor_intermediate: List["_Group"] = []
        self.blocks_cache: Dict[int, bool] = {}       # the reference's sticky `_blocks_<d>` attributes
        self.ns_in, self.ns_out, self.ew_in, self.ew_out = [], [], [], []
        self.ns_lights: List[_Light] = []
        self.ew_lights: List[_Light] = []


class WorldBuilder:
    """One run of the build sequence.  Attribute names follow the reference's constructor arguments."""

    def __init__(self, width=200, height=200, seed=None, rng: Optional[_random.Random] = None,
                 wall_thickness=15, sidewalk_ring_width=2, ring_road_type="R2",
                 r1_chance_mean=0.15, r1_chance_std=0.03, r2_chance_mean=0.70, r2_chance_std=0.05, min_r1_bands=2,
                 min_block_spacing=6, max_block_spacing=18, optimized_intersections=True,
                 carve_subblock_roads=False, subblock_roads_have_intersections=True, subblock_chance=0.3,
                 subblock_road_type="R3", min_subblock_spacing=5, highway_offset_from_edges=7,
                 traffic_light_range=10, forward_traffic_light_range=False,
                 forward_traffic_light_range_intersections="Skip", block_entrance_road_level=0,
                 use_dummy_agents=False, rain_enabled=True, enable_traffic=True,
                 gradual_city_block_resources=True, cache_cell_portrayal=True):   # accepted like CityModel's; no effect on the tables
        self.W, self.H = int(width), int(height)
        self.rng = rng if rng is not None else _random.Random(seed)
        self.wall = wall_thickness
        self.ring_w = sidewalk_ring_width
        self.ring_type = ROAD_BY_NAME[ring_road_type]
        self.r1_mean, self.r1_std, self.r2_mean, self.r2_std = r1_chance_mean, r1_chance_std, r2_chance_mean, r2_chance_std
        self.min_r1_bands = min_r1_bands
        self.min_spacing, self.max_spacing = min_block_spacing, max_block_spacing
        self.optimized = optimized_intersections
        self.carve = carve_subblock_roads
        self.sub_inter = subblock_roads_have_intersections
        self.sub_chance = subblock_chance
        self.sub_type = ROAD_BY_NAME[subblock_road_type]
        self.min_sub = min_subblock_spacing
        self.hw_offset = highway_offset_from_edges
        self.light_range = traffic_light_range
        self.fwd_range = forward_traffic_light_range
        self.fwd_inter = forward_traffic_light_range_intersections
        self.entrance_level = block_entrance_road_level
        self.use_dummy = use_dummy_agents
        self.rain_enabled, self.enable_traffic = rain_enabled, enable_traffic

        self.x_min = self.y_min = self.wall + self.ring_w                      # city_model.py:91-94
        self.x_max = self.W - (self.wall + self.ring_w) - 1
        self.y_max = self.H - (self.wall + self.ring_w) - 1

        n = self.W * self.H
        self.ct = [WALL] * n                 # cell type; _place_thick_wall (city_model.py:315-319)
        self.dirs: List[list] = [[] for _ in range(n)]
        self.rtype: List[Optional[int]] = [None] * n     # CellAgent.road_type
        self.group: List[Optional[_Group]] = [None] * n  # CellAgent.intersection_group
        self.lights: Dict[Tuple[int, int], _Light] = {}
        self.ring_cells = set()              # _ring_road_cells (membership only)
        self.inter_cells: set = set()        # _intersection_cells: iteration order matters
        self.road_cells: dict = {}           # _road_cells
        self.blocks_data: List[dict] = []
        self.block_entrances: List[Tuple[int, int]] = []
        self.entrance_block: Dict[Tuple[int, int], int] = {}    # BlockEntrance cell -> block_id
        self.highway_entrances: List[Tuple[int, int]] = []
        self.highway_exits: List[Tuple[int, int]] = []
        self.groups: List[_Group] = []
        self.h_bands: List[Band] = []
        self.v_bands: List[Band] = []

    # ------------------------------------------------------------------ plane helpers
    def idx(self, x, y):
        return y * self.W + x

    def inb(self, x, y):
        return 0 <= x < self.W and 0 <= y < self.H

    def type_at(self, x, y):
        """cell type, or -1 outside the grid (get_cell_contents returns [] there)"""
        return self.ct[y * self.W + x] if (0 <= x < self.W and 0 <= y < self.H) else -1

    def place(self, x, y, t):
        """place_cell (city_model.py:1864-1870): a fresh cell - no arrows, road_type from the type"""
        i = y * self.W + x
        self.ct[i] = t
        self.dirs[i] = []
        self.rtype[i] = t if t in ROADS else None
        self.group[i] = None

    def inside_interior(self, x, y):
        return self.x_min <= x <= self.x_max and self.y_min <= y <= self.y_max

    @staticmethod
    def band_covering(index, bands):
        for b in bands:
            if b[0] <= index <= b[1]:
                return b
        return None

    def next_is_intersection(self, x, y, d):
        dx, dy = VEC[d]
        return self.ct[self.idx(x + dx, y + dy)] == INTERSECTION

    # ------------------------------------------------------------------ build sequence
    def build(self):
        self.sidewalk_inner_ring()
        self.clear_interior()
        self.build_roads_and_sidewalks()
        if self.carve:
            self.carve_subblock_roads()
        self.flood_fill_blocks()
        self.eliminate_dead_ends()
        self.upgrade_r2_to_intersections()
        self.place_block_entrances()
        self.remove_invalid_intersection_directions()
        self.add_entrance_directions()
        self.add_traffic_lights()
        self.create_light_groups()
        self.instantiate_blocks()
        return self

    def sidewalk_inner_ring(self):
        """city_model.py:329-360"""
        w, h, ws = self.W, self.H, self.wall
        for layer in range(self.ring_w):
            for y in (ws + layer, h - ws - 1 - layer):
                for x in range(ws, w - ws):
                    if self.ct[self.idx(x, y)] == WALL:
                        self.place(x, y, SIDEWALK)
            for x in (ws + layer, w - ws - 1 - layer):
                for y in range(ws, h - ws):
                    if self.ct[self.idx(x, y)] == WALL:
                        self.place(x, y, SIDEWALK)

    def clear_interior(self):
        n = self.x_max - self.x_min + 1           # nothing but Wall / Sidewalk so far: only the type changes
        for y in range(self.y_min, self.y_max + 1):
            a = self.idx(self.x_min, y)
            self.ct[a:a + n] = [NOTHING] * n

    # ---- bands (city_model.py:1076-1267) ----
    def choose_road_type(self):
        clip = lambda v: max(0.0, min(1.0, v))
        p1 = clip(self.rng.gauss(self.r1_mean, self.r1_std))
        p2 = clip(min(1.0 - p1, self.rng.gauss(self.r2_mean, self.r2_std)))
        r = self.rng.random()
        return R1 if r < p1 else (R2 if r < p1 + p2 else R3)

    def make_bands(self, lo, hi, horizontal) -> List[Band]:
        pair = [E, W] if horizontal else [N, S]
        bands: List[Band] = []
        cur, last_r3 = lo, None
        while cur <= hi:
            rt = self.choose_road_type()
            end = min(cur + THICKNESS[rt] - 1, hi)
            bdir = OPP[last_r3] if (rt == R3 and last_r3 is not None) else self.rng.choice(pair)
            bands.append((cur, end, rt, bdir))
            last_r3 = bdir if rt == R3 else None
            nxt = end + 1
            if nxt > hi:
                break
            gap_end = nxt + self.rng.randint(self.min_spacing, self.max_spacing) - 1
            if gap_end > hi:
                break
            cur = gap_end + 1
        if self.ring_type is not None:
            thick = THICKNESS[self.ring_type]
            if self.ring_type == R3:
                first_d, last_d = (E, W) if horizontal else (S, N)
            else:
                first_d = self.rng.choice(pair)
                last_d = self.rng.choice(pair)
            first = (lo, lo + thick - 1, self.ring_type, first_d)
            last = (hi - thick + 1, hi, self.ring_type, last_d)
            if not bands:
                bands.extend([first, last])
            elif len(bands) == 1:
                bands[0] = first
                if first != last:
                    bands.append(last)
            else:
                bands[0], bands[-1] = first, last
        return bands

    def force_one_highway(self, bands, total):
        thick = THICKNESS[R1]
        inset = self.x_min + self.hw_offset
        lo, hi = inset, total - thick - inset
        if lo > hi:
            lo, hi = 0, total - thick
            if hi < 0:
                return
        st = self.rng.randint(lo, hi)
        en = st + thick - 1
        bands.append((st, en, R1, None))
        bands.sort(key=lambda b: b[0])
        skip_lo, skip_hi = st - self.min_spacing, en + self.min_spacing
        bands[:] = [b for b in bands
                    if (b[2] == R1 and (b[0], b[1]) == (st, en)) or (b[1] < skip_lo or b[0] > skip_hi)]

    def ensure_minimum_highways(self, bands, total):
        def count():
            rng_ = range(1, len(bands) - 1) if (self.ring_type == R1 and len(bands) >= 2) else range(len(bands))
            return sum(1 for i in rng_ if bands[i][2] == R1)
        attempts = 0
        while count() < self.min_r1_bands and attempts < 20:
            self.force_one_highway(bands, total)
            attempts += 1

    # ---- lane arrows (city_model.py:1275-1368) ----
    def lane_dirs(self, x, y, rt, horizontal, off, size, bdir):
        if rt == R3:
            return [bdir]
        if rt == R2:
            if horizontal:
                return [E] if off == 0 else [W]
            return [S] if off == 0 else [N]
        if rt == R1:
            half = size // 2
            lo_side, hi_side = (S, N) if horizontal else (W, E)      # toward lower / higher offsets
            out = []
            if off < half:
                out.append(E if horizontal else S)
                if off > 0 and not self.next_is_intersection(x, y, lo_side):
                    out.append(lo_side)
                if off < half - 1 and not self.next_is_intersection(x, y, hi_side):
                    out.append(hi_side)
            else:
                out.append(W if horizontal else N)
                if off < size - 1 and not self.next_is_intersection(x, y, hi_side):
                    out.append(hi_side)
                if off > half and not self.next_is_intersection(x, y, lo_side):
                    out.append(lo_side)
            return out
        return []

    def corner_override(self, x, y, default):
        """city_model.py:498-558: fixed arrows in the four corner squares of an R2 ring"""
        if self.ring_type != R2:
            return default
        hb, ht, vl, vr = self.h_bands[0], self.h_bands[-1], self.v_bands[0], self.v_bands[-1]
        bottom, top = hb[0] <= y <= hb[1], ht[0] <= y <= ht[1]
        left, right = vl[0] <= x <= vl[1], vr[0] <= x <= vr[1]
        if not ((bottom or top) and (left or right)):
            return default
        if bottom and left:
            table, row, col = {(0, 0): E, (0, 1): E, (1, 0): S, (1, 1): N}, y - hb[0], x - vl[0]
        elif bottom and right:
            table, row, col = {(0, 0): E, (0, 1): N, (1, 0): W, (1, 1): N}, y - hb[0], x - vr[0]
        elif top and right:
            table, row, col = {(0, 0): S, (0, 1): N, (1, 0): W, (1, 1): W}, y - ht[0], x - vr[0]
        else:
            table, row, col = {(0, 0): S, (0, 1): E, (1, 0): S, (1, 1): W}, y - ht[0], x - vl[0]
        d = table.get((row, col))
        return [d] if d is not None else default

    # ---- intersection factory (city_model.py:211-306) ----
    def make_intersection(self, x, y):
        hb = self.band_covering(y, self.h_bands)
        vb = self.band_covering(x, self.v_bands)
        st = self.sub_type
        if not hb and (self.type_at(x, y) == st or self.type_at(x - 1, y) == st or self.type_at(x + 1, y) == st):
            hb = (y, y, st, None)
        if not vb and (self.type_at(x, y) == st or self.type_at(x, y - 1) == st or self.type_at(x, y + 1) == st):
            vb = (x, x, st, None)
        if not (hb and vb):
            return
        h_sz, h_off = hb[1] - hb[0] + 1, y - hb[0]
        v_sz, v_off = vb[1] - vb[0] + 1, x - vb[0]
        if self.optimized and ((h_sz == 1 and v_sz > 1) or (v_sz == 1 and h_sz > 1)):
            if h_sz > 1:
                rt, horizontal, off, sz, bdir = hb[2], True, h_off, h_sz, hb[3]
            else:
                rt, horizontal, off, sz, bdir = vb[2], False, v_off, v_sz, vb[3]
            if off not in (0, sz - 1):                       # inner lane of the wide road stays a road cell
                d = self.lane_dirs(x, y, rt, horizontal, off, sz, bdir)
                self.place(x, y, rt)
                self.dirs[self.idx(x, y)] = d
                self.inter_cells.discard((x, y))
                self.road_cells[(x, y)] = (rt, horizontal, off, sz, bdir)
                return
        if self.ct[self.idx(x, y)] == INTERSECTION:
            return
        self.place(x, y, INTERSECTION)
        self.dirs[self.idx(x, y)] = list(ALL_DIRS)
        self.inter_cells.add((x, y))

    # ---- roads (city_model.py:375-495) ----
    def build_roads_and_sidewalks(self):
        w, h = self.W, self.H
        self.h_bands = self.make_bands(self.y_min, self.y_max, True)
        self.v_bands = self.make_bands(self.x_min, self.x_max, False)
        self.ensure_minimum_highways(self.h_bands, h)
        self.ensure_minimum_highways(self.v_bands, w)
        if self.ring_type is not None:
            ft = THICKNESS[self.ring_type]
            in_ring_rows = lambda y: (self.y_min <= y < self.y_min + ft) or (self.y_max - ft + 1 <= y <= self.y_max)
            in_ring_cols = lambda x: (self.x_min <= x < self.x_min + ft) or (self.x_max - ft + 1 <= x <= self.x_max)
        vcover = [self.band_covering(x, self.v_bands) for x in range(w)]
        vcols = [x for x in range(w) if vcover[x]]
        for y in range(h):
            hb = self.band_covering(y, self.h_bands)
            for x in (range(w) if hb else vcols):
                vb = vcover[x]
                if hb and vb:
                    if (hb[2] != R1 or vb[2] != R1) and not self.inside_interior(x, y):
                        continue
                    if self.ring_type is not None and in_ring_rows(y) and in_ring_cols(x):
                        self.road_cells[(x, y)] = (hb[2], True, y - hb[0], hb[1] - hb[0] + 1, hb[3])
                        self.ring_cells.add((x, y))
                        continue
                    self.inter_cells.add((x, y))
                elif hb:
                    if hb[2] != R1 and not self.inside_interior(x, y):
                        continue
                    self.road_cells[(x, y)] = (hb[2], True, y - hb[0], hb[1] - hb[0] + 1, hb[3])
                elif vb:
                    if vb[2] != R1 and not self.inside_interior(x, y):
                        continue
                    self.road_cells[(x, y)] = (vb[2], False, x - vb[0], vb[1] - vb[0] + 1, vb[3])
        for (ix, iy) in list(self.inter_cells):
            self.make_intersection(ix, iy)
        for (rx, ry), (rt, horizontal, off, sz, bdir) in self.road_cells.items():
            if (rx, ry) in self.inter_cells:
                continue
            self.place(rx, ry, rt)
            self.dirs[self.idx(rx, ry)] = self.corner_override(rx, ry, self.lane_dirs(rx, ry, rt, horizontal, off, sz, bdir))
        # sidewalk around roads: each neighbour's outcome is independent of the visiting order
        is_pos = lambda p: p in self.road_cells or p in self.inter_cells
        for (rx, ry) in list(self.road_cells.keys()) + [p for p in self.inter_cells if p not in self.road_cells]:
            cur = self.ct[self.idx(rx, ry)]
            for dx, dy in NB4:
                nx, ny = rx + dx, ry + dy
                if not self.inb(nx, ny) or is_pos((nx, ny)):
                    continue
                t = self.ct[self.idx(nx, ny)]
                if t == NOTHING or (t == WALL and cur in (R1, HW_ENTRANCE, HW_EXIT)):
                    self.place(nx, ny, SIDEWALK)
        self.replace_boundary_highways()

    def replace_boundary_highways(self):
        """city_model.py:1370-1420; the visiting order of the edge set is the order of the entrance / exit lists"""
        w, h, ws = self.W, self.H, self.wall
        edge = set()
        for y in range(ws):
            for x in range(w):
                edge.add((x, y))
        for y in range(h - ws, h):
            for x in range(w):
                edge.add((x, y))
        for x in range(ws):
            for y in range(h):
                edge.add((x, y))
        for x in range(w - ws, w):
            for y in range(h):
                edge.add((x, y))
        for (ex, ey) in edge:
            i = self.idx(ex, ey)
            if self.ct[i] != R1 or not (ex in (0, w - 1) or ey in (0, h - 1)):
                continue
            old = list(self.dirs[i])
            inward = (ex == 0 and E in old) or (ex == w - 1 and W in old) or (ey == 0 and N in old) or (ey == h - 1 and S in old)
            self.place(ex, ey, HW_ENTRANCE if inward else HW_EXIT)
            self.dirs[i] = old
            (self.highway_entrances if inward else self.highway_exits).append((ex, ey))

    # ---- optional L-shaped roads inside large blocks (city_model.py:563-737) ----
    def carve_subblock_roads(self):
        st = self.sub_type

        def lay(x, y, arrow):
            i = self.idx(x, y)
            if self.ct[i] not in ROAD_LIKE:
                self.place(x, y, st)
                self.dirs[i] = [arrow]
            for dx, dy in NB4:
                if self.type_at(x + dx, y + dy) == NOTHING:
                    self.place(x + dx, y + dy, SIDEWALK)

        def extend(sx, sy, march, arrow):
            dx, dy = VEC[march]
            cx, cy = sx, sy
            while self.inb(cx, cy):
                i = self.idx(cx, cy)
                t = self.ct[i]
                if t in ROAD_LIKE:
                    if self.sub_inter:
                        self.make_intersection(cx, cy)
                        self.inter_cells.add((cx, cy))
                    elif arrow not in self.dirs[i]:
                        self.dirs[i].append(arrow)
                    break
                if t in (SIDEWALK, NOTHING):
                    lay(cx, cy, arrow)
                    cx, cy = cx + dx, cy + dy
                else:
                    break

        visited = set()
        for y in range(self.H):
            for x in range(self.W):
                if (x, y) in visited or self.ct[self.idx(x, y)] != NOTHING:
                    continue
                stack, region = [(x, y)], []
                while stack:
                    cx, cy = stack.pop()
                    if (cx, cy) in visited or self.ct[self.idx(cx, cy)] != NOTHING:
                        continue
                    visited.add((cx, cy))
                    region.append((cx, cy))
                    for dx, dy in NB4:
                        nx, ny = cx + dx, cy + dy
                        if self.inb(nx, ny) and (nx, ny) not in visited and self.ct[self.idx(nx, ny)] == NOTHING:
                            stack.append((nx, ny))
                if not region or self.rng.random() > self.sub_chance:
                    continue
                min_x, max_x = min(p[0] for p in region), max(p[0] for p in region)
                min_y, max_y = min(p[1] for p in region), max(p[1] for p in region)
                if max_x - min_x + 1 < 2 * self.min_sub + 1 or max_y - min_y + 1 < 2 * self.min_sub + 1:
                    continue
                for _ in range(20):
                    px = self.rng.randint(min_x + self.min_sub, max_x - self.min_sub)
                    py = self.rng.randint(min_y + self.min_sub, max_y - self.min_sub)
                    hor = self.rng.choice([W, E])
                    ver = self.rng.choice([N, S])
                    small_w = (px - min_x) if hor == W else (max_x - px)
                    small_h = (py - min_y) if ver == S else (max_y - py)
                    if small_w >= self.min_sub and small_h >= self.min_sub:
                        break
                else:
                    continue
                inbound_horizontal = self.rng.choice([True, False])      # ("horizontal","vertical") first
                h_arrow = OPP[hor] if inbound_horizontal else hor
                v_arrow = ver if inbound_horizontal else OPP[ver]
                xs = range(px - 1, min_x - 1, -1) if hor == W else range(px + 1, max_x + 1)
                hx_end = min_x if hor == W else max_x
                for hx in xs:
                    lay(hx, py, h_arrow)
                ys = range(py, min_y - 1, -1) if ver == S else range(py, max_y + 1)
                vy_end = min_y if ver == S else max_y
                for vy in ys:
                    lay(px, vy, v_arrow)
                self.dirs[self.idx(px, py)] = [v_arrow if inbound_horizontal else h_arrow]   # pivot: outbound arrow only
                extend(hx_end + VEC[hor][0], py + VEC[hor][1], hor, h_arrow)
                extend(px + VEC[ver][0], vy_end + VEC[ver][1], ver, v_arrow)
                for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1)):
                    nx, ny = px + dx, py + dy
                    if self.inb(nx, ny):
                        t = self.ct[self.idx(nx, ny)]
                        if t not in ROAD_LIKE and t != WALL:
                            self.place(nx, ny, SIDEWALK)

    # ---- blocks (city_model.py:742-806) ----
    def flood_fill_blocks(self):
        visited = set()
        ct, W = self.ct, self.W
        for y in range(self.H):
            row = y * W
            for x in range(W):
                if ct[row + x] != NOTHING:        # (a visited cell already carries its block type)
                    continue
                stack, region = [(x, y)], []
                while stack:
                    cx, cy = stack.pop()
                    if (cx, cy) in visited or ct[cy * W + cx] != NOTHING:
                        continue
                    visited.add((cx, cy))
                    region.append((cx, cy))
                    for nx, ny in ((cx + 1, cy), (cx - 1, cy), (cx, cy + 1), (cx, cy - 1)):
                        if self.inb(nx, ny) and (nx, ny) not in visited and ct[ny * W + nx] == NOTHING:
                            stack.append((nx, ny))
                if not region:
                    continue
                w_bb = max(p[0] for p in region) - min(p[0] for p in region) + 1
                h_bb = max(p[1] for p in region) - min(p[1] for p in region) + 1
                if w_bb < 3 or h_bb < 3:
                    bt = EMPTY_BLOCK
                else:
                    bt = BLOCK0 + self.rng.choices(range(N_BLOCK_TYPES), weights=BLOCK_WEIGHTS, k=1)[0]
                for bx, by in region:
                    self.place(bx, by, bt)
                # the ring = in-bounds 4-neighbours outside the region.  No other cell of this block type can touch the region
                # (it would have been part of the same blob), so "outside" is a type test on the plane.
                ring = set()                     # iteration order reaches the entrance placement below
                H = self.H
                for bx, by in region:
                    for nx, ny in ((bx + 1, by), (bx - 1, by), (bx, by + 1), (bx, by - 1)):
                        if 0 <= nx < W and 0 <= ny < H and ct[ny * W + nx] != bt:
                            ring.add((nx, ny))
                for sx, sy in ring:
                    if self.ct[self.idx(sx, sy)] == NOTHING:
                        self.place(sx, sy, SIDEWALK)
                self.blocks_data.append(dict(block_id=len(self.blocks_data) + 1, block_type=bt, region=region, ring=list(ring)))

    def eliminate_dead_ends(self):
        """city_model.py:811-830"""
        changed = True
        while changed:
            changed = False
            ct, W = self.ct, self.W
            for y in range(self.H):
                row = y * W
                for x in range(W):
                    if ct[row + x] in REMOVABLE_DEAD_END:
                        k = sum(1 for dx, dy in NB4 if self.type_at(x + dx, y + dy) in ROAD_LIKE)
                        if k < 2:
                            self.place(x, y, SIDEWALK)
                            changed = True

    def upgrade_r2_to_intersections(self):
        """city_model.py:842-879"""
        if self.ring_type == R2:
            hb, ht, vl, vr = self.h_bands[0], self.h_bands[-1], self.v_bands[0], self.v_bands[-1]
        for y in range(self.H):
            for x in range(self.W):
                if self.ct[self.idx(x, y)] != R2:
                    continue
                if self.ring_type == R2 and (hb[0] <= y <= hb[1] or ht[0] <= y <= ht[1]) and (vl[0] <= x <= vl[1] or vr[0] <= x <= vr[1]):
                    continue
                if sum(1 for dx, dy in NB4 if self.type_at(x + dx, y + dy) == SIDEWALK) >= 2:
                    self.make_intersection(x, y)

    def place_block_entrances(self):
        """city_model.py:884-963"""
        disallowed = [set(), {R3}, {R2, R3}][min(self.entrance_level, 2)]
        for info in self.blocks_data:
            if info["block_type"] == EMPTY_BLOCK:
                continue
            ring = [(x, y) for (x, y) in info["ring"]
                    if any(self.type_at(x + dx, y + dy) in TOUCHES_ROAD for dx, dy in NB4)]
            if not ring:
                continue
            if self.entrance_level > 0:
                preferred = []
                for cx, cy in ring:
                    adj = {self.type_at(cx + dx, cy + dy) for dx, dy in NB4} & set(ROADS)
                    if any(rt not in disallowed for rt in adj):
                        preferred.append((cx, cy))
                if preferred:
                    ring = preferred
            ring_set = set(ring)
            runs = []
            while ring_set:
                start = ring_set.pop()
                stack, run = [start], [start]
                while stack:
                    x, y = stack.pop()
                    for nx, ny in ((x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)):
                        if (nx, ny) in ring_set:
                            ring_set.remove((nx, ny))
                            stack.append((nx, ny))
                            run.append((nx, ny))
                runs.append(run)
            longest = max(len(r) for r in runs)
            run = self.rng.choice([r for r in runs if len(r) == longest])
            if all(y == run[0][1] for _, y in run):
                run.sort(key=lambda p: p[0])
            elif all(x == run[0][0] for x, _ in run):
                run.sort(key=lambda p: p[1])
            else:
                run.sort()
            cx, cy = run[len(run) // 2]
            self.place(cx, cy, BLOCK_ENTRANCE)
            self.block_entrances.append((cx, cy))
            self.entrance_block[(cx, cy)] = info["block_id"]

    def remove_invalid_intersection_directions(self):
        """city_model.py:969-1012"""
        for y in range(self.H):
            for x in range(self.W):
                i = self.idx(x, y)
                if self.ct[i] != INTERSECTION:
                    continue
                keep = []
                for d in self.dirs[i]:
                    nx, ny = x + VEC[d][0], y + VEC[d][1]
                    t = self.type_at(nx, ny)
                    if t not in ROAD_LIKE:
                        continue
                    if t == INTERSECTION or d in self.dirs[self.idx(nx, ny)]:
                        keep.append(d)
                self.dirs[i] = keep

    def add_entrance_directions(self):
        """city_model.py:1035-1070"""
        toward = {(1, 0): E, (-1, 0): W, (0, 1): N, (0, -1): S}      # (entrance - road) -> arrow on the road
        for y in range(self.H):
            for x in range(self.W):
                if self.ct[self.idx(x, y)] != BLOCK_ENTRANCE:
                    continue
                own = []
                for nx, ny in ((x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)):
                    if self.type_at(nx, ny) in ROAD_LIKE:
                        need = toward[(x - nx, y - ny)]
                        nd = self.dirs[self.idx(nx, ny)]
                        if need not in nd:
                            nd.append(need)
                        own.append(OPP[need])
                self.dirs[self.idx(x, y)] = own

    # ---- traffic lights (city_model.py:1422-1584) ----
    def leads_to(self, src, dst):
        """CellAgent.leads_to (cell.py:200-226): is dst reachable from src along the arrows"""
        if src == dst:
            return True
        for d in self.dirs[src[1] * self.W + src[0]]:     # the usual case: the very next cell
            if (src[0] + VEC[d][0], src[1] + VEC[d][1]) == dst:
                return True
        seen = {src}
        frontier = [src]
        while frontier:
            nxt = []
            for (cx, cy) in frontier:
                for d in self.dirs[self.idx(cx, cy)]:
                    p = (cx + VEC[d][0], cy + VEC[d][1])
                    if p == dst:
                        return True
                    if not self.inb(*p) or p in seen:
                        continue
                    seen.add(p)
                    nxt.append(p)
            frontier = nxt
        return False

    def directly_leads_to(self, src, dst):
        return any((src[0] + VEC[d][0], src[1] + VEC[d][1]) == dst for d in self.dirs[self.idx(*src)])

    def scan_reverse(self, road, scan_dirs, orig_type, tl, depth):
        for rd in [OPP[d] for d in scan_dirs]:
            bx, by = road[0] + VEC[rd][0], road[1] + VEC[rd][1]
            while depth <= self.light_range:      # the depth carries over from one direction to the next
                if self.inb(bx, by) and self.ct[self.idx(bx, by)] == orig_type and self.leads_to((bx, by), road):
                    tl.incoming.append((bx, by))
                    bx, by = bx + VEC[rd][0], by + VEC[rd][1]
                    depth += 1
                else:
                    break

    def scan_forward(self, road, scan_dirs, orig_type, tl, depth):
        for rd in scan_dirs:
            bx, by = road[0] + VEC[rd][0], road[1] + VEC[rd][1]
            cur = depth
            while cur <= self.light_range:
                if not self.inb(bx, by):
                    break
                t = self.ct[self.idx(bx, by)]
                if t == INTERSECTION:
                    if self.fwd_inter == "Include in Range":
                        tl.outgoing.append((bx, by))
                        cur += 1
                    elif self.fwd_inter == "Include as Extra":
                        tl.outgoing.append((bx, by))
                    bx, by = bx + VEC[rd][0], by + VEC[rd][1]
                elif t == orig_type:
                    if self.directly_leads_to((bx, by), road):
                        self.scan_forward((bx, by), scan_dirs, orig_type, tl, cur + 1)
                    elif rd in self.dirs[self.idx(bx, by)]:
                        tl.outgoing.append((bx, by))
                        cur += 1
                    bx, by = bx + VEC[rd][0], by + VEC[rd][1]
                else:
                    break

    def assign_light(self, road, x, y, orig_type, scan_dirs):
        t = self.type_at(x, y)
        if t == SIDEWALK:
            self.place(x, y, LIGHT)
            self.lights[(x, y)] = _Light((x, y))
        elif t != LIGHT:
            return
        tl = self.lights[(x, y)]
        tl.controlled.append(road)
        self.scan_reverse(road, scan_dirs, orig_type, tl, 0)
        if self.fwd_range:
            self.scan_forward(road, scan_dirs, orig_type, tl, 0)

    def add_traffic_lights(self):
        for x in range(self.W):
            for y in range(self.H):
                i = self.idx(x, y)
                orig = self.ct[i]
                if orig not in ROAD_LIKE_NO_INTER:
                    continue
                rdirs = list(self.dirs[i])
                for d in rdirs:
                    if self.type_at(x + VEC[d][0], y + VEC[d][1]) != INTERSECTION:
                        continue
                    self.place(x, y, CONTROLLED)
                    self.dirs[i] = rdirs
                    self.rtype[i] = orig                 # controlled_road.road_type = original cell_type
                    right = []
                    for cd in rdirs:
                        dx, dy = VEC[RIGHT_OF[cd]]
                        right.append((x + dx, y + dy))
                    for vx, vy in list(set(right)):
                        if not self.inb(vx, vy):
                            continue
                        t = self.ct[self.idx(vx, vy)]
                        if t == CONTROLLED or t == orig:
                            if not any(dd in rdirs for dd in self.dirs[self.idx(vx, vy)]):
                                continue
                            fx, fy = 2 * vx - x, 2 * vy - y
                            if self.inb(fx, fy):
                                self.assign_light((x, y), fx, fy, orig, rdirs)
                        self.assign_light((x, y), vx, vy, orig, rdirs)
                    break

    # ---- light groups (city_model.py:1587-1650, intersection_light_group.py:116-282) ----
    def create_light_groups(self):
        visited = set()
        for seed in self.inter_cells:
            if seed in visited:
                continue
            stack, cluster = [seed], []
            while stack:
                x, y = stack.pop()
                if (x, y) in visited or (x, y) not in self.inter_cells:
                    continue
                visited.add((x, y))
                cluster.append((x, y))
                for dx, dy in NB4:
                    p = (x + dx, y + dy)
                    if p in self.inter_cells and p not in visited:
                        stack.append(p)
            if not cluster:
                continue
            min_x, max_x = min(p[0] for p in cluster), max(p[0] for p in cluster)
            min_y, max_y = min(p[1] for p in cluster), max(p[1] for p in cluster)
            lights = [self.lights[c] for c in ((min_x - 1, min_y - 1), (max_x + 1, min_y - 1), (min_x - 1, max_y + 1), (max_x + 1, max_y + 1))
                      if self.type_at(*c) == LIGHT]
            if not lights:
                continue
            g = _Group(lights, [])
            self.populate_links(g)              # inside the constructor: own cells are not tagged yet
            g.ctor_neighbors = dict(g.neighbors)
            g.ctor_intermediate = list(g.intermediate)
            for tl in lights:
                for (bx, by) in tl.incoming + tl.outgoing:
                    bd = self.dirs[self.idx(bx, by)]
                    if N in bd or S in bd:
                        (g.ns_in if by < tl.pos[1] else g.ns_out).append((bx, by))
                    elif E in bd or W in bd:
                        (g.ew_in if bx < tl.pos[0] else g.ew_out).append((bx, by))
            if not any(tl.incoming or tl.outgoing for tl in lights):
                # the constructor averages a penalty over the lanes its lights watch (intersection_light_group.py:160-166);
                # with none the reference's CityModel() fails, and so does this
                raise ZeroDivisionError("light group without watched lanes: the reference divides by their count")
            g.cells = cluster
            self.groups.append(g)
            for (ix, iy) in cluster:
                self.group[self.idx(ix, iy)] = g

    def blocks_all_lanes(self, ix, iy, d):
        is_int = lambda x, y: self.type_at(x, y) == INTERSECTION
        band = lambda k, bands: self.band_covering(k, bands) or (k, k, None, None)
        if d in (N, S):
            v0, v1 = band(ix, self.v_bands)[:2]
            if v1 == v0:
                h0, h1 = band(iy, self.h_bands)[:2]
                return is_int(v0, iy) and (h1 != h0 or is_int(ix, h0))
            return all(is_int(xx, iy) for xx in range(v0, v1 + 1))
        h0, h1 = band(iy, self.h_bands)[:2]
        if h1 == h0:
            v0, v1 = band(ix, self.v_bands)[:2]
            return is_int(ix, h0) and (v1 != v0 or is_int(v0, iy))
        return all(is_int(ix, yy) for yy in range(h0, h1 + 1))

    def populate_links(self, g: _Group, max_depth=1000):
        g.neighbors = {}
        g.intermediate = []
        starts = []
        for tl in g.lights:
            lx, ly = tl.pos
            for dx, dy in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
                if self.type_at(lx + dx, ly + dy) == INTERSECTION:
                    starts.append((lx + dx, ly + dy))
        ct, grp, Wd, Ht = self.ct, self.group, self.W, self.H
        for cx, cy in starts:
            for d in ALL_DIRS:
                x, y, steps = cx, cy, 0
                dx, dy = VEC[d]
                while steps < max_depth:
                    x, y = x + dx, y + dy
                    if not (0 <= x < Wd and 0 <= y < Ht):
                        break
                    i = y * Wd + x
                    og = grp[i] if ct[i] == INTERSECTION else None
                    if og is None or og is g:
                        steps += 1
                        continue
                    if d not in og.blocks_cache:
                        og.blocks_cache[d] = self.blocks_all_lanes(x, y, d)
                    if og.blocks_cache[d]:
                        g.neighbors[d] = og
                        break
                    if not any(og is o for o in g.intermediate):
                        g.intermediate.append(og)
                    steps += 1
        axis = {N: [], S: [], E: [], W: []}
        for tl in g.lights:
            for (cbx, cby) in tl.controlled:
                for d in self.dirs[self.idx(cbx, cby)]:
                    nx, ny = cbx + VEC[d][0], cby + VEC[d][1]
                    if self.inb(nx, ny) and self.ct[self.idx(nx, ny)] == INTERSECTION and self.group[self.idx(nx, ny)] is g:
                        axis[d].append(tl)
                        break
        def uniq(seq):
            out = []
            for tl in seq:
                if not any(tl is o for o in out):
                    out.append(tl)
            return out
        g.ns_lights = uniq(axis[N] + axis[S])
        g.ew_lights = uniq(axis[E] + axis[W])

    # ---- city blocks (city_model.py:1661-1745) ----
    def instantiate_blocks(self):
        self.blocks = []
        for info in self.blocks_data:
            if info["block_type"] == EMPTY_BLOCK:
                continue
            sidewalks, entrances, seen = [], [], set()
            ct, W, H = self.ct, self.W, self.H
            for x, y in set(info["region"]):
                for dx, dy in NB4:
                    nx, ny = x + dx, y + dy
                    if not (0 <= nx < W and 0 <= ny < H):
                        continue
                    t = ct[ny * W + nx]
                    if t != SIDEWALK and t != BLOCK_ENTRANCE:
                        continue
                    p = (nx, ny)
                    if t == SIDEWALK and p not in seen:
                        seen.add(p)
                        sidewalks.append(p)
                    elif t == BLOCK_ENTRANCE and p not in seen:
                        seen.add(p)
                        entrances.append(p)
            self.blocks.append(dict(id=info["block_id"], type=info["block_type"] - BLOCK0, inner=len(info["region"]), sidewalks=sidewalks,
                                    entrances=entrances, service=self.ranked_service_cells(sidewalks, entrances)))

    def ranked_service_cells(self, sidewalks, entrances):
        """CityBlock.get_service_road_cell's static part (city_block.py:152-202): candidate road cells beside the block's
        sidewalks, minus those beside an entrance, stable-sorted by distance to the nearest entrance; ties stay in
        set-iteration order, hence the same set operations in the same sequence."""
        cand = set()
        for sx, sy in sidewalks:
            for dx, dy in NB4:
                if self.type_at(sx + dx, sy + dy) in ROADS:
                    cand.add((sx + dx, sy + dy))
        if not cand:
            return []
        for ex, ey in entrances:
            for dx, dy in NB4:
                cand.discard((ex + dx, ey + dy))
        if not cand or not entrances:
            return []
        return sorted(cand, key=lambda rc: min(abs(rc[0] - ex) + abs(rc[1] - ey) for ex, ey in entrances))

    # ------------------------------------------------------------------ output tables
    def tables(self) -> dict:
        """Same keys and layouts as tests/golden/make_golden.py::world_tables."""
        W, H = self.W, self.H
        ct = np.asarray(self.ct, dtype=np.int16).reshape(H, W)
        allowed = np.zeros((H, W), np.uint8)
        for i, d in enumerate(self.dirs):
            if d:
                b = 0
                for k in d:
                    b |= 1 << k
                allowed[i // W, i % W] = b
        is_road = np.isin(ct, list(ROAD_LIKE)).astype(np.int8)
        inter = (ct == INTERSECTION).astype(np.int8)
        road_type = np.zeros((H, W), np.int8)               # _build_simple_maps (city_model.py:2151-2200)
        rt = np.asarray([(-1 if r is None else r) for r in self.rtype], dtype=np.int16).reshape(H, W)
        lane = (ct == R1) | (ct == R2) | (ct == R3)     # ControlledRoad is not in ROAD_LIKE_TYPES: the whole branch is skipped for it
        road_type[lane & (rt == R1)] = 1
        road_type[lane & (rt == R2)] = 2
        road_type[lane & (rt == R3)] = 3
        for (x, y) in self.ring_cells:
            if lane[y, x] and rt[y, x] == R2:
                road_type[y, x] = 1
        road_type[(ct == HW_ENTRANCE) | (ct == HW_EXIT) | (ct == BLOCK_ENTRANCE) | (ct == INTERSECTION)] = 1
        out = dict(width=np.int32(W), height=np.int32(H), allowed_dirs_map=allowed, is_road_map=is_road,
                   road_type_map=road_type, intersection_map=inter, stop_map0=np.zeros((H, W), np.int8))
        # what the portrayal layer shows per cell: CellAgent.cell_type (index into CELL_TYPE_NAMES) and .block_id (0 = none)
        out["cell_type_map"] = ct.astype(np.int8)
        block_id = np.zeros((H, W), np.int32)
        for (x, y), b in self.entrance_block.items():
            if ct[y, x] == BLOCK_ENTRANCE:
                block_id[y, x] = b
        for info in self.blocks_data:
            if info["block_type"] != EMPTY_BLOCK:
                for (x, y) in info["region"]:
                    block_id[y, x] = info["block_id"]
        out["block_id_map"] = block_id

        def ragged(rows, width):
            off, flat = [0], []
            for r in rows:
                flat.extend(r)
                off.append(len(flat) // width if width > 1 else len(flat))
            arr = np.asarray(flat, dtype=np.int32)
            if width > 1:
                arr = arr.reshape(-1, width)
            return np.asarray(off, dtype=np.int32), arr

        groups = self.groups
        gidx = {id(g): i for i, g in enumerate(groups)}
        lights_flat, light_off = [], [0]
        for g in groups:
            lights_flat.extend(g.lights)
            light_off.append(len(lights_flat))
        lidx = {id(tl): i for i, tl in enumerate(lights_flat)}
        out["g_light_off"] = np.asarray(light_off, np.int32)
        out["light_xy"] = np.asarray([tl.pos for tl in lights_flat], np.int32).reshape(-1, 2)
        out["light_ctrl_off"], out["light_ctrl_xy"] = ragged([[c for p in tl.controlled for c in p] for tl in lights_flat], 2)

        def neighbor_table(get):
            nb = np.full((len(groups), 4, 2), -1, np.int32)
            for i, g in enumerate(groups):
                for k, (d, ng) in enumerate(get(g).items()):
                    nb[i, k] = (d, gidx.get(id(ng), -1))
            return nb
        out["g_neighbors_ctor"] = neighbor_table(lambda g: g.ctor_neighbors)
        # intermediate_groups is a set of agents in the reference (no order to keep): ascending group indices
        out["g_intermediate_ctor_off"], out["g_intermediate_ctor"] = ragged([sorted(gidx[id(o)] for o in g.ctor_intermediate) for g in groups], 1)
        # the first phase change re-runs populate_links with every cell tagged: tables as they stand after that
        for g in groups:
            self.populate_links(g)
        out["g_ns_lights_off"], out["g_ns_lights"] = ragged([[lidx[id(tl)] for tl in g.ns_lights] for g in groups], 1)
        out["g_ew_lights_off"], out["g_ew_lights"] = ragged([[lidx[id(tl)] for tl in g.ew_lights] for g in groups], 1)
        out["g_neighbors"] = neighbor_table(lambda g: g.neighbors)
        out["g_intermediate_off"], out["g_intermediate"] = ragged([sorted(gidx[id(o)] for o in g.intermediate) for g in groups], 1)
        out["g_icell_off"], out["g_icell_xy"] = ragged([[c for p in g.cells for c in p] for g in groups], 2)
        for nm in ("ns_in", "ns_out", "ew_in", "ew_out"):
            out[f"g_{nm}_off"], out[f"g_{nm}_xy"] = ragged([[c for p in getattr(g, nm) for c in p] for g in groups], 2)
        kinds = [0] * len(groups) + [1] * len(self.blocks) + ([4] * (W * H) if self.use_dummy else [])
        if self.rain_enabled:
            kinds.append(2)
        if self.enable_traffic:
            kinds.append(3)
        out["schedule_kinds0"] = np.asarray(kinds, np.int8)
        out["blk_id"] = np.asarray([b["id"] for b in self.blocks], np.int32)       # keys of CityModel.city_blocks
        out["blk_type"] = np.asarray([b["type"] for b in self.blocks], np.int32)
        out["blk_entr_off"], out["blk_entr_xy"] = ragged([[c for p in b["entrances"] for c in p] for b in self.blocks], 2)
        out["blk_inner_cells"] = np.asarray([b["inner"] for b in self.blocks], np.int32)
        out["blk_service_off"], out["blk_service_xy"] = ragged([[c for p in b["service"] for c in p] for b in self.blocks], 2)
        out["block_entrances_xy"] = np.asarray(self.block_entrances, np.int32).reshape(-1, 2)
        out["highway_entrances_xy"] = np.asarray(self.highway_entrances, np.int32).reshape(-1, 2)
        out["highway_exits_xy"] = np.asarray(self.highway_exits, np.int32).reshape(-1, 2)
        out["global_rng_state"] = np.asarray(self.rng.getstate()[1], dtype=np.uint32)
        return out


def generate_world(width=200, height=200, seed=None, **options) -> dict:
    """`CityModel(width, height, seed=seed, **options)` after `random.seed(seed)`, as world tables.

    `options` are the reference constructor's keyword arguments (city_model.py:27-53) plus the `Defaults` switches that
    reach the tables: `block_entrance_road_level`, `rain_enabled`, `enable_traffic`."""
    import gc
    was_enabled = gc.isenabled()
    gc.disable()          # millions of small tracked objects and no garbage: the cyclic collector only costs (1.6x at 1024^2)
    try:
        return WorldBuilder(width, height, seed=seed, **options).build().tables()
    finally:
        if was_enabled:
            gc.enable()
