"""Synthetic, procedurally generated city grids for benchmarks and large-size parity tests.

This is NOT the seed-compatible restatement of the reference's world generator - that is worldgen.py (exact, but
interpreted: 4 s at 512², minutes at 4096²; the reference's own generator takes 32 s at 512²).
It produces worlds with the same *structure* the hot path consumes, following the reference's
layout rules at band granularity:

  * wall ring + sidewalk ring, interior road bands per axis with random spacing in
    [min_block_spacing, max_block_spacing] (city_model.py:1076-1128), forced ring road of type R2
    (1130-1176), at least `min_r1_bands` highways per axis that run through the walls to the
    map edge (1216-1273, 375-495);
  * lane directions for R1/R2/R3 with right-hand traffic and R1 lane shifts (1275-1368);
  * full-rectangle intersections where bands cross (the reference's non-"optimised" mode),
    intersection directions pruned like `_remove_invalid_intersection_directions` (969-1010);
  * road-type map values of `_build_simple_maps` (2151-2199);
  * ControlledRoad cells in front of every intersection, one traffic light on each diagonal
    corner, incoming lane cells up to `traffic_light_range` behind the stop line, classified
    into ns_in/ns_out/ew_in/ew_out exactly like initialize_cached_lane_coords
    (intersection_light_group.py:141-154), opposite pairs like populate_links (243-279).

Outputs use the same table keys as the golden world tables (tests/golden/make_golden.py).
Routes are random walks along the allowed directions (no A* at setup), so they are valid
4-adjacent chains but not the reference planner's choice; that is immaterial for the
car-following / movement / light kernels, which only consume the path.
"""
from __future__ import annotations

import numpy as np

N_, E_, S_, W_ = 1, 2, 4, 8
DX = np.array([0, 1, 0, -1])
DY = np.array([1, 0, -1, 0])
BIT = np.array([N_, E_, S_, W_])
THICK = {1: 4, 2: 2, 3: 1}


def _make_bands(rng, lo, hi, min_sp, max_sp, p_r1, p_r2, min_r1):
    """[(start, end, type, dir)] along one axis; dir: +1 / -1 for one-way R3, 0 otherwise."""
    bands = [(lo, lo + 1, 2, 0, True)]
    cur = lo + 2
    last_r3 = 0
    while True:
        cur += int(rng.randint(min_sp, max_sp + 1))
        u = rng.rand()
        t = 1 if u < p_r1 else (2 if u < p_r1 + p_r2 else 3)
        end = cur + THICK[t] - 1
        if end + min_sp > hi - 2:
            break
        d = 0
        if t == 3:
            d = -last_r3 if last_r3 else (1 if rng.rand() < 0.5 else -1)
            last_r3 = d
        else:
            last_r3 = 0
        bands.append((cur, end, t, d, False))
        cur = end + 1
    bands.append((hi - 1, hi, 2, 0, True))
    # at least `min_r1` highways: widen interior R2 bands (spacing >= min_sp keeps them apart)
    inner = [i for i, b in enumerate(bands) if not b[4]]
    have = sum(1 for i in inner if bands[i][2] == 1)
    cand = [i for i in inner if bands[i][2] == 2]
    rng.shuffle(cand)
    for i in cand:
        if have >= min_r1:
            break
        s, e, t, d, ring = bands[i]
        nxt = bands[i + 1][0]
        if e + 2 + 3 <= nxt:
            bands[i] = (s, e + 2, 1, 0, False)
            have += 1
    return bands


def _axis_arrays(bands, n):
    typ = np.zeros(n, np.int8)
    off = np.zeros(n, np.int16)
    size = np.zeros(n, np.int16)
    bdir = np.zeros(n, np.int8)
    bid = np.full(n, -1, np.int32)
    ring = np.zeros(n, bool)
    for i, (s, e, t, d, r) in enumerate(bands):
        typ[s:e + 1] = t
        off[s:e + 1] = np.arange(e - s + 1)
        size[s:e + 1] = e - s + 1
        bdir[s:e + 1] = d
        bid[s:e + 1] = i
        ring[s:e + 1] = r
    return typ, off, size, bdir, bid, ring


def _carve_subblock_roads(seed, allowed, is_road, road_type, inter, hb, vb, chance, min_sub):
    """L-shaped one-lane roads inside the large blocks, following `_carve_subblock_roads` (city_model.py:563-737) with
    `subblock_roads_have_intersections=False`: a pivot at least `min_sub` cells from every side of the block's interior, one
    arm to the western or eastern edge, one to the northern or southern, traffic flowing in along one arm and out along the
    other; the arms run through the sidewalk up to the road beside the block, whose cell there receives the arm's arrow as
    an extra direction (666-676).  Returns the number of blocks carved."""
    rng = np.random.RandomState(seed)
    opp = {N_: S_, S_: N_, E_: W_, W_: E_}
    H, W = allowed.shape
    carved = 0
    for i in range(len(hb) - 1):
        y0, y1 = hb[i][1] + 1, hb[i + 1][0] - 1                  # the gap between two bands, sidewalks included
        if (y1 - 1) - (y0 + 1) + 1 < 2 * min_sub + 1:
            continue
        for j in range(len(vb) - 1):
            x0, x1 = vb[j][1] + 1, vb[j + 1][0] - 1
            if (x1 - 1) - (x0 + 1) + 1 < 2 * min_sub + 1 or rng.rand() > chance:
                continue
            px = int(rng.randint(x0 + 1 + min_sub, x1 - 1 - min_sub + 1))
            py = int(rng.randint(y0 + 1 + min_sub, y1 - 1 - min_sub + 1))
            hor = W_ if rng.rand() < 0.5 else E_
            ver = S_ if rng.rand() < 0.5 else N_
            inbound_h = rng.rand() < 0.5
            h_arrow = opp[hor] if inbound_h else hor
            v_arrow = ver if inbound_h else opp[ver]
            xs = range(x0, px) if hor == W_ else range(px + 1, x1 + 1)
            ys = range(y0, py + 1) if ver == S_ else range(py, y1 + 1)
            for x in xs:
                allowed[py, x] = h_arrow
            for y in ys:
                allowed[y, px] = v_arrow
            allowed[py, px] = v_arrow if inbound_h else h_arrow   # the pivot: its outbound arrow only (646)
            is_road[py, xs.start:xs.stop] = True
            is_road[ys.start:ys.stop, px] = True
            road_type[py, xs.start:xs.stop] = 3
            road_type[ys.start:ys.stop, px] = 3
            ex, ey = (x0 - 1 if hor == W_ else x1 + 1), (y0 - 1 if ver == S_ else y1 + 1)
            if is_road[py, ex] and not inter[py, ex]:
                allowed[py, ex] |= h_arrow
            if is_road[ey, px] and not inter[ey, px]:
                allowed[ey, px] |= v_arrow
            carved += 1
    return carved


def generate(width: int, height: int, seed: int = 1, wall_thickness: int = 15, sidewalk_ring_width: int = 2,
             min_block_spacing: int = 6, max_block_spacing: int = 18, r1_chance: float = 0.15,
             r2_chance: float = 0.70, min_r1_bands: int = 2, traffic_light_range: int = 10,
             carve_subblock_roads: bool = False, subblock_chance: float = 0.3, min_subblock_spacing: int = 5) -> dict:
    W, H = int(width), int(height)
    rng = np.random.RandomState(seed)
    m = wall_thickness + sidewalk_ring_width
    if W < 2 * m + 24 or H < 2 * m + 24:
        raise ValueError("grid too small for the wall/sidewalk ring plus a road network")
    ix0, ix1, iy0, iy1 = m, W - m - 1, m, H - m - 1
    hb = _make_bands(rng, iy0, iy1, min_block_spacing, max_block_spacing, r1_chance, r2_chance, min_r1_bands)
    vb = _make_bands(rng, ix0, ix1, min_block_spacing, max_block_spacing, r1_chance, r2_chance, min_r1_bands)
    ht, ho, hs, hd, hid, hring = _axis_arrays(hb, H)   # indexed by y
    vt, vo, vs, vd, vid, vring = _axis_arrays(vb, W)   # indexed by x

    Y, X = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    HT, VT = ht[:, None], vt[None, :]
    in_x = (X >= ix0) & (X <= ix1)
    in_y = (Y >= iy0) & (Y <= iy1)
    h_any, v_any = HT > 0, VT > 0
    corner = hring[:, None] & vring[None, :]
    inter = h_any & v_any & ~corner
    h_road = h_any & ~inter & ((HT == 1) | in_x) & ~(v_any & ~corner)
    v_road = v_any & ~inter & ((VT == 1) | in_y) & ~(h_any & ~corner)
    # ring corners: regular road cells, horizontal-band info (city_model.py:436-442) with fixed arrows
    h_road |= corner
    v_road &= ~corner
    is_road = (inter | h_road | v_road)

    allowed = np.zeros((H, W), np.uint8)
    HO, HS, HD = ho[:, None], hs[:, None], hd[:, None]
    VO, VS, VD = vo[None, :], vs[None, :], vd[None, :]
    # horizontal roads
    r3h = h_road & (HT == 3)
    allowed[r3h & (HD > 0)] = E_
    allowed[r3h & (HD < 0)] = W_
    r2h = h_road & (HT == 2)
    allowed[r2h & (HO == 0)] = E_
    allowed[r2h & (HO == 1)] = W_
    r1h = h_road & (HT == 1)
    allowed[r1h & (HO < 2)] = E_
    allowed[r1h & (HO >= 2)] = W_
    # vertical roads
    r3v = v_road & (VT == 3)
    allowed[r3v & (VD > 0)] = N_
    allowed[r3v & (VD < 0)] = S_
    r2v = v_road & (VT == 2)
    allowed[r2v & (VO == 0)] = S_
    allowed[r2v & (VO == 1)] = N_
    r1v = v_road & (VT == 1)
    allowed[r1v & (VO < 2)] = S_
    allowed[r1v & (VO >= 2)] = N_
    # ring corner overrides (city_model.py:498-558)
    if True:
        hb0, hbl, vb0, vbl = hb[0], hb[-1], vb[0], vb[-1]
        cm = {  # (row, col) -> bit, rows relative to the horizontal ring band, cols to the vertical one
            "bl": {(0, 0): E_, (0, 1): E_, (1, 0): S_, (1, 1): N_},
            "br": {(0, 0): E_, (0, 1): N_, (1, 0): W_, (1, 1): N_},
            "tr": {(0, 0): S_, (0, 1): N_, (1, 0): W_, (1, 1): W_},
            "tl": {(0, 0): S_, (0, 1): E_, (1, 0): S_, (1, 1): W_},
        }
        for key, hband, vband in (("bl", hb0, vb0), ("br", hb0, vbl), ("tr", hbl, vbl), ("tl", hbl, vb0)):
            for (r, c), bit in cm[key].items():
                allowed[hband[0] + r, vband[0] + c] = bit
    # R1 lane shifts (city_model.py:1312-1366): allowed unless the neighbour in that direction is an intersection
    def nb(mask_arr, dy, dx):
        out = np.zeros_like(mask_arr)
        ys = slice(max(0, -dy), H - max(0, dy))
        yd = slice(max(0, dy), H - max(0, -dy))
        xs = slice(max(0, -dx), W - max(0, dx))
        xd = slice(max(0, dx), W - max(0, -dx))
        out[ys, xs] = mask_arr[yd, xd]
        return out
    inter_n, inter_s, inter_e, inter_w = nb(inter, 1, 0), nb(inter, -1, 0), nb(inter, 0, 1), nb(inter, 0, -1)
    allowed[r1h & (HO == 1) & ~inter_s] |= S_
    allowed[r1h & (HO == 0) & ~inter_n] |= N_
    allowed[r1h & (HO == 2) & ~inter_n] |= N_
    allowed[r1h & (HO == 3) & ~inter_s] |= S_
    allowed[r1v & (VO == 1) & ~inter_w] |= W_
    allowed[r1v & (VO == 0) & ~inter_e] |= E_
    allowed[r1v & (VO == 2) & ~inter_e] |= E_
    allowed[r1v & (VO == 3) & ~inter_w] |= W_
    # intersections: every direction whose neighbour is an intersection or a road cell flowing that way
    for bit, dy, dx in ((N_, 1, 0), (E_, 0, 1), (S_, -1, 0), (W_, 0, -1)):
        nb_inter = nb(inter, dy, dx)
        nb_flow = nb((allowed & bit) != 0, dy, dx) & nb(is_road, dy, dx)
        allowed[inter & (nb_inter | nb_flow)] |= bit

    road_type = np.zeros((H, W), np.int8)
    road_type[inter] = 1
    road_type[(h_road & (HT == 1)) | (v_road & (VT == 1))] = 1
    road_type[(h_road & (HT == 2)) | (v_road & (VT == 2))] = 2
    road_type[(h_road & hring[:, None]) | (v_road & vring[None, :])] = 1     # ring road counts as 1
    road_type[(h_road & (HT == 3)) | (v_road & (VT == 3))] = 3

    # ---- light groups: one per band crossing --------------------------------------------------
    g_lights, light_xy, light_ctrl = [], [], []
    g_ns, g_ew, g_icell, g_nsin, g_nsout, g_ewin, g_ewout = [], [], [], [], [], [], []
    hy_prev_end = {i: (hb[i - 1][1] if i > 0 else -1) for i in range(len(hb))}
    hy_next_start = {i: (hb[i + 1][0] if i + 1 < len(hb) else H) for i in range(len(hb))}
    vx_prev_end = {j: (vb[j - 1][1] if j > 0 else -1) for j in range(len(vb))}
    vx_next_start = {j: (vb[j + 1][0] if j + 1 < len(vb) else W) for j in range(len(vb))}
    R = traffic_light_range
    for i, (hy0, hy1, htype, hdir, hr) in enumerate(hb):
        for j, (vx0, vx1, vtype, vdir, vr) in enumerate(vb):
            if hr and vr:
                continue
            lights, ns_l, ew_l = [], [], []
            nsin, nsout, ewin, ewout = [], [], [], []

            def add_light(lx, ly, ctrl, lanes_back, axis):
                li = len(light_xy)
                light_xy.append((lx, ly))
                light_ctrl.append(ctrl)
                lights.append(li)
                (ns_l if axis == 0 else ew_l).append(li)
                for (cx, cy) in lanes_back:
                    bits = int(allowed[cy, cx])
                    if bits & (N_ | S_):
                        (nsin if cy < ly else nsout).append((cx, cy))
                    elif bits & (E_ | W_):
                        (ewin if cx < lx else ewout).append((cx, cy))

            # heading N (from the south): lanes of the vertical band that flow N
            xs_n = [x for x in range(vx0, vx1 + 1) if allowed[hy0 - 1, x] & N_ and v_road[hy0 - 1, x]] if hy0 - 1 >= 0 else []
            if xs_n:
                ylo = max(hy_prev_end[i] + 1, hy0 - 2 - R)
                back = [(x, y) for x in xs_n for y in range(hy0 - 2, ylo - 1, -1) if v_road[y, x]]
                add_light(vx1 + 1, hy0 - 1, [(x, hy0 - 1) for x in xs_n], back, 0)
            xs_s = [x for x in range(vx0, vx1 + 1) if allowed[hy1 + 1, x] & S_ and v_road[hy1 + 1, x]] if hy1 + 1 < H else []
            if xs_s:
                yhi = min(hy_next_start[i] - 1, hy1 + 2 + R)
                back = [(x, y) for x in xs_s for y in range(hy1 + 2, yhi + 1) if v_road[y, x]]
                add_light(vx0 - 1, hy1 + 1, [(x, hy1 + 1) for x in xs_s], back, 0)
            ys_e = [y for y in range(hy0, hy1 + 1) if allowed[y, vx0 - 1] & E_ and h_road[y, vx0 - 1]] if vx0 - 1 >= 0 else []
            if ys_e:
                xlo = max(vx_prev_end[j] + 1, vx0 - 2 - R)
                back = [(x, y) for y in ys_e for x in range(vx0 - 2, xlo - 1, -1) if h_road[y, x]]
                add_light(vx0 - 1, hy0 - 1, [(vx0 - 1, y) for y in ys_e], back, 1)
            ys_w = [y for y in range(hy0, hy1 + 1) if allowed[y, vx1 + 1] & W_ and h_road[y, vx1 + 1]] if vx1 + 1 < W else []
            if ys_w:
                xhi = min(vx_next_start[j] - 1, vx1 + 2 + R)
                back = [(x, y) for y in ys_w for x in range(vx1 + 2, xhi + 1) if h_road[y, x]]
                add_light(vx1 + 1, hy1 + 1, [(vx1 + 1, y) for y in ys_w], back, 1)
            if not lights:
                continue
            g_lights.append(lights)
            g_ns.append(ns_l)
            g_ew.append(ew_l)
            g_icell.append([(x, y) for y in range(hy0, hy1 + 1) for x in range(vx0, vx1 + 1)])
            g_nsin.append(nsin); g_nsout.append(nsout); g_ewin.append(ewin); g_ewout.append(ewout)

    def ragged_xy(rows):
        off = np.zeros(len(rows) + 1, np.int32)
        off[1:] = np.cumsum([len(r) for r in rows])
        flat = np.asarray([c for r in rows for c in r], dtype=np.int32).reshape(-1, 2)
        return off, flat

    def ragged_i(rows):
        off = np.zeros(len(rows) + 1, np.int32)
        off[1:] = np.cumsum([len(r) for r in rows])
        flat = np.asarray([c for r in rows for c in r], dtype=np.int32)
        return off, flat

    n_carved = 0
    if carve_subblock_roads:      # (after the light groups: their lane lists are those of the band roads)
        n_carved = _carve_subblock_roads(seed + 7919, allowed, is_road, road_type, inter, hb, vb, subblock_chance, min_subblock_spacing)
    out = dict(width=np.int32(W), height=np.int32(H), allowed_dirs_map=allowed, is_road_map=is_road.astype(np.int8),
               road_type_map=road_type * is_road.astype(np.int8), intersection_map=inter.astype(np.int8))
    G = len(g_lights)
    out["g_light_off"] = np.zeros(G + 1, np.int32)
    out["g_light_off"][1:] = np.cumsum([len(r) for r in g_lights])
    out["light_xy"] = np.asarray(light_xy, np.int32).reshape(-1, 2)
    out["light_ctrl_off"], out["light_ctrl_xy"] = ragged_xy(light_ctrl)
    out["g_ns_lights_off"], out["g_ns_lights"] = ragged_i(g_ns)
    out["g_ew_lights_off"], out["g_ew_lights"] = ragged_i(g_ew)
    out["g_icell_off"], out["g_icell_xy"] = ragged_xy(g_icell)
    out["g_ns_in_off"], out["g_ns_in_xy"] = ragged_xy(g_nsin)
    out["g_ns_out_off"], out["g_ns_out_xy"] = ragged_xy(g_nsout)
    out["g_ew_in_off"], out["g_ew_in_xy"] = ragged_xy(g_ewin)
    out["g_ew_out_off"], out["g_ew_out_xy"] = ragged_xy(g_ewout)
    out["g_neighbors"] = np.full((G, 4, 2), -1, np.int32)
    out["g_neighbors_ctor"] = np.full((G, 4, 2), -1, np.int32)
    # schedule like the reference's constructor: groups, (no city blocks), traffic generator clock
    out["schedule_kinds0"] = np.asarray([0] * G + [3], np.int8)
    out["carved_blocks"] = np.int32(n_carved)
    return out


def make_routes(tables: dict, n_vehicles: int, seed: int = 2, min_len: int = 200, max_len: int = 600,
                keep_heading: float = 0.85):
    """Distinct start cells on non-intersection road cells and random-walk routes along the allowed
    directions.  Returns (start_xy [V,2], goal_xy [V,2], path_off [V+1], path_dirs uint8 [sum len])
    with direction codes N0 E1 S2 W3 (ts_add_vehicles_dirs)."""
    rng = np.random.RandomState(seed)
    allowed = tables["allowed_dirs_map"]
    H, W = allowed.shape
    cand = np.flatnonzero((tables["is_road_map"].ravel() == 1) & (tables["intersection_map"].ravel() == 0)
                          & (allowed.ravel() != 0))
    V = min(int(n_vehicles), len(cand))
    start = rng.choice(cand, size=V, replace=False)
    want = rng.randint(min_len, max_len + 1, size=V)
    pos = start.copy()
    heading = np.full(V, -1, np.int64)
    alive = np.ones(V, bool)
    length = np.zeros(V, np.int64)
    L = int(want.max())
    dirs = np.zeros((V, L), np.uint8) if V * L <= 2_000_000_000 else None
    if dirs is None:
        raise MemoryError("route buffer too large; lower max_len")
    # choice tables: for each bitmask, the list of set directions
    nset = np.array([bin(b).count("1") for b in range(16)])
    opts = np.zeros((16, 4), np.int64)
    for b in range(16):
        ds = [k for k in range(4) if b & (1 << k)]
        for q in range(4):
            opts[b, q] = ds[q % len(ds)] if ds else 0
    aflat = allowed.ravel()
    drivable = (tables["is_road_map"].ravel() == 1) | (aflat != 0)
    step = np.array([W, 1, -W, -1])
    for t in range(L):
        bits = aflat[pos].astype(np.int64)
        # never reverse
        rev = np.where(heading >= 0, 1 << ((heading + 2) & 3), 0)
        bits_nr = bits & ~rev
        bits = np.where(bits_nr != 0, bits_nr, bits)
        ok = alive & (bits != 0) & (length < want)
        if not ok.any():
            break
        pick = opts[bits, rng.randint(0, 4, size=V)]
        keep = (heading >= 0) & ((bits >> np.maximum(heading, 0)) & 1).astype(bool) & (rng.rand(V) < keep_heading)
        d = np.where(keep, heading, pick)
        npos = pos + step[d]
        x, y = pos % W, pos // W
        nx, ny = x + DX[d], y + DY[d]
        inb = (nx >= 0) & (nx < W) & (ny >= 0) & (ny < H)
        ok &= inb
        # drivable = road-like or carrying arrows (the reference's ControlledRoad cells have arrows but is_road_map 0)
        ok &= drivable[np.where(inb, npos, 0)]
        dirs[ok, t] = d[ok]
        pos = np.where(ok, npos, pos)
        heading = np.where(ok, d, heading)
        length += ok
        alive &= ok
    keepv = (length > 0) & (pos != start)   # start == goal would despawn inside the decide phase
    start, pos, length, dirs = start[keepv], pos[keepv], length[keepv], dirs[keepv]
    V = len(start)
    path_off = np.zeros(V + 1, np.int64)
    path_off[1:] = np.cumsum(length)
    mask = np.arange(dirs.shape[1])[None, :] < length[:, None]
    flat = dirs[mask]
    start_xy = np.stack([start % W, start // W], axis=1).astype(np.int32)
    goal_xy = np.stack([pos % W, pos // W], axis=1).astype(np.int32)
    return start_xy, goal_xy, path_off, flat.astype(np.uint8)


def dirs_to_xy(start_xy, path_off, dirs):
    """Expand direction-coded routes to (x, y) cell lists (for the plain ts_add_vehicles entry)."""
    V = len(start_xy)
    total = int(path_off[-1])
    dx = DX[dirs].astype(np.int64)
    dy = DY[dirs].astype(np.int64)
    cx, cy = np.cumsum(dx), np.cumsum(dy)
    # subtract the running sum at each vehicle's start offset
    lens = np.diff(path_off)
    basex = np.concatenate([[0], cx])[path_off[:-1]]
    basey = np.concatenate([[0], cy])[path_off[:-1]]
    rep = np.repeat(np.arange(V), lens)
    x = start_xy[rep, 0] + cx - basex[rep]
    y = start_xy[rep, 1] + cy - basey[rep]
    return np.stack([x, y], axis=1).astype(np.int32).reshape(total, 2)
