// host_agents.h - agents whose step() runs on the host between device windows: rain manager and clouds, the traffic generator,
// city blocks and the service-vehicle start/finish logic.
// Part of the single translation unit engine.hip (included from there, in order).
#pragma once

namespace {

// a host-side agent (rain manager = id 0, rain clouds = ids 1..) gets a row in the device table of schedule slots
int host_agent_register(E* e, int slot) {
  const int hid = e->n_host_agents;
  if (hid + 1 > e->cap_hslot) {
    int nc = std::max(64, e->cap_hslot * 2);
    int rc = regrow(e, &e->d.hslot, (size_t)e->n_host_agents, (size_t)nc);
    if (rc) return rc;
    e->cap_hslot = nc;
  }
  HIPOK(hipMemcpy(e->d.hslot + hid, &slot, 4, hipMemcpyHostToDevice));
  e->n_host_agents++;
  return TS_OK;
}

// math.hypot of CPython 3.10 (Modules/mathmodule.c vector_norm); libm's hypot can differ in the last bit
double py_hypot(double a, double b) {
  double vec[2] = {std::fabs(a), std::fabs(b)};
  double max = vec[0] > vec[1] ? vec[0] : vec[1];
  if (max == 0.0) return 0.0;
  const double T27 = 134217729.0;
  double x, scale, oldcsum, csum = 1.0, frac1 = 0.0, frac2 = 0.0, frac3 = 0.0, t, hi, lo, h;
  int max_e;
  std::frexp(max, &max_e);
  scale = std::ldexp(1.0, -max_e);
  for (int i = 0; i < 2; i++) {
    x = vec[i]; x *= scale;
    t = x * T27; hi = t - (t - x); lo = x - hi;
    x = hi * hi; oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
    x = 2.0 * hi * lo; oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
    frac3 += lo * lo;
  }
  h = std::sqrt(csum - 1.0 + (frac1 + frac2 + frac3));
  x = h; t = x * T27; hi = t - (t - x); lo = x - hi;
  x = -hi * hi; oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
  x = -2.0 * hi * lo; oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
  x = -lo * lo; oldcsum = csum; csum += x; frac3 += (oldcsum - csum) + x;
  x = csum - 1.0 + (frac1 + frac2 + frac3);
  return (h + x / (2.0 * h)) / scale;
}

// RainManager.add_random_rain (rain.py:100-148) + RainAgent.__init__ (24-57) + schedule.add(rain)
int rain_add_random(E* e) {
  MTPipe& r = e->rng_global;
  const double w = e->W, h = e->H, off = e->P.rain_spawn_offset;
  const int edge = (int)r.randbelow(4);  // random.choice(['N', 'S', 'E', 'W'])
  double x0, y0, xt, yt;
  int corner;  // 0 NW, 1 NE, 2 SW, 3 SE
  if (edge == 0) { x0 = 0.0 + (w - 0.0) * r.random(); y0 = h - off; corner = r.randbelow(2) ? 3 : 2; }
  else if (edge == 1) { x0 = 0.0 + (w - 0.0) * r.random(); y0 = off; corner = r.randbelow(2) ? 1 : 0; }
  else if (edge == 2) { x0 = w - off; y0 = 0.0 + (h - 0.0) * r.random(); corner = r.randbelow(2) ? 2 : 0; }
  else { x0 = off; y0 = 0.0 + (h - 0.0) * r.random(); corner = r.randbelow(2) ? 3 : 1; }
  if (corner == 0) { xt = 0; yt = h; } else if (corner == 1) { xt = w; yt = h; } else if (corner == 2) { xt = 0; yt = 0; } else { xt = w; yt = 0; }
  double dx = xt - x0, dy = yt - y0;
  double length = py_hypot(dx, dy);
  if (length == 0.0) length = 1.0;
  dx /= length; dy /= length;
  ts_engine::Rain c;
  c.x = x0; c.y = y0;
  double l2 = py_hypot(dx, dy);
  if (l2 == 0.0) l2 = 1.0;
  c.dx = dx / l2; c.dy = dy / l2;
  c.radius = r.randint(e->P.rain_radius_min, e->P.rain_radius_max);
  // schedule.add(rain): a new entry at the end of the schedule (it does not step in the tick that created it)
  if ((long long)e->n_sched + 1 >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^24 agents");
  int rc = ensure_vehicle_capacity(e, e->cap_v, e->n_sched + 1);
  if (rc) return rc;
  const int hid = e->n_host_agents;
  rc = host_agent_register(e, e->n_sched);
  if (rc) return rc;
  const int8_t kind = K_RAIN;
  HIPOK(hipMemcpy(e->d.sched_kind + e->n_sched, &kind, 1, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(e->d.sched_ref + e->n_sched, &hid, 4, hipMemcpyHostToDevice));
  e->n_sched++;
  e->mixed_order = true;
  e->rains.push_back(hid);
  e->rains_all.resize((size_t)hid);   // ids are 1-based behind the manager
  e->rains_all[(size_t)hid - 1] = c;
  e->rain_counter++;
  return TS_OK;
}

// RainManager.step (rain.py:156-184); the discs it saw are what rain_map becomes
int rain_manager_step(E* e, RainDiscs& discs) {
  if (e->rain_cooldown_left > 0) e->rain_cooldown_left--;
  if ((int)e->rains.size() < e->P.rain_occurrences_max && e->rain_cooldown_left == 0 &&
      e->rng_global.random() < e->P.rain_spawn_chance) {
    int rc = rain_add_random(e);
    if (rc) return rc;
  }
  discs.n = 0;
  for (int hid : e->rains) {
    const auto& c = e->rains_all[(size_t)hid - 1];
    if (!c.stepped) continue;   // covered_cells is empty until the cloud's first step
    if (discs.n >= 16) return fail(e, TS_E_CAPACITY, "more than 16 rain clouds");
    discs.cx[discs.n] = c.cx; discs.cy[discs.n] = c.cy; discs.r[discs.n] = c.radius; discs.n++;
  }
  return TS_OK;
}

// RainAgent.step (rain.py:60-84).  Returns 1 if the cloud left the map (schedule.remove(self)).
int rain_agent_step(E* e, int hid) {
  auto& c = e->rains_all[(size_t)hid - 1];
  c.x += c.dx; c.y += c.dy;
  c.cx = (int)c.x; c.cy = (int)c.y;   // int(): truncation toward zero
  c.stepped = true;
  const int R = c.radius;
  if (c.x < -R || c.x > e->W + R || c.y < -R || c.y > e->H + R) {
    // on_rain_exit runs while the cloud is still in city_model.rains: `not rains` is never true there, so the
    // cooldown never starts (rain.py:150-154)
    for (size_t k = 0; k < e->rains.size(); k++) if (e->rains[k] == hid) { e->rains.erase(e->rains.begin() + k); break; }
    c.alive = false;
    return 1;
  }
  return 0;
}

// _generate_day (dynamic_traffic_generator.py:307-396): internal, service and through trips of one day
void generate_day(E* e, int day_idx) {
  auto& G = e->gen;
  MTPipe& r = e->rng_global;
  // compute_quotas (319-331): floors, then +1 for the largest fractional parts (stable, descending)
  auto quotas = [&](int total) {
    const int nz = G.T.n_zones;
    std::vector<double> fc(nz);
    std::vector<int> fl(nz), order(nz);
    long long sum = 0;
    for (int z = 0; z < nz; z++) {
      fc[z] = (double)total * G.T.zones[z].through_distribution;
      fl[z] = (int)std::floor(fc[z]); sum += fl[z]; order[z] = z;
    }
    std::stable_sort(order.begin(), order.end(),
                     [&](int a, int b) { return fc[a] - std::floor(fc[a]) > fc[b] - std::floor(fc[b]); });
    const long long rem = total - sum;
    for (long long i = 0; i < rem && i < nz; i++) fl[order[i]] += 1;
    return fl;
  };
  const std::vector<int> food_q = quotas(G.T.total_service_vehicles_food), waste_q = quotas(G.T.total_service_vehicles_waste);
  for (int zi = 0; zi < G.T.n_zones; zi++) {
    const TsTrafficZone& z = G.T.zones[zi];
    const double z0 = (double)((long long)day_idx * 86400 + (long long)z.start_hour * 3600 - G.T.start_offset_seconds);
    const double z1 = (double)((long long)day_idx * 86400 + (long long)z.end_hour * 3600 - G.T.start_offset_seconds);
    const double span = z1 - z0;
    for (int k = 0; k < z.n_internal; k++) {
      const long long cnt = (long long)std::nearbyint((double)G.T.internal_population_per_day * z.fraction[k]);
      if (cnt == 0) continue;
      std::vector<int> origins, dests;
      for (size_t b = 0; b < G.blk_type.size(); b++) {
        if (G.blk_type[b] == z.origin_type[k]) origins.push_back((int)b);
        if (G.blk_type[b] == z.dest_type[k]) dests.push_back((int)b);
      }
      if (origins.empty() || dests.empty()) continue;
      for (long long q = 0; q < cnt; q++) {
        const double t = z0 + r.random() * span;
        const int ob = origins[r.randbelow((uint32_t)origins.size())];
        const int db = dests[r.randbelow((uint32_t)dests.size())];
        const int oc = G.blk_entr[ob][r.randbelow((uint32_t)G.blk_entr[ob].size())];
        const int dc = G.blk_entr[db][r.randbelow((uint32_t)G.blk_entr[db].size())];
        G.pending.push_back(ts_engine::Trip{oc, dc, t, TS_POP_INTERNAL, day_idx});
      }
    }
    // service vehicles, uniform per zone (362-376): one entrance draw per trip
    const int Nf = food_q[zi], Nw = waste_q[zi];
    for (int j = 1; j <= Nf; j++) {
      const double t = z0 + (double)((long long)j * (long long)span) / (double)(Nf + 1);
      const int sc = G.hw_in[r.randbelow((uint32_t)G.hw_in.size())];
      G.pending.push_back(ts_engine::Trip{sc, -1, t, TS_TRIP_SERVICE_FOOD, day_idx});
    }
    for (int j = 1; j <= Nw; j++) {
      const double t = z0 + (double)((long long)j * (long long)span) / (double)(Nw + 1);
      const int sc = G.hw_in[r.randbelow((uint32_t)G.hw_in.size())];
      G.pending.push_back(ts_engine::Trip{sc, -1, t, TS_TRIP_SERVICE_WASTE, day_idx});
    }
    long long thr = (long long)std::nearbyint((double)G.T.passing_population_per_day * z.through_distribution);
    thr -= Nf + Nw;   // SERVICE_VEHICLES_COUNT_AS_THROUGH defaults to True (90, 381-382)
    for (long long q = 0; q < thr; q++) {
      const double t = z0 + r.random() * span;
      const int ent = G.hw_in[r.randbelow((uint32_t)G.hw_in.size())];
      const int ex = G.hw_out[r.randbelow((uint32_t)G.hw_out.size())];
      G.pending.push_back(ts_engine::Trip{ent, ex, t, TS_POP_THROUGH, day_idx});
    }
  }
}


// ------------------------------ city blocks + service vehicles (host state) -------------------------------
// CityBlock.step (city_block.py:110-150)
void block_step(E* e, int bi) {
  if (bi >= (int)e->blocks.size()) return;
  auto& b = e->blocks[bi];
  const TsTrafficTables& T = e->gen.T;
  if (b.needs_food) {
    if (T.gradual_city_block_resources) {
      b.food_rem += b.food_rate;
      if (b.food_rem >= 1.0) { const double whole = std::trunc(b.food_rem); b.food = std::max(b.food - whole, 0.0); b.food_rem -= whole; }
    } else if (++b.ticks_since_food >= T.food_consumption_ticks) {
      b.food = std::max(b.food - (double)b.cells, 0.0); b.ticks_since_food = 0;
    }
  }
  if (b.produces_waste) {
    if (T.gradual_city_block_resources) {
      b.waste_rem += b.waste_rate;
      if (b.waste_rem >= 1.0) { const double whole = std::trunc(b.waste_rem); b.waste = std::min(b.waste + whole, b.max_waste); b.waste_rem -= whole; }
    } else if (++b.ticks_since_waste >= T.waste_production_ticks) {
      b.waste = std::min(b.waste + (double)b.cells, b.max_waste); b.ticks_since_waste = 0;
    }
  }
}

// CityBlock.get_service_road_cell step 4 (city_block.py:192-202): first ranked cell without a parked vehicle
int service_road_cell(E* e, int bi) {
  for (int c : e->blocks[bi].service_cells) {
    auto it = e->parked_cells.find(c);
    if (it == e->parked_cells.end() || it->second <= 0) return c;
  }
  return -1;
}

int svc_find(E* e, int vid) {
  for (size_t k = 0; k < e->svc.size(); k++) if (e->svc[k].vid == vid) return (int)k;
  return -1;
}

// ServiceVehicleAgent._start_service, host part (vehicle_service.py:85-104); `pos` = the cell it parked on
void svc_start(E* e, ts_engine::SvcVeh& v) {
  if (v.phase != 0 || v.block < 0) {   // a vehicle that merely parks (base on_target_reached with remove_on_arrival False)
    e->parked_cells[v.target]++;
    return;
  }
  e->parked_cells[v.target]++;
  v.pos = v.target;
  auto& b = e->blocks[v.block];
  if (v.type == TS_TRIP_SERVICE_FOOD) {
    const double need = b.max_food - b.food;
    const double amt = std::min(v.load, need);
    b.food = std::min(b.food + amt, b.max_food);
    v.load -= amt;
  } else {
    const double surplus = b.waste;
    const double cap = v.max_load - v.load;
    const double amt = std::min(cap, surplus);
    b.waste = std::max(b.waste - amt, 0.0);
    v.load += amt;
  }
  v.ticks = e->gen.T.service_load_time;
  v.phase = 1;
}

// ServiceVehicleAgent._finish_service (vehicle_service.py:106-141) at the vehicle's place in the shuffled order:
// every lower-ranked agent has stepped on the device, every higher-ranked one has not
int svc_finish(E* e, ts_engine::SvcVeh& v) {
  auto& G = e->gen;
  { auto it = e->parked_cells.find(v.pos); if (it != e->parked_cells.end() && --it->second <= 0) e->parked_cells.erase(it); }
  const bool more = v.type == TS_TRIP_SERVICE_FOOD ? v.load > 0 : v.load < v.max_load;
  int target = -1, to_block = 0;
  if (more) {
    int nb = -1;   // get_block_most_in_need_of_food / _waste_pickup (city_model.py:2078-2087): stable sort, first element
    for (size_t b = 0; b < e->blocks.size(); b++) {
      const auto& B = e->blocks[b];
      if (v.type == TS_TRIP_SERVICE_FOOD) { if (B.needs_food && (nb < 0 || B.food < e->blocks[nb].food)) nb = (int)b; }
      else { if (B.produces_waste && (nb < 0 || B.waste > e->blocks[nb].waste)) nb = (int)b; }
    }
    if (nb >= 0) {
      v.block = nb;
      target = service_road_cell(e, nb);
      if (target < 0) {
        e->fatal = TS_E_UNSUPPORTED;
        return fail(e, TS_E_UNSUPPORTED, "service vehicle: every service road cell of the next block holds a parked vehicle (the reference raises)");
      }
      to_block = 1;
    }
  }
  if (!to_block) {
    int best_d = 0;
    for (int c : G.hw_out) {   // min(exits, key=manhattan): first minimum
      const int dd = std::abs(c % e->W - v.pos % e->W) + std::abs(c / e->W - v.pos / e->W);
      if (target < 0 || dd < best_d) { target = c; best_d = dd; }
    }
    if (target < 0) { e->fatal = TS_E_UNSUPPORTED; return fail(e, TS_E_UNSUPPORTED, "service vehicle without highway exits (the reference raises)"); }
  }
  hipLaunchKernelGGL(k_svc_finish, dim3(1), dim3(64), 0, e->stream, e->d, e->P, v.vid, target, to_block);
  v.target = target;
  v.phase = to_block ? 0 : 2;
  return plan_vehicle(e, v.vid, v.pos, target);
}

// _spawn for service trips (dynamic_traffic_generator.py:419-430) + ServiceVehicleAgent.__init__ (vehicle_service.py:19-41)
// `id` = index into the fleet's id pool, -1 for a vehicle the UI created with an id of its own
int spawn_service_at(E* e, int origin, int kind, int id) {
  auto& G = e->gen;
  const bool food = kind == TS_TRIP_SERVICE_FOOD;
  // _find_initial_target (62-83): `attempt` is never advanced, so only valid_blocks[0] is ever tried
  int blk = -1;
  for (size_t b = 0; b < e->blocks.size(); b++)
    if (food ? e->blocks[b].needs_food : e->blocks[b].produces_waste) { blk = (int)b; break; }
  int target, phase;
  if (blk >= 0) {
    target = service_road_cell(e, blk);
    if (target < 0) {
      e->fatal = TS_E_UNSUPPORTED;
      return fail(e, TS_E_UNSUPPORTED, "service vehicle: no free service road cell at its first block (the reference loops forever)");
    }
    phase = 0;
  } else {
    if (G.hw_out.empty()) { e->fatal = TS_E_UNSUPPORTED; return fail(e, TS_E_UNSUPPORTED, "service vehicle without highway exits (IndexError in the reference)"); }
    target = G.hw_out[0];
    phase = 2;
  }
  if (id >= 0) {
    char& live = e->sv_live[(size_t)(food ? 0 : G.T.total_service_vehicles_food) + id];
    if (live) {   // BaseScheduler.add raises on a unique_id that is already scheduled (Mesa <= 2.1)
      e->fatal = TS_E_UNSUPPORTED;
      return fail(e, TS_E_UNSUPPORTED, "service vehicle id drawn while a vehicle with that id is still live (the scheduler raises in the reference)");
    }
    live = 1;
  }
  if ((long long)e->n_sched + 1 >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^24 agents");
  if (!e->d.arr) {   // first service vehicle of this engine: the record buffer the kernels report arrivals in
    e->d.arr_cap = 1 << 16;
    HIPOK(dalloc(e, &e->d.arr, (size_t)e->d.arr_cap * 3));
  }
  int rc = add_vehicle_planned(e, origin, target, TS_POP_THROUGH);
  if (rc) return rc;
  const int vid = e->n_vehicles_total - 1;
  hipLaunchKernelGGL(k_flags_or, dim3(1), dim3(64), 0, e->stream, e->d, vid, (int)(VF_SVC | VF_KEEP | (phase == 0 ? VF_TOBLOCK : 0)));
  ts_engine::SvcVeh v;
  v.vid = vid; v.type = kind; v.id = id; v.block = blk;
  v.max_load = food ? G.T.service_max_load_food : G.T.service_max_load_waste;
  v.load = food ? v.max_load : 0.0;
  v.phase = phase; v.ticks = 0; v.pos = origin; v.target = target;
  e->svc.push_back(v);
  if (food) e->C.live_service_food++; else e->C.live_service_waste++;
  return TS_OK;
}
int spawn_service(E* e, const ts_engine::Trip& t) {
  auto& G = e->gen;
  const bool food = t.kind == TS_TRIP_SERVICE_FOOD;
  if (food) e->C.created_service_food++; else e->C.created_service_waste++;
  const int pool = food ? G.T.total_service_vehicles_food : G.T.total_service_vehicles_waste;
  const int id = (int)e->rng_global.randbelow((uint32_t)pool);   // vid = random.choice(pool)
  return spawn_service_at(e, t.origin, t.kind, id);
}

// DynamicTrafficAgent._update_cached_stats (525-648).  The generator takes it inside its own step: the move phase has run
// every lower-ranked agent on the device and none of the higher-ranked ones, the tick's spawns are placed - exactly the
// `city.schedule.agents` the reference sums over.  The sums are a device reduction (k_live_stats), the counters come down
// with them, the daily figures are host state (pending_trips_today 244-248, next_service_eta 278-288).
int update_cached_stats(E* e) {
  auto& G = e->gen;
  Dev& d = e->d;
  if (!e->d_live_stats) HIPOK(dalloc(e, &e->d_live_stats, 8));
  HIPOK(hipMemsetAsync(e->d_live_stats, 0, 8 * sizeof(double), e->stream));
  if (e->n_active > 0)
    hipLaunchKernelGGL(k_live_stats, dim3(nblk(e->n_active)), dim3(BLK), 0, e->stream, d, e->n_active, e->C.elapsed, e->d_live_stats);
  double raw[8];
  HIPOK(hipMemcpyAsync(raw, e->d_live_stats, sizeof(raw), hipMemcpyDeviceToHost, e->stream));
  int rc = sync_counters(e);
  if (rc) return rc;
  TsCachedStats& c = G.cs;
  memset(&c, 0, sizeof(c));
  c.valid = 1;
  c.update_step = e->C.step_count;
  long long li[8];
  memcpy(li, raw, sizeof(li));
  c.dur_live[0] = raw[0]; c.dur_live[1] = raw[1];
  c.dist_live[0] = li[2]; c.dist_live[1] = li[3]; c.n_live[0] = li[4]; c.n_live[1] = li[5];
  c.stuck_ticks_sum = li[6]; c.stuck_ticks_max = li[7];
  c.stuck = e->C.stuck; c.collisions = e->C.collisions; c.malfunctions = e->C.malfunctions; c.parked = e->C.parked;
  c.overtaking = e->C.overtaking; c.in_stuck_detour = e->C.in_stuck_detour;
  c.live_internal = e->C.live_internal; c.live_through = e->C.live_through;
  c.live_service_food = e->C.live_service_food; c.live_service_waste = e->C.live_service_waste;
  c.count_completed[0] = e->C.count_completed_internal; c.count_completed[1] = e->C.count_completed_through;
  c.total_distance[0] = e->C.total_distance_internal; c.total_distance[1] = e->C.total_distance_through;
  c.total_duration[0] = e->C.total_duration_internal; c.total_duration[1] = e->C.total_duration_through;
  const int kinds[4] = {TS_POP_INTERNAL, TS_POP_THROUGH, TS_TRIP_SERVICE_FOOD, TS_TRIP_SERVICE_WASTE};
  const int64_t created[4] = {e->C.created_internal, e->C.created_through, e->C.created_service_food, e->C.created_service_waste};
  for (int q = 0; q < 4; q++) {
    long long pending_today = 0;
    double eta = std::numeric_limits<double>::quiet_NaN();
    for (const auto& t : G.pending) {
      if (t.day != G.current_day || t.kind != kinds[q]) continue;
      pending_today++;
      if (t.depart > e->C.elapsed) { const double dt = t.depart - e->C.elapsed; if (!(eta <= dt)) eta = dt; }
    }
    c.created[q] = created[q];
    c.daily_total[q] = q == 0 ? G.T.internal_population_per_day : q == 1 ? G.T.passing_population_per_day : created[q] + pending_today;
    c.eta[q] = eta;
  }
  c.errored[0] = e->C.errored_internal; c.errored[1] = e->C.errored_through;
  double sum = 0.0;
  for (long long x : G.daily_difference_history) sum += (double)x;
  c.avg_daily_difference = G.daily_difference_history.empty() ? 0.0 : sum / (double)G.daily_difference_history.size();
  return TS_OK;
}

// DynamicTrafficAgent.step (153-194) and _spawn (398-416), executed at the agent's place in the shuffled order:
// every lower-ranked agent has stepped on the device, every higher-ranked one has not yet.
int generator_step(E* e) {
  auto& G = e->gen;
  const double prev = e->C.elapsed;
  e->C.elapsed += e->P.time_per_step_seconds;
  const double total_secs = G.T.start_offset_seconds + e->C.elapsed;
  const int new_day = (int)std::floor(total_secs / 86400.0);
  if (new_day > G.current_day) {
    int rc = sync_counters(e);      // (the completed counts live on the device)
    if (rc) return rc;
    const long long done = e->C.count_completed_internal + e->C.count_completed_through;
    G.daily_difference_history.push_back((done - G.completed_at_day_start) - (e->C.created_internal + e->C.created_through));   // finished - spawned (165-167)
    G.completed_at_day_start = done;
    for (int dd = G.current_day + 1; dd <= new_day; dd++) generate_day(e, dd);
    G.current_day = new_day;
    e->C.created_internal = 0; e->C.created_through = 0;
    e->C.created_service_food = 0; e->C.created_service_waste = 0;
  }
  std::vector<ts_engine::Trip> keep, spawn;
  for (const auto& t : G.pending) (prev < t.depart && t.depart <= e->C.elapsed ? spawn : keep).push_back(t);
  G.pending.swap(keep);
  for (const auto& t : spawn) {
    if (t.kind == TS_TRIP_SERVICE_FOOD || t.kind == TS_TRIP_SERVICE_WASTE) {
      int rc = spawn_service(e, t);
      if (rc) return rc;
      continue;
    }
    if (t.kind == TS_POP_INTERNAL) e->C.created_internal++; else e->C.created_through++;
    (void)e->rng_global.randint(0, 9999);  // the id suffix of "V_{depart:06d}_{randint(0, 9999):04d}"
    if ((long long)e->n_sched + 1 >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^24 agents");
    int rc = add_vehicle_planned(e, t.origin, t.dest, t.kind);
    if (rc) return rc;
  }
  // _update_cached_stats every STATISTICS_UPDATE_INTERVAL ticks, inside this step (188-194)
  const int interval = G.T.statistics_update_interval > 0 ? G.T.statistics_update_interval : 20;
  if (++G.ticks_since_stats >= interval) { int rc = update_cached_stats(e); if (rc) return rc; G.ticks_since_stats = 0; }
  return TS_OK;
}

}  // namespace
