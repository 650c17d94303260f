// kernels.h - the per-tick device kernels other than the A* family (astar.h): decide front, move rounds, compaction, events.
// Part of the single translation unit engine.hip (included from there, in order).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// decide, part 1 (pure): which draws of the global MT19937 stream does each vehicle consume?
// step_decide prologue, vehicle_base.py:616-643 with _tick_stranded 552-565, _check_malfunction
// 608-610, _check_sideswipe_collision 567-605, _is_at_stopped_cell 121-127, _compute_speed 94-107.
// ---------------------------------------------------------------------------------------------
__global__ void k_decide_pre(Dev d, TsParams P, int start, int n_active) {
  int i = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  int vid = d.active[i];
  uint8_t F = 0;
  int cand = -1;
  if (vid >= 0) {
    uint16_t f = d.flags[vid];
    bool sb = (f & (VF_COLL | VF_MALF)) != 0;
    bool still = sb && (d.stranded_left[vid] - 1 > 0);
    // strandedness before / after this vehicle's own step_decide, for k_decide_main's blocker checks
    d.st_before[vid] = sb;
    d.st_after[vid] = still || !P.malfunction_active;  // `not ACTIVE or ...`: everyone malfunctions (609)
    if (!still && P.malfunction_active) {
      F |= F_DRAW_MALF;
      const int W = d.W, H = d.H;
      int pos = d.pos[vid];
      int dir = d.dir[vid];
      if (P.sideswipe_active && dir >= 0) {
        int x, y;
        cell_xy(d, pos, x, y);
        const int opposite = (dir + 2) & 3;
        for (int k = 0; k < 2 && cand < 0; k++) {
          int ld = k == 0 ? ((dir + 3) & 3) : ((dir + 1) & 3);  // left, then right
          int nx = x + (ld == 1) - (ld == 3), ny = y + (ld == 0) - (ld == 2);
          if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
          for (int ag = d.cell[ny * W + nx].veh; ag >= 0; ag = d.next_in_cell[ag]) {
            uint16_t af = d.flags[ag];
            bool earlier = !d.seq && d.active_idx[ag] < i;
            bool ag_sb = (af & (VF_COLL | VF_MALF)) != 0;
            bool ag_str = earlier ? (d.ev[ag] ? true : ((ag_sb && d.stranded_left[ag] - 1 > 0) || !P.malfunction_active)) : ag_sb;
            bool cs_pos = earlier ? (!ag_str && d.cell[d.pos[ag]].stop != 1) : (d.cur_speed[ag] > 0);
            if (!cs_pos || (af & (VF_STUCK | VF_PARKED)) || ag_str) continue;
            if (earlier && (af & VF_KEEP) && d.pos[ag] == d.target[ag]) continue;   // it parked inside its own step_decide
            if (d.dir[ag] != opposite) continue;
            cand = ag;
            break;
          }
        }
        if (cand >= 0) F |= F_DRAW_SWIPE;
      }
      if (d.cell[pos].stop != 1 && d.base_speed[vid] == 0) F |= F_DRAW_SPEED;
    }
  }
  d.F[i] = F;
  d.cand[i] = cand;
}

// _set_malfunction / _set_collision (vehicle_base.py:534-550) for the (rare) events the host scan finds.
// ev: 1 = stranded at its own decide point (early exit there); 2 = hit by a later vehicle after deciding;
// 3 = hit before its own turn (it will find itself stranded).  ev_idx = decide-order index of the event.
__global__ void k_apply_event(Dev d, TsParams P, int vid, int is_collision, int partner, int my_idx) {
  if (threadIdx.x || blockIdx.x) return;
  {
    // the vehicle reached its draws, so a stranding it still carried from earlier ticks expired in this very
    // step_decide (_tick_stranded, vehicle_base.py:556-564): do that bookkeeping before the new stranding
    uint16_t f0 = d.flags[vid];
    if (f0 & VF_COLL) atomicAdd((unsigned long long*)&d.cnt->collisions, (unsigned long long)-1LL);
    if (f0 & VF_MALF) atomicAdd((unsigned long long*)&d.cnt->malfunctions, (unsigned long long)-1LL);
    d.flags[vid] = f0 & ~(VF_COLL | VF_MALF);
  }
  if (!is_collision) {
    d.flags[vid] = (d.flags[vid] | VF_MALF) & ~VF_COLL;
    d.stranded_left[vid] = P.malfunction_duration;
    d.base_speed[vid] = 0; d.cur_speed[vid] = 0;
    d.ev[vid] = 1; d.ev_idx[vid] = my_idx;
    atomicAdd((unsigned long long*)&d.cnt->malfunctions, 1ULL);
  } else {
    d.flags[vid] = (d.flags[vid] | VF_COLL) & ~VF_MALF;
    d.stranded_left[vid] = P.sideswipe_duration;
    d.base_speed[vid] = 0; d.cur_speed[vid] = 0;
    d.ev[vid] = 1; d.ev_idx[vid] = my_idx;
    {
      // an earlier partner that was a valid candidate had any old stranding expire in its own step_decide of
      // this tick; its stored flags still show it because k_decide_main has not run yet
      uint16_t pf = d.flags[partner];
      if (pf & VF_COLL) atomicAdd((unsigned long long*)&d.cnt->collisions, (unsigned long long)-1LL);
      if (pf & VF_MALF) atomicAdd((unsigned long long*)&d.cnt->malfunctions, (unsigned long long)-1LL);
    }
    d.flags[partner] = (d.flags[partner] | VF_COLL) & ~VF_MALF;
    d.stranded_left[partner] = P.sideswipe_duration;
    if (!d.seq && d.active_idx[partner] < my_idx) {
      // already decided this tick: k_decide_main still needs its pre-collision base_speed to reproduce that
      // decision, and zeroes base/current speed itself afterwards (ev == 2)
      d.ev[partner] = 2;
    } else {
      d.ev[partner] = 3;
      d.base_speed[partner] = 0; d.cur_speed[partner] = 0;
    }
    d.ev_idx[partner] = my_idx;
    atomicAdd((unsigned long long*)&d.cnt->collisions, 2ULL);
  }
}

// ---------------------------------------------------------------------------------------------
// move phase
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void claim(const Dev& d, int cell, int plane, uint32_t key) { atomicMin(&d.cell[cell].claim[plane], key); }

__device__ __forceinline__ bool group_reads_out(const TsParams& P) {
  return P.light_algorithm == TS_LIGHTS_PRESSURE_CONTROL || P.light_algorithm == TS_LIGHTS_NEIGHBOR_PRESSURE_CONTROL;
}
__device__ __forceinline__ bool group_reads_neighbors(const TsParams& P) {
  return P.light_algorithm == TS_LIGHTS_NEIGHBOR_PRESSURE_CONTROL || P.light_algorithm == TS_LIGHTS_NEIGHBOR_GREEN_WAVE;
}

// Every unresolved agent announces the cells it will read/write: cw_* = min rank of writers,
// cr_* = min rank of readers.  Keys carry an epoch prefix that DEcreases every round, so atomicMin
// makes stale entries of earlier rounds lose and nothing has to be cleared.
// `list` == nullptr: every schedule slot (first round); otherwise the slots left unresolved by the previous round.
__global__ void k_move_claim(Dev d, TsParams P, int n_sched, uint32_t prefix, const int32_t* list, const int* list_n,
                             uint32_t rank_limit, int group_cells_elsewhere) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int s;
  if (list) { if (t >= *list_n) return; s = list[t]; } else { s = t; if (s >= n_sched) return; }
  if (d.resolved[s] || d.rank[s] >= rank_limit) return;   // rank_limit: agents behind a host-side agent wait for it
  const int8_t kind = d.sched_kind[s];
  const uint32_t key = (prefix << RANK_BITS) | d.rank[s];
  if (kind == K_VEHICLE) {
    const int vid = d.sched_ref[s];
    const uint16_t f = d.flags[vid];
    const int pos = d.pos[vid];
    if (f & VF_SERVICING) return;   // ServiceVehicleAgent.step only counts down (vehicle_service.py:43-49)
    if (f & VF_EARLY) {
      if (d.G > 0 && P.light_algorithm != TS_LIGHTS_DISABLED) claim(d, pos, 3, key);  // tick_stuck reads stop[pos]
      if (pos == d.target[vid] || P.stuck_despawn_enabled) claim(d, pos, 0, key);   // it may leave: occupancy[pos] = 0
    } else {
      const int m = d.max_steps[vid];
      claim(d, pos, 0, key);
      const uint32_t off = d.path_off[vid];
      const int pcur = d.path_cur[vid];
      int c = pos;
      for (int k = 0; k < m; k++) {
        c = step_cell(c, path_dir(d.pool, off, pcur + k), d.W);
        claim(d, c, 0, key);
        if (d.G > 0) claim(d, c, 3, key);
      }
    }
  } else if (kind == TS_AGENT_LIGHT_GROUP && P.light_algorithm != TS_LIGHTS_DISABLED) {
    const int g = d.sched_ref[s];
    if (!group_cells_elsewhere) {   // (the first round of a phase claims the cells in k_move_claim_groups)
      for (int k = d.g_icell_off[g]; k < d.g_icell_off[g + 1]; k++) claim(d, d.g_icell[k], 1, key);
      for (int k = d.g_nsin_off[g]; k < d.g_nsin_off[g + 1]; k++) claim(d, d.g_nsin[k], 1, key);
      for (int k = d.g_ewin_off[g]; k < d.g_ewin_off[g + 1]; k++) claim(d, d.g_ewin[k], 1, key);
      if (group_reads_out(P)) {
        for (int k = d.g_nsout_off[g]; k < d.g_nsout_off[g + 1]; k++) claim(d, d.g_nsout[k], 1, key);
        for (int k = d.g_ewout_off[g]; k < d.g_ewout_off[g + 1]; k++) claim(d, d.g_ewout[k], 1, key);
      }
      for (int l = d.g_light_off[g]; l < d.g_light_off[g + 1]; l++) {
        claim(d, d.light_cell[l], 2, key);
        for (int k = d.light_ctrl_off[l]; k < d.light_ctrl_off[l + 1]; k++) claim(d, d.light_ctrl[k], 2, key);
      }
    }
    if (group_reads_neighbors(P)) {
      for (int k = 0; k < 4; k++) {
        int n1 = d.g_nb[(g * 4 + k) * 2 + 1], n2 = d.g_nb_ctor[(g * 4 + k) * 2 + 1];
        if (n1 >= 0) atomicMin(&d.gclaim_r[n1], key);
        if (n2 >= 0) atomicMin(&d.gclaim_r[n2], key);
      }
    }
  }
}

// the cell claims of every unresolved light group below the rank limit, one thread per (cell, group, plane) pair
__global__ void k_move_claim_groups(Dev d, uint32_t prefix, uint32_t rank_limit) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d.gc_n) return;
  const int s = d.g_slot[d.gc_group[j]];
  const uint32_t rk = d.rank[s];
  if (d.resolved[s] || rk >= rank_limit) return;
  atomicMin(&d.cell[d.gc_cell[j]].claim[d.gc_plane[j]], (prefix << RANK_BITS) | rk);
}

__device__ __forceinline__ void cell_unlink(const Dev& d, int cell, int vid) {
  int h = d.cell[cell].veh;
  if (h == vid) { d.cell[cell].veh = d.next_in_cell[vid]; return; }
  while (h >= 0 && d.next_in_cell[h] != vid) h = d.next_in_cell[h];
  if (h >= 0) d.next_in_cell[h] = d.next_in_cell[vid];
}
__device__ __forceinline__ void cell_append(const Dev& d, int cell, int vid) {
  d.next_in_cell[vid] = -1;
  int h = d.cell[cell].veh;
  if (h < 0) { d.cell[cell].veh = vid; return; }
  while (d.next_in_cell[h] >= 0) h = d.next_in_cell[h];
  d.next_in_cell[h] = vid;
}

// CityModel.remove_vehicle (city_model.py:1920-1941): off the maps, the cell list, the schedule and the decide order
// (pop_arg: the population_type argument of CityModel.remove_vehicle; the vehicle's own despawn passes its own, -1 here)
__device__ __forceinline__ void remove_vehicle_dev(const Dev& d, int vid, int s, int pos, uint16_t& f, int key, int pop_arg = -1) {
  set_occ(d, pos, 0); d.cell[pos].stuck = 0;
  cell_unlink(d, pos, vid);
  f &= ~VF_ALIVE;
  d.sched_kind[s] = K_DEAD;
  d.active[d.active_idx[vid]] = -1;
  int pop = pop_arg >= 0 ? pop_arg : d.pop[vid];
  if (pop == TS_POP_INTERNAL) atomicAdd((unsigned long long*)&d.cnt->live_internal, (unsigned long long)-1LL);
  else if (pop == TS_POP_THROUGH) atomicAdd((unsigned long long*)&d.cnt->live_through, (unsigned long long)-1LL);
  atomicAdd(&d.cnt->deaths, 1);
  if (f & VF_SVC) svc_record(d, key, vid, AR_DESPAWN);
}

// on_target_reached (vehicle_base.py:755-775) -> _despawn -> CityModel.remove_vehicle (city_model.py:1920-1941)
__device__ void on_target_reached_dev(const Dev& d, const TsParams& P, int vid, int s, int pos, uint16_t& f,
                                      double elapsed_now, int key) {
  if (f & VF_TOBLOCK) {   // ServiceVehicleAgent.on_target_reached -> _start_service (vehicle_service.py:54-60, 85-104):
    // park and start the load timer here; the load / block bookkeeping is host state (AR_START record)
    if (!(f & VF_PARKED)) { f |= VF_PARKED; atomicAdd((unsigned long long*)&d.cnt->parked, 1ULL); }
    f = (f & ~VF_TOBLOCK) | VF_SERVICING;
    svc_record(d, key, vid, AR_START);
    return;
  }
  if (P.enable_traffic) {
    double duration = elapsed_now - d.depart[vid];
    int pop = d.pop[vid];
    if (pop == TS_POP_INTERNAL) {
      atomicAdd(&d.cnt->dur_internal, duration);
      atomicAdd((unsigned long long*)&d.cnt->dist_internal, (unsigned long long)d.steps[vid]);
      atomicAdd((unsigned long long*)&d.cnt->completed_internal, 1ULL);
    } else if (pop == TS_POP_THROUGH) {
      atomicAdd(&d.cnt->dur_through, duration);
      atomicAdd((unsigned long long*)&d.cnt->dist_through, (unsigned long long)d.steps[vid]);
      atomicAdd((unsigned long long*)&d.cnt->completed_through, 1ULL);
    }
  }
  if (!(f & VF_KEEP)) {
    remove_vehicle_dev(d, vid, s, pos, f, key);
  } else if (!(f & VF_PARKED)) {
    f |= VF_PARKED;
    atomicAdd((unsigned long long*)&d.cnt->parked, 1ULL);
    if (f & VF_SVC) svc_record(d, key, vid, AR_START);   // the host keeps the set of cells with a parked vehicle
  }
}

// VehicleAgent.step with PATHFINDING_BATCHING (vehicle_base.py:666-685): _execute_movement 733-753,
// _move_to 521-532 + CityModel.move_vehicle (city_model.py:1945-1963), tick_stuck 687-693.
// `cells` / `recs`: the vehicle's cell and the next max_steps path cells with their records as k_move_resolve
// loaded them for the claim test (nothing can have changed them since: that is what "safe" means); pass nullptr
// to read them here.
constexpr int MOVE_MAX = 8;
__device__ __forceinline__ void vehicle_step_dev(const Dev& d, const TsParams& P, int vid, int s, double elapsed_now, int key,
                                                 const bool pre, const int (&cells)[MOVE_MAX + 1], const uint4 (&dyn)[MOVE_MAX + 1]) {
  uint16_t f = d.flags[vid];
  if (f & VF_SERVICING) return;   // the countdown and _finish_service are host state
  int pos = d.pos[vid];
  if (!(f & VF_EARLY)) {
    const int m = d.max_steps[vid];
    const uint32_t off = d.path_off[vid];
    const int pcur = d.path_cur[vid];
    const int plen = d.path_len[vid] - pcur;
    const bool was_stuck = (f & VF_STUCK) != 0;
    int c = pos, moved = 0, lastdir = -1;
    if (pre) {
      bool go = true;
#pragma unroll
      for (int k = 0; k < MOVE_MAX; k++) {
        if (go && k < m && k < plen) {
          const int nc = cells[k + 1];
          const uint32_t dw = dyn[k + 1].y;   // occ | stop << 8 | stuck << 16 | stat << 24
          int occ = (int8_t)(dw & 0xFF);
          const int stop = (int8_t)((dw >> 8) & 0xFF);
          // the record was read before this vehicle started to move: a cell it has itself left in the meantime
          // (a route may loop back through it) is free now - leaving clears the byte whoever else stands there
#pragma unroll
          for (int j = 0; j < MOVE_MAX; j++) if (j < k && cells[j] == nc) occ = 0;
          if (occ == 1 || (stop == 1 && k != m - 1)) go = false;
          else {
            set_occ(d, c, 0); set_occ(d, nc, 1);
            d.cell[c].stuck = 0; d.cell[nc].stuck = (k == 0 && was_stuck) ? 1 : 0;
            lastdir = nc == c + d.W ? 0 : nc == c + 1 ? 1 : nc == c - d.W ? 2 : 3;
            c = nc; moved++;
          }
        }
      }
    } else {
      for (int k = 0; k < m; k++) {
        if (k >= plen) break;
        const int nd = path_dir(d.pool, off, pcur + k);
        const int nc = step_cell(c, nd, d.W);
        const Cell ncell = d.cell[nc];
        if (ncell.occ == 1) break;
        if (ncell.stop == 1 && k != m - 1) break;
        set_occ(d, c, 0); set_occ(d, nc, 1);
        d.cell[c].stuck = 0; d.cell[nc].stuck = (k == 0 && was_stuck) ? 1 : 0;
        c = nc; moved++; lastdir = nd;
      }
    }
    if (moved) {
      cell_unlink(d, pos, vid);
      cell_append(d, c, vid);
      pos = c;
      d.pos[vid] = c;
      d.dir[vid] = (int8_t)lastdir;
      if (d.stuck_ticks[vid] > 0) {
        if (was_stuck) { atomicAdd((unsigned long long*)&d.cnt->stuck, (unsigned long long)-1LL); f &= ~VF_STUCK; }
        d.stuck_ticks[vid] = 0;
      }
      d.steps[vid] += moved;
      d.path_cur[vid] = pcur + moved;
    }
    f |= VF_HASPREV;
  } else {
    f &= ~VF_EARLY;
    const int stop_here = pre ? (int)(int8_t)((dyn[0].y >> 8) & 0xFF) : (int)d.cell[pos].stop;
    if ((f & VF_HASPREV) && stop_here != 1) {
      int st = d.stuck_ticks[vid] + 1;
      d.stuck_ticks[vid] = st;
      if (st > P.stuck_recompute_threshold && !(f & VF_STUCK)) {
        atomicAdd((unsigned long long*)&d.cnt->stuck, 1ULL);
        f |= VF_STUCK;
      }
    }
  }
  if (pos == d.target[vid]) on_target_reached_dev(d, P, vid, s, pos, f, elapsed_now, key);
  // _despawn_check (vehicle_base.py:695-706); a vehicle that has just been removed is not checked again
  if (P.stuck_despawn_enabled && (f & VF_ALIVE)) {
    const int thr = st_inter(d.cell[pos].stat) == 1 ? P.stuck_despawn_threshold_intersection : P.stuck_despawn_threshold;
    if (d.stuck_ticks[vid] >= thr) {
      if (f & VF_STUCK) { atomicAdd((unsigned long long*)&d.cnt->stuck, (unsigned long long)-1LL); f &= ~VF_STUCK; }
      if (d.pop[vid] == TS_POP_INTERNAL) atomicAdd((unsigned long long*)&d.cnt->errored_internal, 1ULL);
      else atomicAdd((unsigned long long*)&d.cnt->errored_through, 1ULL);
      remove_vehicle_dev(d, vid, s, pos, f, key);
    }
  }
  if (f & VF_SVCNEW) f = (f & ~VF_SVCNEW) | VF_SERVICING;   // (the step() in which _start_service ran inside step_decide is over)
  d.flags[vid] = f;
}

__device__ __forceinline__ void light_set(const Dev& d, int l, int8_t v) {  // cell.py:241-251
  set_stop(d, d.light_cell[l], v);
  for (int k = d.light_ctrl_off[l]; k < d.light_ctrl_off[l + 1]; k++) set_stop(d, d.light_ctrl[k], v);
}
__device__ __forceinline__ int queue_sum(const Dev& d, const int32_t* off, const int32_t* cells, int g) {
  int q = 0;  // compute_approach_queue (numba_utilities.py:65-72)
  for (int k = off[g]; k < off[g + 1]; k++) q += d.occ[cells[k]];
  return q;
}
__device__ __forceinline__ void apply_phase(int& cur, int& pend, int phase) {  // intersection_light_group.py:386-393
  if (phase == cur || phase == pend) return;
  pend = phase;
}

// IntersectionLightGroup.step (intersection_light_group.py:396-423) and _execute_phase_change (348-384)
__device__ void group_step_dev(const Dev& d, const TsParams& P, int g) {
  int cur = d.gs_cur[g], pend = d.gs_pend[g];
  if (pend < 0) {
    switch (P.light_algorithm) {
      case TS_LIGHTS_FIXED_TIME: {
        int t = d.gs_fttimer[g] + 1, ph = d.gs_ftphase[g];
        if (t == 1) apply_phase(cur, pend, ph);
        if (t >= P.green_duration) { ph = 1 - ph; t = 0; }
        d.gs_fttimer[g] = t; d.gs_ftphase[g] = ph;
        break;
      }
      case TS_LIGHTS_QUEUE_ACTUATED: {
        int qt = d.gs_qtimer[g] + 1, gap = d.gs_gap[g], last = d.gs_last[g];
        int ns_q = queue_sum(d, d.g_nsin_off, d.g_nsin, g), ew_q = queue_sum(d, d.g_ewin_off, d.g_ewin, g);
        int cq = cur == 0 ? ns_q : ew_q, oq = cur == 0 ? ew_q : ns_q;
        if (qt == 1) { last = cq; gap = 0; }
        if (cq > last) { last = cq; gap = 0; } else gap += 1;
        if (qt >= P.qa_min_green && (gap >= P.qa_gap || qt >= P.qa_max_green || (oq > cq && cq == 0))) {
          apply_phase(cur, pend, 1 - cur);
          qt = 0;
        }
        d.gs_qtimer[g] = qt; d.gs_gap[g] = gap; d.gs_last[g] = last;
        break;
      }
      case TS_LIGHTS_PRESSURE_CONTROL:
      case TS_LIGHTS_NEIGHBOR_PRESSURE_CONTROL: {
        int ns_p = queue_sum(d, d.g_nsin_off, d.g_nsin, g) - queue_sum(d, d.g_nsout_off, d.g_nsout, g);
        int ew_p = queue_sum(d, d.g_ewin_off, d.g_ewin, g) - queue_sum(d, d.g_ewout_off, d.g_ewout, g);
        if (P.light_algorithm == TS_LIGHTS_NEIGHBOR_PRESSURE_CONTROL) {
          const int32_t* nb = d.gs_repop[g] ? d.g_nb : d.g_nb_ctor;
          for (int k = 0; k < 4; k++) {
            int nd = nb[(g * 4 + k) * 2], n = nb[(g * 4 + k) * 2 + 1];
            if (nd < 0 || n < 0) continue;
            if (nd == 0 || nd == 2) ns_p -= d.gs_nsp[n]; else ew_p -= d.gs_ewp[n];
          }
        }
        d.gs_nsp[g] = ns_p; d.gs_ewp[g] = ew_p;
        apply_phase(cur, pend, ns_p > ew_p ? 0 : 1);
        break;
      }
      case TS_LIGHTS_NEIGHBOR_GREEN_WAVE: {
        int ns_q = queue_sum(d, d.g_nsin_off, d.g_nsin, g), ew_q = queue_sum(d, d.g_ewin_off, d.g_ewin, g);
        bool fns = false, few = false;
        const int32_t* nb = d.gs_repop[g] ? d.g_nb : d.g_nb_ctor;
        for (int k = 0; k < 4; k++) {
          int nd = nb[(g * 4 + k) * 2], n = nb[(g * 4 + k) * 2 + 1];
          if (nd < 0 || n < 0) continue;
          if ((nd == 0 || nd == 2) && d.gs_cur[n] == 0) fns = true;
          if ((nd == 1 || nd == 3) && d.gs_cur[n] == 1) few = true;
        }
        if (fns && !few) apply_phase(cur, pend, 0);
        else if (few && !fns) apply_phase(cur, pend, 1);
        else apply_phase(cur, pend, ns_q > ew_q ? 0 : 1);
        break;
      }
      default: break;
    }
  }
  if (pend >= 0) {
    bool done = false;
    if (P.transition_duration_enabled && d.gs_trans[g] > 0) {
      d.gs_trans[g] -= 1;
      for (int l = d.g_light_off[g]; l < d.g_light_off[g + 1]; l++) light_set(d, l, 1);
      done = true;
    }
    if (!done && P.transition_clearance_enabled) {
      bool occupied = false;  // is_intersection_occupied (285-291)
      for (int k = d.g_icell_off[g]; k < d.g_icell_off[g + 1]; k++) if (d.occ[d.g_icell[k]]) { occupied = true; break; }
      if (occupied) {
        for (int l = d.g_light_off[g]; l < d.g_light_off[g + 1]; l++) light_set(d, l, 1);
        done = true;
      }
    }
    if (!done) {
      if (P.transition_duration_enabled && d.gs_clear[g] > 0) d.gs_trans[g] = P.all_red_duration;
      d.gs_repop[g] = 1;  // get_opposite_traffic_lights() re-ran populate_links() (303-307)
      const int32_t *go_off = pend == 0 ? d.g_ns_off : d.g_ew_off, *go = pend == 0 ? d.g_ns : d.g_ew;
      const int32_t *st_off = pend == 0 ? d.g_ew_off : d.g_ns_off, *st = pend == 0 ? d.g_ew : d.g_ns;
      for (int k = go_off[g]; k < go_off[g + 1]; k++) light_set(d, go[k], 0);
      for (int k = st_off[g]; k < st_off[g + 1]; k++) light_set(d, st[k], 1);
      cur = pend; pend = -1;
    }
  }
  d.gs_cur[g] = cur; d.gs_pend[g] = pend;
}

// An agent steps in this round iff no unresolved agent of lower rank claims a cell it reads or writes.
__global__ void k_move_resolve(Dev d, TsParams P, int n_sched, uint32_t prefix, uint32_t rank_clock, double elapsed0,
                               const int32_t* list, const int* list_n, int32_t* out_list, int* out_n, uint32_t rank_limit) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int s;
  if (list) { if (t >= *list_n) return; s = list[t]; } else { s = t; if (s >= n_sched) return; }
  if (d.resolved[s] || d.rank[s] >= rank_limit) return;
  const int8_t kind = d.sched_kind[s];
  const uint32_t r = d.rank[s];
  bool safe = true;
  if (kind == K_VEHICLE) {
    const int vid = d.sched_ref[s];
    const uint16_t f = d.flags[vid];
    const int pos = d.pos[vid];
    const bool lights = d.G > 0 && P.light_algorithm != TS_LIGHTS_DISABLED;
    // The vehicle's own cell and the cells it may enter: decode them, then load their records together (the
    // claim words in .x-.w of the first 16 bytes, the dynamic dword right behind) - one memory round trip for
    // both the claim test and the movement.
    int cells[MOVE_MAX + 1];
    uint4 claims[MOVE_MAX + 1], dyn[MOVE_MAX + 1];
    const int m = (f & (VF_EARLY | VF_SERVICING)) ? 0 : (int)d.max_steps[vid];
    const bool fast = m <= MOVE_MAX;
    if (fast) {
      const uint32_t off = d.path_off[vid];
      const int pcur = d.path_cur[vid];
      // the next 8 steps are at most 16 bits of the direction string: two pool words, decoded in registers
      uint64_t bits = 0;
      if (m > 0) {
        const uint32_t wi = (uint32_t)pcur >> 4, nwords = ((uint32_t)d.path_len[vid] + 15u) >> 4;
        bits = d.pool[off + wi];
        if (wi + 1 < nwords) bits |= (uint64_t)d.pool[off + wi + 1] << 32;
        bits >>= (pcur & 15) * 2;
      }
      int c = pos;
      cells[0] = pos;
#pragma unroll
      for (int k = 0; k < MOVE_MAX; k++) {
        if (k < m) c = step_cell(c, (int)((bits >> (2 * k)) & 3), d.W);
        cells[k + 1] = c;
      }
#pragma unroll
      for (int k = 0; k <= MOVE_MAX; k++) {
        if (k <= m) {
          const uint4* rp = reinterpret_cast<const uint4*>(&d.cell[cells[k]]);
          claims[k] = rp[0];
          dyn[k] = rp[1];   // .x = veh, .y = occ | stop << 8 | stuck << 16 | stat << 24
        }
      }
    }
    if (f & VF_SERVICING) {
      // nothing on the maps is read or written
    } else if (f & VF_EARLY) {
      const uint4 cl = fast ? claims[0] : *reinterpret_cast<const uint4*>(&d.cell[pos]);
      if (lights && claim_rank(cl.z, prefix) < r) safe = false;
      if ((pos == d.target[vid] || P.stuck_despawn_enabled) && (claim_rank(cl.x, prefix) < r || claim_rank(cl.y, prefix) < r)) safe = false;
    } else if (fast) {
#pragma unroll
      for (int k = 0; k <= MOVE_MAX; k++) {
        if (k <= m && (claim_rank(claims[k].x, prefix) < r || claim_rank(claims[k].y, prefix) < r)) safe = false;
        if (k > 0 && k <= m && lights && claim_rank(claims[k].z, prefix) < r) safe = false;
      }
    } else {
      const int mm = d.max_steps[vid];
      if (claim_rank(d.cell[pos].claim[0], prefix) < r || claim_rank(d.cell[pos].claim[1], prefix) < r) safe = false;
      const uint32_t off = d.path_off[vid];
      const int pcur = d.path_cur[vid];
      int c = pos;
      for (int k = 0; k < mm && safe; k++) {
        c = step_cell(c, path_dir(d.pool, off, pcur + k), d.W);
        if (claim_rank(d.cell[c].claim[0], prefix) < r || claim_rank(d.cell[c].claim[1], prefix) < r) safe = false;
        if (lights && claim_rank(d.cell[c].claim[2], prefix) < r) safe = false;
      }
    }
    if (!safe) { out_list[atomicAdd(out_n, 1)] = s; return; }
    vehicle_step_dev(d, P, vid, s, elapsed0 + (r > rank_clock ? (double)P.time_per_step_seconds : 0.0), (int)r, fast, cells, dyn);
  } else if (kind == TS_AGENT_LIGHT_GROUP && P.light_algorithm != TS_LIGHTS_DISABLED) {
    const int g = d.sched_ref[s];
    for (int k = d.g_icell_off[g]; k < d.g_icell_off[g + 1] && safe; k++)
      if (claim_rank(d.cell[d.g_icell[k]].claim[0], prefix) < r) safe = false;
    for (int k = d.g_nsin_off[g]; k < d.g_nsin_off[g + 1] && safe; k++)
      if (claim_rank(d.cell[d.g_nsin[k]].claim[0], prefix) < r) safe = false;
    for (int k = d.g_ewin_off[g]; k < d.g_ewin_off[g + 1] && safe; k++)
      if (claim_rank(d.cell[d.g_ewin[k]].claim[0], prefix) < r) safe = false;
    if (group_reads_out(P)) {
      for (int k = d.g_nsout_off[g]; k < d.g_nsout_off[g + 1] && safe; k++)
        if (claim_rank(d.cell[d.g_nsout[k]].claim[0], prefix) < r) safe = false;
      for (int k = d.g_ewout_off[g]; k < d.g_ewout_off[g + 1] && safe; k++)
        if (claim_rank(d.cell[d.g_ewout[k]].claim[0], prefix) < r) safe = false;
    }
    for (int l = d.g_light_off[g]; l < d.g_light_off[g + 1] && safe; l++) {
      int lc = d.light_cell[l];
      if (claim_rank(d.cell[lc].claim[2], prefix) < r || claim_rank(d.cell[lc].claim[3], prefix) < r) safe = false;
      for (int k = d.light_ctrl_off[l]; k < d.light_ctrl_off[l + 1] && safe; k++) {
        int cc = d.light_ctrl[k];
        if (claim_rank(d.cell[cc].claim[2], prefix) < r || claim_rank(d.cell[cc].claim[3], prefix) < r) safe = false;
      }
    }
    if (safe && group_reads_neighbors(P)) {
      for (int k = 0; k < 4 && safe; k++) {
        for (int w = 0; w < 2; w++) {
          int n = (w ? d.g_nb_ctor : d.g_nb)[(g * 4 + k) * 2 + 1];
          if (n < 0) continue;
          int ns = d.g_slot[n];
          if (!d.resolved[ns] && d.rank[ns] < r) safe = false;  // the neighbour writes its state first
        }
      }
      if (claim_rank(d.gclaim_r[g], prefix) < r) safe = false;  // a lower-ranked group still has to read mine
    }
    if (!safe) { out_list[atomicAdd(out_n, 1)] = s; return; }
    group_step_dev(d, P, g);
  }
  d.resolved[s] = 1;
  atomicAdd(&d.cnt->resolved, 1);
}

// ---------------------------------------------------------------------------------------------
// stable compaction of the two ordered lists after despawns (ballot/popc within a wave, block scan in LDS)
// ---------------------------------------------------------------------------------------------
constexpr int CITEMS = 4;  // elements per thread
__global__ void k_compact_count(const int32_t* active, const int8_t* kind, int n, int which, int* block_counts) {
  __shared__ int wsum[BLK / 64];
  int base = blockIdx.x * BLK * CITEMS;
  int c = 0;
  for (int j = 0; j < CITEMS; j++) {
    int i = base + j * BLK + threadIdx.x;
    if (i < n) c += which == 0 ? (active[i] >= 0) : (kind[i] != K_DEAD);
  }
  for (int o = 32; o; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < BLK / 64; w++) t += wsum[w]; block_counts[blockIdx.x] = t; }
}
__global__ void k_scan_blocks(int* block_counts, int nb, int* total) {  // single block, exclusive scan in place
  __shared__ int carry;
  __shared__ int buf[1024];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < nb ? block_counts[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) block_counts[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
// Rows are visited j-major inside a block, so the block's output order is preserved by scanning each
// j-slab in turn (slab = BLK consecutive elements).
__global__ void k_compact_scatter(Dev d, int n, int which, const int* block_off, int32_t* out_a, int8_t* out_kind,
                                  int32_t* out_ref) {
  __shared__ int wcnt[BLK / 64];
  __shared__ int running;
  if (threadIdx.x == 0) running = block_off[blockIdx.x];
  __syncthreads();
  int base = blockIdx.x * BLK * CITEMS;
  for (int j = 0; j < CITEMS; j++) {
    int i = base + j * BLK + threadIdx.x;
    bool keep = false;
    if (i < n) keep = which == 0 ? (d.active[i] >= 0) : (d.sched_kind[i] != K_DEAD);
    unsigned long long m = __ballot(keep);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int before = __popcll(m & ((1ULL << lane) - 1));
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < w; q++) woff += wcnt[q];
    int dst = running + woff + before;
    if (keep) {
      if (which == 0) {
        int vid = d.active[i];
        out_a[dst] = vid;
        d.active_idx[vid] = dst;
      } else {
        int8_t k = d.sched_kind[i];
        int ref = d.sched_ref[i];
        out_kind[dst] = k; out_ref[dst] = ref;
        // which table remembers this agent's slot (a plain select: the if / else-if chain with the two-kind arm
        // was miscompiled for gfx950 by ROCm 7.2's hipcc at -O3, leaving the K_RAIN lane's base pointer undefined)
        int32_t* tab = nullptr;
        switch (k) {
          case K_VEHICLE: tab = d.sched_slot; break;
          case TS_AGENT_LIGHT_GROUP: tab = d.g_slot; break;
          case TS_AGENT_RAIN_MANAGER: tab = d.hslot; break;
          case K_RAIN: tab = d.hslot; break;
          case TS_AGENT_CITY_BLOCK: tab = d.bslot; break;
          default: break;
        }
        if (tab) tab[ref] = dst;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < BLK / 64; q++) t += wcnt[q]; running += t; }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// spawn, read-back and density kernels
// ---------------------------------------------------------------------------------------------
struct SpawnArgs {
  const int32_t *start, *goal, *pop, *plen;
  const uint32_t* poff;
  const uint8_t* serial;  // 1 = start cell shared inside the batch -> placed by the serial kernel
};
__global__ void k_spawn(Dev d, TsParams P, SpawnArgs a, int n, int vid0, int active0, int sched0, double elapsed,
                        int* overflow, int* n_overflow, int amap_live) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int vid = vid0 + i;
  int pos = a.start[i];
  d.pos[vid] = pos; d.target[vid] = a.goal[i];
  d.path_off[vid] = a.poff[i]; d.path_len[vid] = a.plen[i]; d.path_cur[vid] = 0;
  d.stuck_ticks[vid] = 0; d.cooldown[vid] = P.pathfinding_cooldown; d.stranded_left[vid] = 0; d.steps[vid] = 0;
  d.over_dur[vid] = -1; d.det_dur[vid] = -1; d.next_in_cell[vid] = -1;
  d.base_speed[vid] = 0; d.cur_speed[vid] = 0; d.max_steps[vid] = 0; d.dir[vid] = -1; d.pop[vid] = (int8_t)a.pop[i];
  d.flags[vid] = VF_ALIVE; d.depart[vid] = P.enable_traffic ? elapsed : 0.0;
  d.ev[vid] = 0; d.st_before[vid] = 0; d.st_after[vid] = 0; d.tier_hint[vid] = 0; d.chg[vid] = 0;
  if (amap_live) {   // keep the A* snapshot current: the spawn-time planners of this batch run on it
    int x, y;
    cell_xy(d, pos, x, y);
    const uint32_t t = tix(d, x, y);
    atomicOr(&d.amap[t], 0x100ull);
  }
  for (int k = 0; k < 4; k++) { d.ax_len[k][vid] = 0; d.ax_off[k][vid] = 0; d.ax_start[k][vid] = pos; }
  d.active[active0 + i] = vid; d.active_idx[vid] = active0 + i;
  d.sched_kind[sched0 + i] = K_VEHICLE; d.sched_ref[sched0 + i] = vid; d.sched_slot[vid] = sched0 + i;
  set_occ(d, pos, 1); d.cell[pos].stuck = 0;  // place_vehicle (city_model.py:1897-1918)
  if (a.serial[i] || atomicCAS(&d.cell[pos].veh, -1, vid) != -1) overflow[atomicAdd(n_overflow, 1)] = vid;
}
__global__ void k_spawn_serial(Dev d, int* overflow, int n_overflow) {  // cells holding several vehicles: list order = spawn order
  if (threadIdx.x || blockIdx.x) return;
  for (int a = 1; a < n_overflow; a++) {  // insertion sort by vehicle id (tiny)
    int v = overflow[a], b = a - 1;
    while (b >= 0 && overflow[b] > v) { overflow[b + 1] = overflow[b]; b--; }
    overflow[b + 1] = v;
  }
  for (int a = 0; a < n_overflow; a++) cell_append(d, d.pos[overflow[a]], overflow[a]);
}

__global__ void k_rows(Dev d, int n_active, const uint32_t* crc_table, int32_t* rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  int vid = d.active[i];
  int32_t* r = rows + (size_t)i * TS_V_NFIELDS;
  int pos = d.pos[vid];
  r[TS_V_SPAWN_IDX] = vid; r[TS_V_X] = pos % d.W; r[TS_V_Y] = pos / d.W;
  r[TS_V_BASE_SPEED] = d.base_speed[vid]; r[TS_V_CURRENT_SPEED] = d.cur_speed[vid]; r[TS_V_MAX_STEPS] = d.max_steps[vid];
  r[TS_V_DIRECTION] = d.dir[vid]; r[TS_V_STUCK_TICKS] = d.stuck_ticks[vid]; r[TS_V_COOLDOWN] = d.cooldown[vid];
  r[TS_V_FLAGS] = d.flags[vid] & 0x1FF; r[TS_V_STRANDED_LEFT] = d.stranded_left[vid];
  r[TS_V_STEPS_TRAVELED] = d.steps[vid];
  int pcur = d.path_cur[vid], plen = d.path_len[vid] - pcur;
  r[TS_V_PATH_LEN] = plen;
  uint32_t crc = 0;
  if (plen > 0) {
    crc = 0xFFFFFFFFu;
    int c = pos;
    uint32_t off = d.path_off[vid];
    for (int k = 0; k < plen; k++) {
      c = step_cell(c, path_dir(d.pool, off, pcur + k), d.W);
      int32_t xy[2] = {c % d.W, c / d.W};
      const uint8_t* p = (const uint8_t*)xy;
      for (int b = 0; b < 8; b++) crc = crc_table[(crc ^ p[b]) & 0xff] ^ (crc >> 8);
    }
    crc ^= 0xFFFFFFFFu;
  }
  r[TS_V_PATH_CRC] = (int32_t)crc;
  r[TS_V_OVERTAKE_DUR] = d.over_dur[vid]; r[TS_V_DETOUR_DUR] = d.det_dur[vid];
}
__global__ void k_meta_rows(Dev d, int n_active, int32_t* rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  const int vid = d.active[i];
  int32_t* r = rows + (size_t)i * TS_M_NFIELDS;
  const int tgt = d.target[vid];
  const uint16_t f = d.flags[vid];
  r[TS_M_SPAWN_IDX] = vid; r[TS_M_POPULATION] = d.pop[vid]; r[TS_M_TARGET_X] = tgt % d.W; r[TS_M_TARGET_Y] = tgt / d.W;
  r[TS_M_VEHICLE_TYPE] = 0;   // the fleet (food / waste) is host state: filled in by ts_download_vehicle_meta
  r[TS_M_SERVICE_PHASE] = !(f & VF_SVC) ? -1 : (f & VF_TOBLOCK) ? 0 : (f & VF_SERVICING) ? 1 : 2;
}

__global__ void k_path_cells(Dev d, int vid, int32_t* xy) {
  if (threadIdx.x || blockIdx.x) return;
  int pcur = d.path_cur[vid], plen = d.path_len[vid] - pcur, c = d.pos[vid];
  uint32_t off = d.path_off[vid];
  for (int k = 0; k < plen; k++) {
    c = step_cell(c, path_dir(d.pool, off, pcur + k), d.W);
    xy[2 * k] = c % d.W; xy[2 * k + 1] = c / d.W;
  }
}
__global__ void k_group_rows(Dev d, int32_t* rows) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= d.G) return;
  int32_t* r = rows + (size_t)g * TS_G_NFIELDS;
  r[TS_G_CURRENT_PHASE] = d.gs_cur[g]; r[TS_G_PENDING_PHASE] = d.gs_pend[g]; r[TS_G_QUEUE_TIMER] = d.gs_qtimer[g];
  r[TS_G_GAP_TIMER] = d.gs_gap[g]; r[TS_G_LAST_ARRIVAL] = d.gs_last[g]; r[TS_G_FIXED_TIME_TIMER] = d.gs_fttimer[g];
  r[TS_G_FT_PHASE] = d.gs_ftphase[g]; r[TS_G_NS_PRESSURE] = d.gs_nsp[g]; r[TS_G_EW_PRESSURE] = d.gs_ewp[g];
}

// _update_density_map (city_model.py:1764-1778): scipy.ndimage.uniform_filter on float32 = two 1-D passes
// with double accumulators and a float32 intermediate; `* 441` and the division in float32.
__global__ void k_density_pass0(const int8_t* occ, const int8_t* road, int W, int H, int r, float* t_occ, float* t_road) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  int x = i % W, y = i / W;
  int y0 = max(0, y - r), y1 = min(H - 1, y + r);
  int c0 = 0, c1 = 0;
  for (int yy = y0; yy <= y1; yy++) { c0 += occ[yy * W + x]; c1 += road[yy * W + x]; }
  const double size = (double)(2 * r + 1);
  t_occ[i] = (float)((double)c0 / size);
  t_road[i] = (float)((double)c1 / size);
}
__global__ void k_density_pass1(const float* t_occ, const float* t_road, int W, int H, int r, float* density) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  int x = i % W, y = i / W;
  int x0 = max(0, x - r), x1 = min(W - 1, x + r);
  double s0 = 0.0, s1 = 0.0;
  for (int xx = x0; xx <= x1; xx++) { s0 += (double)t_occ[y * W + xx]; s1 += (double)t_road[y * W + xx]; }
  const double size = (double)(2 * r + 1);
  const float area = (float)((2 * r + 1) * (2 * r + 1));
  float v0 = (float)(s0 / size) * area, v1 = (float)(s1 / size) * area;
  density[i] = v1 > 0.f ? __fdiv_rn(v0, v1) : 0.f;
}

// rain_map = union of the clouds' discs as RainManager.step saw them (rain.py:156-184): a cell is covered by a
// cloud when (x - cx)^2 + (y - cy)^2 <= r^2 for the cloud's integer centre at its last step
struct RainDiscs { int n; int cx[16], cy[16], r[16]; };
// RainManager.step (rain.py:156-184): the cells that rained at its previous step are cleared, the cells under the
// clouds it sees now are set; everything else (a host may have written the map) stays as it is
__global__ void k_rain_map(int8_t* rain, int W, int H, RainDiscs prev, RainDiscs D) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  const int x = i % W, y = i / W;
  bool was = false, is = false;
  for (int k = 0; k < prev.n; k++) {
    const int dx = x - prev.cx[k], dy = y - prev.cy[k];
    if (dx * dx + dy * dy <= prev.r[k] * prev.r[k]) was = true;
  }
  for (int k = 0; k < D.n; k++) {
    const int dx = x - D.cx[k], dy = y - D.cy[k];
    if (dx * dx + dy * dy <= D.r[k] * D.r[k]) is = true;
  }
  if (is) rain[i] = 1;
  else if (was) rain[i] = 0;
}

// rank[slot] = position of the slot in the shuffled key order
// cell records <-> byte planes
__global__ void k_cells_init(Cell* cell, int n, const uint8_t* allowed, const int8_t* is_road, const int8_t* road_type,
                             const int8_t* inter, const int32_t* node_of_cell) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  Cell x;
  x.claim[0] = x.claim[1] = x.claim[2] = x.claim[3] = 0xFFFFFFFFu;
  x.veh = -1; x.occ = 0; x.stop = 0; x.stuck = 0;
  x.stat = (uint8_t)((allowed[c] & 15) | ((is_road[c] == 1) << 4) | ((inter[c] == 1) << 5) | ((road_type[c] & 3) << 6));
  x.pad_[0] = (uint32_t)node_of_cell[c];   // search-node number (A* snapshot, Dev::amap), 0xFFFFFFFF = none
  x.pad_[1] = 0;
  cell[c] = x;
}
// the A* snapshot of the maps (Dev::amap) from the cell records
// pen_mode: what the penalty bits carry for searches in half units - 0 nothing (penalties are not multiples of 0.5: those
// searches read the density map themselves), 1 the constant vehicle penalty, 2 the density-dependent one of this cell
// (VEHICLE_DYNAMIC_PENALTIES; density = the tick-start snapshot, so a rebuild inside the tick gives the same bits)
__global__ void k_amap_build(Dev d, int pen_mode, double veh_pen, double dyn_scale) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.N) return;
  int x, y;
  cell_xy(d, c, x, y);
  const uint2 w = *reinterpret_cast<const uint2*>(&d.cell[c].occ);   // .x = occ | stop << 8 | stuck << 16 | stat << 24, .y = node number
  uint32_t flags = (w.x >> 24) | (((int8_t)(w.x & 0xFF) == 1) ? 0x100u : 0u) | (((int8_t)((w.x >> 8) & 0xFF) == 1) ? 0x200u : 0u);
  const int pen2 = pen_mode == 0 ? 0 : pen_mode == 1 ? (int)(veh_pen * 2.0) : 2 * (int)occ_penalty_dyn(veh_pen, dyn_scale, d.density[c]);
  flags |= (uint32_t)pen2 << AMAP_PEN_SHIFT;
  d.amap[tix(d, x, y)] = (unsigned long long)flags | ((unsigned long long)w.y << 32);
}
__global__ void k_claims_reset(Cell* cell, int n) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  cell[c].claim[0] = cell[c].claim[1] = cell[c].claim[2] = cell[c].claim[3] = 0xFFFFFFFFu;
}
__global__ void k_plane_to_cells(Cell* cell, int n, const int8_t* plane, int which) {   // which: 0 occ, 1 stop
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  if (which == 0) cell[c].occ = plane[c]; else cell[c].stop = plane[c];
}
__global__ void k_cells_to_plane(const Cell* cell, int n, int8_t* plane) {   // stuck_map
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  plane[c] = cell[c].stuck;
}

// on_target_reached inside step_decide for the vehicles that stay (AR_DECIDE records): the flag changes other
// deciders must not see half-way are applied once the decide kernels are done
__global__ void k_decide_arrive(Dev d, int first, int n_rec) {
  int k = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_rec || d.arr[3 * k + 2] != AR_DECIDE) return;
  const int vid = d.arr[3 * k + 1];
  uint16_t f = d.flags[vid];
  if (!(f & VF_PARKED)) { f |= VF_PARKED; atomicAdd((unsigned long long*)&d.cnt->parked, 1ULL); }
  if (f & VF_TOBLOCK) f = (f & ~VF_TOBLOCK) | (d.seq ? VF_SVCNEW : VF_SERVICING);
  d.flags[vid] = f;
}
// Vehicles that stand on their target at the start of a tick and are not kept on arrival (a trip that ends where it
// starts): they despawn inside step_decide (vehicle_base.py:657-661).  list[0] = count, list[1..] = decide indices.
__global__ void k_find_standing(Dev d, int n_active, int32_t* list, int cap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  const int vid = d.active[i];
  if (vid < 0) return;
  const uint16_t f = d.flags[vid];
  if ((f & VF_KEEP) || !(f & VF_ALIVE) || d.pos[vid] != d.target[vid]) return;
  const int k = atomicAdd(&list[0], 1);
  if (k < cap) list[1 + k] = i;
}
// ... and the despawn itself, run once every vehicle before number i_arrived of the decide order is through with its
// step_decide (searches included): on_target_reached -> _despawn -> CityModel.remove_vehicle.  The removal happens while
// run_parallel_decide iterates active_vehicle_agents (city_model.py:1817-1827), so the list iterator skips the vehicle
// that follows (i_skipped, -1 = none): it does not decide in this tick at all.  Whoever looks at it later in this decide
// phase must see its stored state - it is marked as "not yet decided" (the compaction that follows every despawn
// restores active_idx).
__global__ void k_decide_despawn(Dev d, TsParams P, int i_arrived, int i_skipped) {
  if (threadIdx.x || blockIdx.x) return;
  const int vid = d.active[i_arrived];
  uint16_t f = d.flags[vid];
  on_target_reached_dev(d, P, vid, d.sched_slot[vid], d.pos[vid], f, d.elapsed, i_arrived);
  d.flags[vid] = f;
  if (i_skipped >= 0) {
    const int w = d.active[i_skipped];
    if (w >= 0) {
      const bool sb = (d.flags[w] & (VF_COLL | VF_MALF)) != 0;
      d.st_before[w] = sb; d.st_after[w] = sb;
      d.active_idx[w] = 0x7FFFFFFF;
    }
  }
}
// CityModel.remove_vehicle called by the host between ticks (ts_remove_vehicle)
// _update_cached_stats' loop over the scheduled vehicles (dynamic_traffic_generator.py:537-556): per population the sum of
// (elapsed - depart_time), of steps_traveled and the count; over the stuck ones the sum and the maximum of stuck_ticks.  All
// of it is integer-valued (depart times are multiples of the tick length), so the order of the additions cannot show.
// out: [0..1] doubles, [2..7] the same words as int64 (dist x 2, n x 2, stuck sum, stuck max).
__global__ void k_live_stats(Dev d, int n_active, double elapsed, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double dur[2] = {0.0, 0.0};
  long long dist[2] = {0, 0}, cnt[2] = {0, 0}, ssum = 0, smax = 0;
  if (i < n_active) {
    const int vid = d.active[i];
    if (vid >= 0 && (d.flags[vid] & VF_ALIVE)) {
      const int pop = d.pop[vid];
      const int k = pop == TS_POP_INTERNAL ? 0 : pop == TS_POP_THROUGH ? 1 : -1;
      if (k >= 0) { dur[k] = elapsed - d.depart[vid]; dist[k] = d.steps[vid]; cnt[k] = 1; }
      if (d.flags[vid] & VF_STUCK) { ssum = d.stuck_ticks[vid]; smax = ssum; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    for (int k = 0; k < 2; k++) {
      dur[k] += __shfl_down(dur[k], o);
      dist[k] += __shfl_down(dist[k], o);
      cnt[k] += __shfl_down(cnt[k], o);
    }
    ssum += __shfl_down(ssum, o);
    smax = max(smax, (long long)__shfl_down(smax, o));
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* oi = reinterpret_cast<unsigned long long*>(out);
    for (int k = 0; k < 2; k++) {
      if (cnt[k]) { atomicAdd(&out[k], dur[k]); atomicAdd(&oi[2 + k], (unsigned long long)dist[k]); atomicAdd(&oi[4 + k], (unsigned long long)cnt[k]); }
    }
    if (ssum) atomicAdd(&oi[6], (unsigned long long)ssum);
    if (smax) atomicMax(&oi[7], (unsigned long long)smax);
  }
}
__global__ void k_remove_one(Dev d, int vid, int pop_arg) {
  if (threadIdx.x || blockIdx.x) return;
  uint16_t f = d.flags[vid];
  remove_vehicle_dev(d, vid, d.sched_slot[vid], d.pos[vid], f, 0, pop_arg);
  d.flags[vid] = f;
}
// (schedule slot, rank) of CityBlocks (which = 0, ids = block index) or vehicles (which = 1, ids = vehicle id)
__global__ void k_gather_ranks(Dev d, const int32_t* ids, int n, int which, int32_t* out) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int slot = which == 0 ? d.bslot[ids[k]] : d.sched_slot[ids[k]];
  out[2 * k] = slot; out[2 * k + 1] = (int)d.rank[slot];
}
// ServiceVehicleAgent._finish_service, device part: _unpark, new target, phase (vehicle_service.py:106-141);
// _compute_path's cooldown reset (vehicle_base.py:147)
__global__ void k_svc_finish(Dev d, TsParams P, int vid, int target, int to_block) {
  if (threadIdx.x || blockIdx.x) return;
  uint16_t f = d.flags[vid];
  if (f & VF_PARKED) { f &= ~VF_PARKED; atomicAdd((unsigned long long*)&d.cnt->parked, (unsigned long long)-1LL); }
  f &= ~(VF_SERVICING | VF_TOBLOCK);
  if (to_block) f |= VF_TOBLOCK; else f &= ~VF_KEEP;   // remove_on_arrival = True on the way out
  d.flags[vid] = f;
  d.target[vid] = target;
  d.cooldown[vid] = P.pathfinding_cooldown;
}
__global__ void k_flags_or(Dev d, int vid, int bits) {
  if (threadIdx.x || blockIdx.x) return;
  d.flags[vid] |= (uint16_t)bits;
}

__global__ void k_rank_invert(const uint32_t* perm, uint32_t* rank, int n) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) rank[perm[q]] = (uint32_t)q;
}


// ---------------------------------------------------------------------------------------------
// decide-phase RNG bookkeeping on the device.  Per vehicle the draw byte F says: 2 words for the malfunction
// draw, 2 for a sideswipe draw, then a rejection-sampled speed roll.  The fixed parts are a prefix sum
// (pass 1, here); only the rolls form a serial chain (pass 2, host, ~0.45 chain steps per vehicle through the
// producer's take table); pass 3 (k_rng_apply) turns stream positions into decisions for every vehicle.
// ---------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 4;
__device__ __forceinline__ uint2 rng_item(uint8_t f) {
  return make_uint2(2u * (f & 1u) + 2u * ((f >> 1) & 1u), (f >> 2) & 1u);  // (fixed words, is a roller)
}
__global__ void k_rng_blocksum(const uint8_t* F, int start, int n, uint2* block_sums) {
  __shared__ uint2 wsum[BLK / 64];
  const int base = blockIdx.x * BLK * RS_ITEMS + threadIdx.x * RS_ITEMS;
  uint2 a = make_uint2(0, 0);
  for (int j = 0; j < RS_ITEMS; j++) {
    int i = base + j;
    if (i < n) { uint2 v = rng_item(F[start + i]); a.x += v.x; a.y += v.y; }
  }
  for (int o = 32; o; o >>= 1) { a.x += __shfl_down(a.x, o); a.y += __shfl_down(a.y, o); }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint2 t = make_uint2(0, 0);
    for (int w = 0; w < BLK / 64; w++) { t.x += wsum[w].x; t.y += wsum[w].y; }
    block_sums[blockIdx.x] = t;
  }
}
__global__ void k_rng_scanblocks(uint2* block_sums, int nb, unsigned int* total) {  // single block, exclusive, in place
  __shared__ uint2 carry;
  __shared__ uint2 buf[1024];
  if (threadIdx.x == 0) carry = make_uint2(0, 0);
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    int i = base + threadIdx.x;
    uint2 v = i < nb ? block_sums[i] : make_uint2(0, 0);
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      uint2 t = threadIdx.x >= o ? buf[threadIdx.x - o] : make_uint2(0, 0);
      __syncthreads();
      buf[threadIdx.x].x += t.x; buf[threadIdx.x].y += t.y;
      __syncthreads();
    }
    if (i < nb) block_sums[i] = make_uint2(carry.x + buf[threadIdx.x].x - v.x, carry.y + buf[threadIdx.x].y - v.y);
    __syncthreads();
    if (threadIdx.x == 1023) { carry.x += buf[1023].x; carry.y += buf[1023].y; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { total[0] = carry.x; total[1] = carry.y; }
}
__global__ void k_rng_final(Dev d, int start, int n, const uint2* block_off) {
  __shared__ uint2 tsum[BLK];
  const int base = blockIdx.x * BLK * RS_ITEMS + threadIdx.x * RS_ITEMS;
  uint2 item[RS_ITEMS];
  uint2 a = make_uint2(0, 0);
  for (int j = 0; j < RS_ITEMS; j++) {
    int i = base + j;
    item[j] = i < n ? rng_item(d.F[start + i]) : make_uint2(0, 0);
    a.x += item[j].x; a.y += item[j].y;
  }
  tsum[threadIdx.x] = a;
  __syncthreads();
  for (int o = 1; o < BLK; o <<= 1) {  // inclusive scan of the per-thread sums
    uint2 t = threadIdx.x >= o ? tsum[threadIdx.x - o] : make_uint2(0, 0);
    __syncthreads();
    tsum[threadIdx.x].x += t.x; tsum[threadIdx.x].y += t.y;
    __syncthreads();
  }
  uint2 run = block_off[blockIdx.x];
  run.x += tsum[threadIdx.x].x - a.x; run.y += tsum[threadIdx.x].y - a.y;
  for (int j = 0; j < RS_ITEMS; j++) {
    int i = base + j;
    if (i >= n) break;
    d.Cx[start + i] = run.x;          // fixed words consumed by the vehicles before this one (in this pass)
    d.rollrank[start + i] = run.y;    // rolls before this one
    if (item[j].y) d.rollD[run.y] = run.x + item[j].x;  // where its roll starts, apart from earlier rolls' lengths
    run.x += item[j].x; run.y += item[j].y;
  }
}
// take table for the host chain: out[i] = words a speed roll starting at stream position base + i consumes
// (1 + number of rejected words from there on; 0 = more than 64, the host counts those by hand)
__global__ void k_rng_take(const uint32_t* words, unsigned long long base, int n, uint32_t span, int rshift, uint8_t* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long p = base + (unsigned long long)i;
  int t = 1;
  while (t <= 64 && (words[(p + t - 1) & WORDS_MASK] >> rshift) >= span) t++;
  out[i] = t <= 64 ? (uint8_t)t : (uint8_t)0;
}

// pass 3: every vehicle of [start, start + n) reads its words.  base = stream position of vehicle `start`.
__global__ void k_rng_apply(Dev d, int start, int n, unsigned long long base, unsigned long long t_malf,
                            unsigned long long t_swipe, uint32_t span, int rshift, int min_speed) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = start + t;
  const uint32_t f = d.F[i];
  const uint32_t rr = d.rollrank[i];
  const uint32_t tb = d.Tcum[rr];
  unsigned long long w = base + d.Cx[i] + tb;
  uint8_t roll = 0;
  bool fired = false;
  if (f & F_DRAW_MALF) {
    const unsigned long long k = ((unsigned long long)(d.words[w & WORDS_MASK] >> 5) << 26) |
                                 (unsigned long long)(d.words[(w + 1) & WORDS_MASK] >> 6);
    w += 2;
    if (k < t_malf) { atomicMin(&d.cnt->rng_event, (unsigned int)i * 2u); fired = true; }
  }
  if (!fired && (f & F_DRAW_SWIPE)) {
    const unsigned long long k = ((unsigned long long)(d.words[w & WORDS_MASK] >> 5) << 26) |
                                 (unsigned long long)(d.words[(w + 1) & WORDS_MASK] >> 6);
    w += 2;
    if (k < t_swipe) { atomicMin(&d.cnt->rng_event, (unsigned int)i * 2u + 1u); fired = true; }
  } else if (f & F_DRAW_SWIPE) w += 2;
  if (!fired && (f & F_DRAW_SPEED)) {
    const uint32_t tk = d.Tcum[rr + 1] - tb;
    roll = (uint8_t)(min_speed + (int)(d.words[(w + tk - 1) & WORDS_MASK] >> rshift));
  }
  d.R[i] = roll;
}

template <typename T>
__global__ void k_fill(T* p, T v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

inline int nblk(long long n, int per = BLK) { return (int)((n + per - 1) / per); }

}  // namespace
