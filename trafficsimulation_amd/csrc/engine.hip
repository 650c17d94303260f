// engine.hip - MI355X-native per-timestep agent-update engine (gfx950), C-ABI of include/trafficsim.h.
//
// One tick = CityModel.step() (city_model.py:1831-1860):
//   decide  : k_decide_pre  -> host MT19937 scan (data-dependent sequential stream) -> k_decide_main
//   move    : host MT19937 shuffle (model.random) -> rank per scheduled agent ->
//             rounds of { k_move_claim (per-cell min-rank claims) ; k_move_resolve } until every agent
//             has stepped.  An agent executes in the round in which no lower-ranked unresolved agent
//             touches a cell it reads or writes, which reproduces the sequential shuffled order exactly.
//   compact : stable compaction of active_vehicle_agents / schedule after despawns.
//
// State lives in HBM as structure-of-arrays; maps are (H, W) byte planes.  All arithmetic is integer
// except the float32 density map.  See DESIGN.md for the layout and the per-kernel byte counts.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <unordered_map>
#include <vector>

#include "../../include/trafficsim.h"
#include "mt19937.h"

#include "dev.h"
#include "astar.h"

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
        int x = pos % W, y = pos / W;
        const int opposite = (dir + 2) & 3;
        for (int k = 0; k < 2 && cand < 0; k++) {
          int ld = k == 0 ? ((dir + 3) & 3) : ((dir + 1) & 3);  // left, then right
          int nx = x + (ld == 1) - (ld == 3), ny = y + (ld == 0) - (ld == 2);
          if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
          for (int ag = d.cell[ny * W + nx].veh; ag >= 0; ag = d.next_in_cell[ag]) {
            uint16_t af = d.flags[ag];
            bool earlier = d.active_idx[ag] < i;
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
    if (d.active_idx[partner] < my_idx) {
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
      if (pos == d.target[vid]) claim(d, pos, 0, key);
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
    set_occ(d, pos, 0); d.cell[pos].stuck = 0;
    cell_unlink(d, pos, vid);
    f &= ~VF_ALIVE;
    d.sched_kind[s] = K_DEAD;
    d.active[d.active_idx[vid]] = -1;
    int pop = d.pop[vid];
    if (pop == TS_POP_INTERNAL) atomicAdd((unsigned long long*)&d.cnt->live_internal, (unsigned long long)-1LL);
    else if (pop == TS_POP_THROUGH) atomicAdd((unsigned long long*)&d.cnt->live_through, (unsigned long long)-1LL);
    atomicAdd(&d.cnt->deaths, 1);
    if (f & VF_SVC) svc_record(d, key, vid, AR_DESPAWN);
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
      if (pos == d.target[vid] && (claim_rank(cl.x, prefix) < r || claim_rank(cl.y, prefix) < r)) safe = false;
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
                        int* overflow, int* n_overflow) {
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
  d.ev[vid] = 0; d.st_before[vid] = 0; d.st_after[vid] = 0; d.tier_hint[vid] = 0;
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
                             const int8_t* inter) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  Cell x;
  x.claim[0] = x.claim[1] = x.claim[2] = x.claim[3] = 0xFFFFFFFFu;
  x.veh = -1; x.occ = 0; x.stop = 0; x.stuck = 0;
  x.stat = (uint8_t)((allowed[c] & 15) | ((is_road[c] == 1) << 4) | ((inter[c] == 1) << 5) | ((road_type[c] & 3) << 6));
  x.pad_[0] = x.pad_[1] = 0;
  cell[c] = x;
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
__global__ void k_decide_arrive(Dev d, int n_rec) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_rec || d.arr[3 * k + 2] != AR_DECIDE) return;
  const int vid = d.arr[3 * k + 1];
  uint16_t f = d.flags[vid];
  if (!(f & VF_PARKED)) { f |= VF_PARKED; atomicAdd((unsigned long long*)&d.cnt->parked, 1ULL); }
  if (f & VF_TOBLOCK) f = (f & ~VF_TOBLOCK) | VF_SERVICING;
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

// =============================================================================================
// host side
// =============================================================================================
struct ts_engine {
  TsParams P;
  Dev d;
  int W = 0, H = 0, N = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // capacities
  int cap_v = 0, cap_sched = 0;
  size_t pool_cap = 0, pool_used = 0;
  int n_vehicles_total = 0;  // vehicle ids handed out
  int n_active = 0, n_sched = 0;
  int n_sched_vehicles = 0;
  int clock_slot = -1;
  bool mixed_order = false;  // a non-vehicle agent was scheduled after a vehicle
  int groups_scheduled = 0;
  bool lights_set = false;
  // double buffers for compaction
  int32_t* active_alt = nullptr;
  int8_t* kind_alt = nullptr;
  int32_t* ref_alt = nullptr;
  int* block_counts = nullptr;
  int cap_blocks = 0;
  int* d_total = nullptr;
  uint32_t* d_crc = nullptr;
  int* d_overflow = nullptr;
  int cap_overflow = 0;
  // pinned host staging
  uint8_t *hF = nullptr, *hR = nullptr;
  uint32_t* hrank = nullptr;
  DevCnt* hcnt = nullptr;
  int* hint = nullptr;
  int cap_host = 0;
  // RNG streams (host)
  MTPipe rng_global, rng_sched;   // word streams pre-generated by producer threads
  uint32_t* d_perm = nullptr;     // shuffled key order (perm[q] = slot stepping at time q)
  uint32_t rank_clock_host = 0xFFFFFFFFu;
  uint32_t epoch = 0;
  TsCounters C;
  std::vector<uint32_t> perm, shuffle_j;
  int32_t* pend_list[2] = {nullptr, nullptr};
  // RainManager / RainAgent (host side, rain.py): clouds are host agents with a schedule entry of their own
  struct Rain { double x, y, dx, dy; int radius; bool stepped = false; bool alive = true; int cx = 0, cy = 0; };
  std::vector<Rain> rains_all;   // indexed by host-agent id - 1 (id 0 is the manager)
  std::vector<int> rains;        // city_model.rains: ids of live clouds in list order
  bool rain_manager = false;
  RainDiscs prev_discs{};         // what the manager saw at its previous step (RainManager._prev_raining)
  int rain_counter = 0, rain_cooldown_left = 0;
  int n_host_agents = 0;         // manager + clouds ever created (hslot entries)
  int cap_hslot = 0;
  // DynamicTrafficAgent (host side): trip schedule + mid-tick spawning behind the clock agent's schedule slot
  struct Trip { int origin, dest; double depart; int kind; };
  struct Generator {
    bool armed = false;
    TsTrafficTables T;
    std::vector<int> blk_type;
    std::vector<std::vector<int>> blk_entr;
    std::vector<int> hw_in, hw_out;
    std::vector<Trip> pending;
    int current_day = 0;
  } gen;
  // CityBlock food / waste and ServiceVehicleAgent loads: host state, advanced in rank order next to the device's
  // move phase (city_block.py, vehicle_service.py)
  struct Block {
    int cells = 0;
    bool needs_food = false, produces_waste = false;
    double max_food = 0, max_waste = 0, food = 0, waste = 0, food_rate = 0, waste_rate = 0, food_rem = 0, waste_rem = 0;
    int ticks_since_food = 0, ticks_since_waste = 0;
    std::vector<int> service_cells;
  };
  std::vector<Block> blocks;
  int blocks_scheduled = 0, cap_bslot = 0;
  struct SvcVeh {
    int vid, type, id, block;     // type: TS_TRIP_SERVICE_FOOD / _WASTE; id: index into the fleet's id pool
    double load, max_load;
    int phase;                    // 0 to_block, 1 servicing, 2 to_exit
    int ticks;                    // service_ticks
    int pos;                      // where it parked (valid while servicing)
    int target;
  };
  std::vector<SvcVeh> svc;                      // live service vehicles
  std::unordered_map<int, int> parked_cells;    // cell -> parked vehicles on it (only service vehicles ever park)
  std::vector<char> sv_live;                    // ids in the scheduler: [food ids..., waste ids...]
  int32_t *d_ids = nullptr, *d_sr = nullptr;    // k_gather_ranks staging
  int cap_ids = 0;
  int fatal = 0;                                // an exception the reference would have raised inside model.step()
  // device-side RNG bookkeeping
  uint32_t* h_words = nullptr;        // pinned storage of the global stream's tempered-word ring
  uint64_t words_uploaded = 0;        // absolute word index up to which d.words mirrors it
  hipStream_t copy_stream = nullptr;
  hipEvent_t words_ev = nullptr;
  uint2* rng_blocks = nullptr;
  int cap_rng_blocks = 0;
  uint32_t *h_rollD = nullptr, *h_Tcum = nullptr;
  int roll_guess = 0;
  uint32_t* bfs_visited = nullptr;   // k_reach_strict scratch: per wave a visited bitmap and a queue
  int32_t* bfs_queue = nullptr;
  int bfs_slots = 0;
  uint8_t *d_take = nullptr, *h_take = nullptr;   // take table of the stream range the next pass will walk
  size_t cap_take = 0;
  uint64_t take_guess = 0;
  // the table for the NEXT tick is built right after this tick's decide phase (copy stream, overlapping the move
  // phase): [take_base, take_base + take_n) in absolute stream positions, ready once take_ev has fired
  uint64_t take_base = 0;
  size_t take_n = 0;
  hipEvent_t take_ev = nullptr;
  std::thread sh_thread, sh2_thread;
  std::atomic<int> sh_progress{0};   // draws extracted so far (shuffle pipeline)
  unsigned sh_gen = 0;
  int sh_err = 0;
  int device = 0;
  hipStream_t perm_stream = nullptr;
  hipEvent_t perm_ev = nullptr;
  std::mutex sh_mu;
  std::condition_variable sh_cv;
  bool sh_done = true, sh_quit = false;
  int sh_n = 0;
  std::vector<void*> allocs;
  // per-kernel HIP-event timing (ts_profile_*)
  // replanning: work lists, scratch tiers, density state, host-side _path_cache
  int32_t* replan_list[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // see run_replans
  int cap_replan = 0;
  static constexpr int N_TIERS = 4;
  ATier tier[N_TIERS];
  bool tier_ready[N_TIERS] = {false, false, false, false};
  bool density_valid = false;
  float *dens_t0 = nullptr, *dens_t1 = nullptr;
  int32_t* d_status = nullptr;
  struct CachedPath { std::vector<uint32_t> words; int len; };
  std::unordered_map<uint64_t, CachedPath> path_cache;
  bool prof = false;
  std::vector<hipEvent_t> ev_pool;
  struct ProfRec { int id; int e0, e1; long long items; };
  std::vector<ProfRec> prof_pending;
  size_t ev_used = 0;
  double prof_ms[24] = {0};
  long long prof_launches[24] = {0}, prof_items[24] = {0};
  double shuffle_ms = 0;
};

static int add_vehicle_planned(ts_handle e, int start, int goal, int pop_type);  // defined with the C-ABI entries
static int plan_vehicle(ts_handle e, int vid, int start, int goal);

namespace {

typedef ts_engine E;

enum { PK_DECIDE_PRE = 0, PK_DECIDE_MAIN, PK_MOVE_CLAIM, PK_MOVE_RESOLVE, PK_COMPACT, PK_EVENT, PK_REPLAN, PK_DENSITY, PK_RNG, PK_REACH, PH_SCAN, PH_SHUFFLE, PH_SHUFFLE_WAIT, PH_DECIDE_WALL, PH_MOVE_WALL, PH_WAIT1, PH_WORDS, PH_WAIT3, PH_NEED, PK_COUNT };
const char* PK_NAMES[PK_COUNT] = {"k_decide_pre", "k_decide_main", "k_move_claim", "k_move_resolve",
                                  "k_compact", "k_apply_event", "k_decide_replan", "k_density", "k_rng", "k_reach_strict",
                                  "host_rng_scan", "host_shuffle", "host_shuffle_wait", "host_decide_wall", "host_move_wall",
                                  "host_wait_pass1", "host_words_upload", "host_wait_pass3", "host_words_need"};

int prof_begin(E* e, int id, long long items) {
  if (!e->prof) return -1;
  if (e->ev_used + 2 > e->ev_pool.size()) {
    for (int k = 0; k < 64; k++) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return -1; e->ev_pool.push_back(ev); }
  }
  int a = (int)e->ev_used, b = a + 1;
  e->ev_used += 2;
  (void)hipEventRecord(e->ev_pool[a], e->stream);
  e->prof_pending.push_back({id, a, b, items});
  return b;
}
inline void prof_end(E* e, int tok) { if (tok >= 0) (void)hipEventRecord(e->ev_pool[tok], e->stream); }
void prof_collect(E* e) {  // call after a stream synchronize
  for (auto& r : e->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_pool[r.e0], e->ev_pool[r.e1]) == hipSuccess) {
      e->prof_ms[r.id] += ms; e->prof_launches[r.id]++; e->prof_items[r.id] += r.items;
    }
  }
  e->prof_pending.clear();
  e->ev_used = 0;
}
#define LAUNCH(e, id, items, kernel, grid, block, ...)                                   \
  do {                                                                                   \
    int _tok = prof_begin((e), (id), (items));                                           \
    hipLaunchKernelGGL(kernel, grid, block, 0, (e)->stream, __VA_ARGS__);                \
    prof_end((e), _tok);                                                                 \
  } while (0)

int fail(E* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  return code;
}
#define HIPOK(expr)                                                                                   \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return fail(e, TS_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                  \
  } while (0)

template <typename T>
hipError_t dalloc(E* e, T** p, size_t n) {
  hipError_t r = hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
  if (r == hipSuccess) e->allocs.push_back((void*)*p);
  return r;
}
void dfree(E* e, void* p) {
  if (!p) return;
  auto it = std::find(e->allocs.begin(), e->allocs.end(), p);
  if (it != e->allocs.end()) e->allocs.erase(it);
  (void)hipFree(p);
}
// grow a device array, preserving `keep` elements
template <typename T>
int regrow(E* e, T** p, size_t keep, size_t n) {
  T* q = nullptr;
  HIPOK(dalloc(e, &q, n));
  if (*p && keep) HIPOK(hipMemcpyAsync(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  dfree(e, *p);
  *p = q;
  return TS_OK;
}

int ensure_vehicle_capacity(E* e, int need_v, int need_sched) {
  if (need_v > e->cap_v) {
    int nc = std::max(need_v, e->cap_v * 2 + 1024);
    Dev& d = e->d;
    size_t k = e->n_vehicles_total;
#define RG(field) { int rc = regrow(e, &d.field, k, (size_t)nc); if (rc) return rc; }
    RG(pos) RG(target) RG(path_len) RG(path_cur) RG(stuck_ticks) RG(cooldown) RG(stranded_left) RG(steps) RG(over_dur)
    RG(det_dur) RG(next_in_cell) RG(active_idx) RG(sched_slot) RG(path_off) RG(base_speed) RG(cur_speed) RG(max_steps)
    RG(dir) RG(pop) RG(flags) RG(depart) RG(ev) RG(st_before) RG(st_after) RG(ev_idx) RG(reach) RG(tier_hint)
    for (int k = 0; k < 4; k++) { RG(ax_start[k]) RG(ax_off[k]) RG(ax_len[k]) }
#undef RG
    { int rc = regrow(e, &e->replan_list[0], 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->replan_list[1], 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->replan_list[2], 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->replan_list[3], 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->replan_list[4], 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->replan_list[5], 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.active, (size_t)e->n_active, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->active_alt, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.F, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.R, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.cand, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.Cx, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.rollrank, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.rollD, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.Tcum, 0, (size_t)nc + 1); if (rc) return rc; }
    {
      int nbk = nblk(nc, BLK * RS_ITEMS) + 1;
      int rc = regrow(e, &e->rng_blocks, 0, (size_t)nbk); if (rc) return rc;
      e->cap_rng_blocks = nbk;
    }
    if (e->h_rollD) (void)hipHostFree(e->h_rollD);
    if (e->h_Tcum) (void)hipHostFree(e->h_Tcum);
    HIPOK(hipHostMalloc((void**)&e->h_rollD, (size_t)nc * 4));
    HIPOK(hipHostMalloc((void**)&e->h_Tcum, ((size_t)nc + 1) * 4));
    e->cap_v = nc;
  }
  if (need_sched > e->cap_sched) {
    int nc = std::max(need_sched, e->cap_sched * 2 + 1024);
    Dev& d = e->d;
    { int rc = regrow(e, &d.sched_kind, (size_t)e->n_sched, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.sched_ref, (size_t)e->n_sched, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.rank, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->d_perm, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->pend_list[0], 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->pend_list[1], 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &d.resolved, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->kind_alt, 0, (size_t)nc); if (rc) return rc; }
    { int rc = regrow(e, &e->ref_alt, 0, (size_t)nc); if (rc) return rc; }
    e->cap_sched = nc;
  }
  int need_host = std::max(need_v, need_sched);
  if (need_host > e->cap_host) {
    int nc = std::max(need_host, e->cap_host * 2 + 1024);
    if (e->hF) (void)hipHostFree(e->hF);
    if (e->hR) (void)hipHostFree(e->hR);
    if (e->perm_stream) HIPOK(hipStreamSynchronize(e->perm_stream));   // a copy out of hrank may still be in flight
    if (e->hrank) (void)hipHostFree(e->hrank);
    HIPOK(hipHostMalloc((void**)&e->hF, nc));
    HIPOK(hipHostMalloc((void**)&e->hR, nc));
    HIPOK(hipHostMalloc((void**)&e->hrank, (size_t)nc * 4));
    e->cap_host = nc;
  }
  int need_blocks = nblk(std::max(e->cap_v, e->cap_sched), BLK * CITEMS) + 1;
  if (need_blocks > e->cap_blocks) {
    { int rc = regrow(e, &e->block_counts, 0, (size_t)need_blocks); if (rc) return rc; }
    e->cap_blocks = need_blocks;
  }
  return TS_OK;
}

int ensure_pool(E* e, size_t need_words) {
  if (need_words <= e->pool_cap) return TS_OK;
  size_t nc = std::max(need_words, e->pool_cap * 2 + (1u << 16));
  if (nc >= (1ull << 32)) return fail(e, TS_E_CAPACITY, "path pool exceeds 2^32 words");
  int rc = regrow(e, &e->d.pool, e->pool_used, nc);
  if (rc) return rc;
  e->pool_cap = nc;
  e->d.pool_cap_words = nc;
  return TS_OK;
}

int sync_counters(E* e) {  // device counters -> e->C
  HIPOK(hipMemcpyAsync(e->hcnt, e->d.cnt, sizeof(DevCnt), hipMemcpyDeviceToHost, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  const DevCnt& c = *e->hcnt;
  e->C.stuck = c.stuck; e->C.collisions = c.collisions; e->C.malfunctions = c.malfunctions;
  e->C.overtaking = c.overtaking; e->C.in_stuck_detour = c.in_stuck_detour; e->C.parked = c.parked;
  e->C.live_internal = c.live_internal; e->C.live_through = c.live_through;
  e->C.count_completed_internal = c.completed_internal; e->C.count_completed_through = c.completed_through;
  e->C.total_distance_internal = c.dist_internal; e->C.total_distance_through = c.dist_through;
  e->C.total_duration_internal = c.dur_internal; e->C.total_duration_through = c.dur_through;
  e->C.astar_calls = c.astar_calls; e->C.astar_expansions = c.astar_exp; e->C.astar_relaxations = c.astar_relax;
  return TS_OK;
}

// stable compaction of active_vehicle_agents (which = 0) or the schedule (which = 1); returns new length
int compact(E* e, int which, int n, int* out_n) {
  Dev& d = e->d;
  int nb = nblk(n, BLK * CITEMS);
  if (n == 0) { *out_n = 0; return TS_OK; }
  int _tok = prof_begin(e, PK_COMPACT, n);
  hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(BLK), 0, e->stream, d.active, d.sched_kind, n, which, e->block_counts);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, e->stream, e->block_counts, nb, e->d_total);
  hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(BLK), 0, e->stream, d, n, which, e->block_counts, e->active_alt,
                     e->kind_alt, e->ref_alt);
  prof_end(e, _tok);
  HIPOK(hipMemcpyAsync(e->hint, e->d_total, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  *out_n = e->hint[0];
  if (which == 0) std::swap(d.active, e->active_alt);
  else { std::swap(d.sched_kind, e->kind_alt); std::swap(d.sched_ref, e->ref_alt); }
  return TS_OK;
}

int pool_from_device(E* e) {
  unsigned long long v = 0;
  HIPOK(hipMemcpy(&v, &e->d.cnt->pool_used, sizeof(v), hipMemcpyDeviceToHost));
  e->pool_used = (size_t)std::min<unsigned long long>(v, e->pool_cap);
  return TS_OK;
}
int pool_to_device(E* e) {
  unsigned long long v = e->pool_used;
  HIPOK(hipMemcpy(&e->d.cnt->pool_used, &v, sizeof(v), hipMemcpyHostToDevice));
  return TS_OK;
}

// scratch tiers for the GPU A*: many small searchers, fewer large ones, a handful that can hold the whole grid
int ensure_tier(E* e, int t) {
  if (e->tier_ready[t]) return TS_OK;
  ATier& T = e->tier[t];
  const long long N = e->N;
  // nodes a search may touch per tier: 2048 / 32 768 / 262 144 / the whole map; the smaller a search's footprint,
  // the more of them run side by side.  The last tier gets as many slots as ~16 GB of scratch allow (4 .. 64).
  const long long caps[ts_engine::N_TIERS] = {std::min<long long>(N, 2048), std::min<long long>(N, 32768),
                                              std::min<long long>(N, 262144), N};
  const long long last_bytes = N * (2 * 16 + 4 * 17 + 5 * 4);   // table + heap + dir bytes + five cell buffers, per slot
  const int last_slots = (int)std::max<long long>(4, std::min<long long>(64, (16ll << 30) / std::max<long long>(last_bytes, 1)));
  // slots by a ~16 GB budget per tier (allocated on first use; small maps cap the per-slot size at N)
  const int slots[ts_engine::N_TIERS] = {16384, 4096, 512, last_slots};
  T.cap = (int)caps[t];
  T.n_slots = slots[t];
  uint32_t hs = 1;
  while (hs < 2ull * (unsigned long long)T.cap) hs <<= 1;
  T.hsize = hs;
  T.heap_cap = (int)std::min<long long>(4ll * T.cap, 0x7FFFFFF0ll);
  const size_t S = (size_t)T.n_slots;
  HIPOK(dalloc(e, &T.ht, S * hs));
  HIPOK(dalloc(e, &T.hq, S * T.heap_cap));
  HIPOK(dalloc(e, &T.hd, S * T.heap_cap));
  HIPOK(dalloc(e, &T.cells, S * ((size_t)5 * T.cap + 3 * MAXB)));
  HIPOK(dalloc(e, &T.slot_epoch, S));
  HIPOK(hipMemsetAsync(T.ht, 0, S * hs * sizeof(HEnt), e->stream));
  HIPOK(hipMemsetAsync(T.slot_epoch, 0, S * 4, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  e->tier_ready[t] = true;
  return TS_OK;
}

// density_map as of the last tick start (city_model.py:1853), materialised only when a search may need it
int ensure_density(E* e, const int8_t* occ_src) {
  const size_t N = e->N;
  if (!e->dens_t0) { HIPOK(dalloc(e, &e->dens_t0, N)); HIPOK(dalloc(e, &e->dens_t1, N)); }
  if (!e->d.density) HIPOK(dalloc(e, &e->d.density, N));
  const int r = e->P.vehicle_awareness_range;
  int tok = prof_begin(e, PK_DENSITY, (long long)N);
  hipLaunchKernelGGL(k_density_pass0, dim3(nblk((long long)N)), dim3(BLK), 0, e->stream, occ_src, e->d.is_road, e->W, e->H, r,
                     e->dens_t0, e->dens_t1);
  hipLaunchKernelGGL(k_density_pass1, dim3(nblk((long long)N)), dim3(BLK), 0, e->stream, e->dens_t0, e->dens_t1, e->W, e->H, r,
                     e->d.density);
  prof_end(e, tok);
  return TS_OK;
}

// garbage-collect / grow the path pool so that at least `need_free` words are available
int pool_make_room(E* e, size_t need_free) {
  int rc = pool_from_device(e);
  if (rc) return rc;
  if (e->pool_used + need_free <= e->pool_cap && e->pool_used * 4 < e->pool_cap * 3) return TS_OK;
  // copy the live words of every active vehicle into a fresh pool (twice as large if it was mostly live)
  size_t new_cap = e->pool_cap;
  for (int attempt = 0; attempt < 2; attempt++) {
    if (new_cap >= (1ull << 32)) return fail(e, TS_E_CAPACITY, "path pool exceeds 2^32 words");
    uint32_t* np = nullptr;
    unsigned long long* used = nullptr;
    HIPOK(dalloc(e, &np, new_cap));
    HIPOK(dalloc(e, &used, 1));
    HIPOK(hipMemsetAsync(used, 0, sizeof(unsigned long long), e->stream));
    // the GC kernel rewrites offsets in place, so it can only run once per pool: size the target generously
    if (attempt == 0 && e->pool_used * 2 > new_cap) { dfree(e, np); dfree(e, used); new_cap = std::min<size_t>(new_cap * 2, (1ull << 32) - 1); continue; }
    if (e->n_active > 0)
      hipLaunchKernelGGL(k_pool_gc, dim3(nblk(e->n_active)), dim3(BLK), 0, e->stream, e->d, e->n_active, np, used);
    unsigned long long u = 0;
    HIPOK(hipMemcpyAsync(&u, used, sizeof(u), hipMemcpyDeviceToHost, e->stream));
    HIPOK(hipStreamSynchronize(e->stream));
    dfree(e, e->d.pool);
    dfree(e, used);
    e->d.pool = np; e->pool_cap = new_cap; e->d.pool_cap_words = new_cap; e->pool_used = (size_t)u;
    rc = pool_to_device(e);
    if (rc) return rc;
    break;
  }
  if (e->pool_used + need_free > e->pool_cap) {
    rc = ensure_pool(e, (e->pool_used + need_free) * 2);
    if (rc) return rc;
  }
  return TS_OK;
}

inline double now_ms();
// k_decide_replan over the work lists.  Stage 0 keeps the search structures in LDS (one wave per vehicle), stages
// 1-4 are the HBM tiers; a search that outgrows its stage moves to the next one, and k_decide_main queues every
// vehicle directly on the stage its last search fitted in (Dev::tier_hint).
// Lists: 0 / 4 / 1 / 2 / 5 = input of stages 0..4, 3 = pool-full retries.  Counters (replan_n): 0 / 5 / 1 / 2 / 6 for
// those inputs, 3 = retries, 4 = beyond the last tier (an error).
struct ReplanStage { int tier; int in_list; int in_counter; int out_list; int out_counter; };
static const ReplanStage REPLAN_STAGES[5] = {{-1, 0, 0, 4, 5}, {0, 4, 5, 1, 1}, {1, 1, 1, 2, 2}, {2, 2, 2, 5, 6}, {3, 5, 6, 3, 4}};
inline int replan_pending(const int* n8) { return n8[0] + n8[5] + n8[1] + n8[2] + n8[6]; }

int run_replans(E* e) {   // e->hint[8..15] = replan_n as k_decide_main left it
  Dev& d = e->d;
  const TsParams& P = e->P;
  hipStream_t st = e->stream;
  StageCaps caps;
  caps.nodes[0] = LDS_NODES;
  {
    const long long N = e->N;
    const long long c[4] = {std::min<long long>(N, 2048), std::min<long long>(N, 32768), std::min<long long>(N, 262144), N};
    for (int t = 0; t < 4; t++) caps.nodes[t + 1] = (int)c[t];
  }
  while (replan_pending(e->hint + 8) > 0) {
    if (!e->density_valid) { int rc = ensure_density(e, d.occ_snap); if (rc) return rc; e->density_valid = true; }
    {  // strict reachability of every replanner's target (skips the searches that would flood and fail)
      const size_t words_per = ((size_t)e->N + 31) / 32, queue_per = (size_t)e->N;
      if (!e->bfs_visited) {
        size_t per = words_per * 4 + queue_per * 4;
        e->bfs_slots = (int)std::max<size_t>(8, std::min<size_t>(2048, (3ull << 30) / per));
        HIPOK(dalloc(e, &e->bfs_visited, words_per * e->bfs_slots));
        HIPOK(dalloc(e, &e->bfs_queue, queue_per * e->bfs_slots));
      }
      for (const ReplanStage& sg : REPLAN_STAGES) {
        const int n_in = e->hint[8 + sg.in_counter];
        for (int begin = 0; begin < n_in; begin += e->bfs_slots) {
          int cnt = std::min(e->bfs_slots, n_in - begin);
          LAUNCH(e, PK_REACH, cnt, k_reach_strict, dim3(cnt), dim3(64), d, e->replan_list[sg.in_list] + begin, cnt, e->bfs_visited,
                 e->bfs_queue, words_per, queue_per);
        }
      }
    }
    for (int sidx = 0; sidx < 5; sidx++) {
      const ReplanStage& sg = REPLAN_STAGES[sidx];
      const int n = e->hint[8 + sg.in_counter];   // hinted entries plus what the previous stage overflowed
      if (n <= 0) continue;
      const int t = sg.tier < 0 ? 0 : sg.tier;   // the LDS stage borrows the first tier's cell buffers
      int rc = ensure_tier(e, t);
      if (rc) return rc;
      const ATier& T = e->tier[t];
      for (int begin = 0; begin < n; begin += T.n_slots) {
        int cnt = std::min(T.n_slots, n - begin);
        if (sg.tier < 0)
          LAUNCH(e, PK_REPLAN, cnt, k_decide_replan_lds, dim3(cnt), dim3(64), d, P, T, e->replan_list[sg.in_list], begin, cnt,
                 e->replan_list[sg.out_list], sg.out_counter, e->replan_list[3], caps);
        else
          LAUNCH(e, PK_REPLAN, cnt, k_decide_replan, dim3(cnt), dim3(64), d, P, T, e->replan_list[sg.in_list], begin,
                 cnt, e->replan_list[sg.out_list], sg.out_counter, e->replan_list[3], sidx, caps);
      }
      const double tl = now_ms();
      HIPOK(hipMemcpyAsync(e->hint + 8, d.cnt->replan_n, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
      HIPOK(hipStreamSynchronize(st));
      if (e->hint[8 + 4] > 0) return fail(e, TS_E_CAPACITY, "an A* search exceeded the largest scratch tier");
      if (getenv("TS_DEBUG_REPLAN"))
        fprintf(stderr, "[replan] tick %lld stage %d: in=%d overflow=%d retry=%d wait=%.2f ms\n", (long long)e->C.step_count, sidx, n,
                sidx < 4 ? e->hint[8 + sg.out_counter] - 0 : 0, e->hint[8 + 3], now_ms() - tl);
    }
    const int retry = e->hint[8 + 3];
    if (retry == 0) break;
    // the path pool filled up: make room (GC, then growth) and run the entries that could not commit again
    int rc = pool_make_room(e, (size_t)retry * 1024 + (1u << 20));
    if (rc) return rc;
    HIPOK(hipMemcpyAsync(e->replan_list[0], e->replan_list[3], (size_t)retry * 4, hipMemcpyDeviceToDevice, st));
    HIPOK(hipMemsetAsync(d.cnt->replan_n, 0, sizeof(int) * 8, st));
    for (int q = 0; q < 8; q++) e->hint[8 + q] = 0;
    e->hint[8] = retry;
  }
  return TS_OK;
}

// random.shuffle(keys) with model.random: Fisher-Yates from the top with _randbelow's rejection
// sampling, reading pre-generated words.  Two persistent threads form a pipeline: the first extracts the draws
// (they depend only on the word stream and on n), the second applies the swaps a chunk behind it, copies every
// finished stretch of the permutation (position i is final once element i has been swapped) into the pinned
// buffer e->hrank and sends it to the device on its own stream.  Result: d_perm[q] = schedule slot stepping at
// time q (inverted to rank[slot] by k_rank_invert), the clock agent's rank in rank_clock_host.
constexpr int SH_CH = 1 << 12;
void shuffle_draws(E* e, int n) {
  if ((int)e->shuffle_j.size() < n + 64) e->shuffle_j.resize((size_t)n + 64);
  uint32_t* jb = e->shuffle_j.data();   // jb[(n - 1) - i] = draw of element i
  MTPipe& r = e->rng_sched;
  uint64_t w = r.pos();
  uint32_t cnt = 0;
  for (int hi = n - 1; hi >= 1; hi -= SH_CH) {
    const int lo = std::max(1, hi - SH_CH + 1);
    r.need((uint64_t)(hi - lo + 1) * 4 + 512 + (w - r.pos()));
    uint64_t limit = w + (uint64_t)(hi - lo + 1) * 4 + 256;
    // Walk the WORDS in order (addresses are not data dependent, so the loads pipeline): a word is the draw of the
    // current element if it is below i + 1, otherwise it is a rejected try.  The only loop-carried state is the
    // element counter.
    uint32_t nn = (uint32_t)hi + 1;             // i + 1 of the element being drawn
    const uint32_t nn_end = (uint32_t)lo;       // stop once nn == lo  (element lo - 1 is not ours)
    while (nn > nn_end) {
      const int shift = __builtin_clz(nn);      // 32 - bit_length(nn); constant while nn >= 2^(k-1)
      const uint32_t band_end = std::max(nn_end, (1u << (31 - shift)) - 1u);  // last nn of this band, exclusive
      while (nn > band_end) {
        if (w + 256 >= limit) { r.advance_to(w); r.need(8192); limit = w + 8192 - 256; }
        int burst = 64;   // a short unrolled burst; bounds: at most 64 elements / words per burst
        while (burst-- > 0 && nn > band_end) {
          const uint32_t c = r.at(w++) >> shift;
          const uint32_t acc = c < nn;
          jb[cnt] = c;
          cnt += acc;
          nn -= acc;
        }
      }
    }
    r.advance_to(w);
    e->sh_progress.store((int)cnt, std::memory_order_release);
  }
}
void shuffle_swaps(E* e, int n) {
  e->perm.resize(n);
  uint32_t* p = e->perm.data();  // ordinary cached memory, first touched by this thread
  for (int i = 0; i < n; i++) p[i] = (uint32_t)i;
  const uint32_t cs = e->clock_slot >= 0 ? (uint32_t)e->clock_slot : 0xFFFFFFFFu;
  uint32_t cpos = cs;
  const int total = std::max(0, n - 1);
  int done = 0, sent_hi = n;   // positions [sent_hi, n) are already on their way to the device
  auto send = [&](int lo) {    // positions [lo, sent_hi) are final
    if (lo >= sent_hi) return;
    memcpy(e->hrank + lo, p + lo, (size_t)(sent_hi - lo) * 4);
    if (hipMemcpyAsync(e->d_perm + lo, e->hrank + lo, (size_t)(sent_hi - lo) * 4, hipMemcpyHostToDevice, e->perm_stream) != hipSuccess)
      e->sh_err = 1;
    sent_hi = lo;
  };
  while (done < total) {
    const int avail = e->sh_progress.load(std::memory_order_acquire);
    if (avail <= done) { std::this_thread::yield(); continue; }
    const int m = std::min(avail - done, SH_CH);
    const uint32_t* jb = e->shuffle_j.data() + done;
    for (int q = 0; q < m; q++) __builtin_prefetch(&p[jb[q]], 1, 1);
    const int hi = n - 1 - done;
    for (int q = 0; q < m; q++) {
      const int i = hi - q;
      const uint32_t j = jb[q];
      const uint32_t a = p[i], b = p[j];
      p[i] = b; p[j] = a;
      if (a == cs) cpos = j; else if (b == cs) cpos = (uint32_t)i;
    }
    done += m;
    if (sent_hi - (n - done) >= (1 << 18)) send(n - done);
  }
  send(0);
  if (hipEventRecord(e->perm_ev, e->perm_stream) != hipSuccess) e->sh_err = 1;
  e->rank_clock_host = cs == 0xFFFFFFFFu ? 0xFFFFFFFFu : cpos;
}

// persistent workers for the scheduler shuffle (fresh std::threads per tick cost ~50 us each and lose locality)
void shuffle_worker(E* e, int role) {
  if (role == 1) (void)hipSetDevice(e->device);
  std::unique_lock<std::mutex> lk(e->sh_mu);
  unsigned seen = 0;
  for (;;) {
    e->sh_cv.wait(lk, [&]() { return e->sh_gen != seen || e->sh_quit; });
    if (e->sh_quit) return;
    seen = e->sh_gen;
    const int n = e->sh_n;
    lk.unlock();
    const double t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (role == 0) shuffle_draws(e, n); else shuffle_swaps(e, n);
    const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
    lk.lock();
    if (role == 1) { e->shuffle_ms = dt; e->sh_done = true; e->sh_cv.notify_all(); }
  }
}
void shuffle_start(E* e, int n) {
  if (!e->sh_thread.joinable()) { e->sh_thread = std::thread(shuffle_worker, e, 0); e->sh2_thread = std::thread(shuffle_worker, e, 1); }
  std::lock_guard<std::mutex> lk(e->sh_mu);
  e->sh_progress.store(0, std::memory_order_relaxed);
  e->sh_n = n; e->sh_done = false; e->sh_gen++;
  e->sh_cv.notify_all();
}
void shuffle_wait(E* e) {
  std::unique_lock<std::mutex> lk(e->sh_mu);
  e->sh_cv.wait(lk, [e]() { return e->sh_done; });
}

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
  if ((long long)e->n_sched + 1 >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^22 agents");
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
        G.pending.push_back(ts_engine::Trip{oc, dc, t, TS_POP_INTERNAL});
      }
    }
    // service vehicles, uniform per zone (362-376): one entrance draw per trip
    const int Nf = food_q[zi], Nw = waste_q[zi];
    for (int j = 1; j <= Nf; j++) {
      const double t = z0 + (double)((long long)j * (long long)span) / (double)(Nf + 1);
      const int sc = G.hw_in[r.randbelow((uint32_t)G.hw_in.size())];
      G.pending.push_back(ts_engine::Trip{sc, -1, t, TS_TRIP_SERVICE_FOOD});
    }
    for (int j = 1; j <= Nw; j++) {
      const double t = z0 + (double)((long long)j * (long long)span) / (double)(Nw + 1);
      const int sc = G.hw_in[r.randbelow((uint32_t)G.hw_in.size())];
      G.pending.push_back(ts_engine::Trip{sc, -1, t, TS_TRIP_SERVICE_WASTE});
    }
    long long thr = (long long)std::nearbyint((double)G.T.passing_population_per_day * z.through_distribution);
    thr -= Nf + Nw;   // SERVICE_VEHICLES_COUNT_AS_THROUGH defaults to True (90, 381-382)
    for (long long q = 0; q < thr; q++) {
      const double t = z0 + r.random() * span;
      const int ent = G.hw_in[r.randbelow((uint32_t)G.hw_in.size())];
      const int ex = G.hw_out[r.randbelow((uint32_t)G.hw_out.size())];
      G.pending.push_back(ts_engine::Trip{ent, ex, t, TS_POP_THROUGH});
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
  if ((long long)e->n_sched + 1 >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^22 agents");
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

// DynamicTrafficAgent.step (153-194) and _spawn (398-416), executed at the agent's place in the shuffled order:
// every lower-ranked agent has stepped on the device, every higher-ranked one has not yet.
int generator_step(E* e) {
  auto& G = e->gen;
  const double prev = e->C.elapsed;
  e->C.elapsed += e->P.time_per_step_seconds;
  const double total_secs = G.T.start_offset_seconds + e->C.elapsed;
  const int new_day = (int)std::floor(total_secs / 86400.0);
  if (new_day > G.current_day) {
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
    if (t.origin == t.dest) return fail(e, TS_E_UNSUPPORTED, "generated trip with origin == destination");
    if ((long long)e->n_sched + 1 >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^22 agents");
    int rc = add_vehicle_planned(e, t.origin, t.dest, t.kind);
    if (rc) return rc;
  }
  return TS_OK;
}

// mirror the global stream's tempered words [uploaded, upto) into the device ring (copy stream + event)
int words_upload(E* e, uint64_t upto) {
  MTPipe& r = e->rng_global;
  if (e->words_uploaded < r.pos()) e->words_uploaded = r.pos();
  if (upto <= e->words_uploaded) return TS_OK;
  if (upto - r.pos() > MTPipe::MAX_AHEAD_BLOCKS * 600ull)
    return fail(e, TS_E_CAPACITY, "one decide pass would read more of the MT19937 stream than the ring holds");
  { const double t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    r.need(upto - r.pos());
    if (e->prof) { e->prof_ms[PH_WAIT1 + 3] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; e->prof_launches[PH_WAIT1 + 3]++; } }
  uint64_t a = e->words_uploaded;
  while (a < upto) {
    const uint64_t off = a & (MTPipe::TW_CAP - 1);
    const uint64_t len = std::min<uint64_t>(upto - a, MTPipe::TW_CAP - off);
    HIPOK(hipMemcpyAsync(e->d.words + off, e->h_words + off, len * 4, hipMemcpyHostToDevice, e->copy_stream));
    a += len;
  }
  HIPOK(hipEventRecord(e->words_ev, e->copy_stream));
  e->words_uploaded = upto;
  return TS_OK;
}

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void host_prof(E* e, int id, double ms, long long items) {
  if (!e->prof) return;
  e->prof_ms[id] += ms; e->prof_launches[id]++; e->prof_items[id] += items;
}

int tick(E* e) {
  Dev& d = e->d;
  const TsParams& P = e->P;
  hipStream_t st = e->stream;
  const int nA = e->n_active, nS = e->n_sched;
  // the scheduler stream is independent of everything the decide phase does: shuffle on a host thread
  shuffle_start(e, nS);
  struct Joiner { E* e; bool done = false; ~Joiner() { if (!done) shuffle_wait(e); } } joiner{e};
  const double t_tick0 = now_ms();

  // density_map is a function of the occupancy at this point (city_model.py:1853)
  HIPOK(hipMemcpyAsync(d.occ_snap, d.occ, (size_t)e->N, hipMemcpyDeviceToDevice, st));
  e->density_valid = false;
  d.elapsed = e->C.elapsed;
  const bool svc_on = !e->svc.empty();
  int arr_read = 0;   // service records of this tick consumed so far
  if (svc_on) HIPOK(hipMemsetAsync(&d.cnt->arr_n, 0, sizeof(int), st));
  // fetch records [arr_read, upto) from the device
  std::vector<int32_t> recs;
  auto fetch_records = [&](int upto) -> int {
    recs.clear();
    if (upto > d.arr_cap) return fail(e, TS_E_CAPACITY, "more service records in one tick than the record buffer holds");
    if (upto <= arr_read) return TS_OK;
    recs.resize((size_t)(upto - arr_read) * 3);
    HIPOK(hipMemcpyAsync(recs.data(), d.arr + 3 * (size_t)arr_read, recs.size() * 4, hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    arr_read = upto;
    return TS_OK;
  };
  // ---------------- decide ----------------
  if (nA > 0) {
    HIPOK(hipMemsetAsync(d.ev, 0, (size_t)e->n_vehicles_total, st));
    HIPOK(hipMemsetAsync(d.cnt->replan_n, 0, sizeof(int) * 8, st));
    // random() < c  <=>  the 53-bit integer (a << 26 | b) < ceil(c * 2^53)   (exact: power-of-two scaling)
    auto thr53 = [](double c) -> unsigned long long {
      if (!(c > 0.0)) return 0;
      if (c >= 1.0) return 1ull << 53;
      return (unsigned long long)std::ceil(std::ldexp(c, 53));
    };
    const unsigned long long T_malf = thr53(P.malfunction_chance), T_swipe = thr53(P.sideswipe_chance);
    const uint32_t span = (uint32_t)(P.vehicle_max_speed - P.vehicle_min_speed + 1);
    const int rshift = __builtin_clz(span);  // getrandbits(span.bit_length())
    MTPipe& r = e->rng_global;
    RLists rlists;
    for (int q = 0; q < 6; q++) rlists.l[q] = e->replan_list[q];
    // vehicles per pass (bounds the look-ahead into the word ring); TS_DEBUG_SEG shrinks it so that tests can walk
    // the multi-pass path on small worlds
    static const int SEG = getenv("TS_DEBUG_SEG") ? std::max(64, atoi(getenv("TS_DEBUG_SEG"))) : (1 << 20);
    int start = 0;
    bool main_done = false;
    LAUNCH(e, PK_DECIDE_PRE, nA, k_decide_pre, dim3(nblk(nA)), dim3(BLK), d, P, 0, nA);
    while (start < nA) {
      const int seg_end = std::min(nA, start + SEG), n = seg_end - start;
      const int nb = nblk(n, BLK * RS_ITEMS);
      // pass 1 (device): fixed-word prefix sums, roll ranks, roll start offsets
      HIPOK(hipMemsetAsync(&d.cnt->rng_event, 0xFF, sizeof(unsigned int), st));
      {
        int tok = prof_begin(e, PK_RNG, n);
        hipLaunchKernelGGL(k_rng_blocksum, dim3(nb), dim3(BLK), 0, st, d.F, start, n, e->rng_blocks);
        hipLaunchKernelGGL(k_rng_scanblocks, dim3(1), dim3(1024), 0, st, e->rng_blocks, nb, d.cnt->rng_tot);
        hipLaunchKernelGGL(k_rng_final, dim3(nb), dim3(BLK), 0, st, d, start, n, e->rng_blocks);
        prof_end(e, tok);
      }
      // take table for the stretch of the stream this pass will most likely walk (estimate from the last pass;
      // positions beyond it fall back to the accept bitmask on the host)
      const uint64_t base = r.pos();
      // the table built ahead of time covers [take_base, take_base + take_n); `toff` = where this pass starts in it
      size_t n_take = 0, toff = 0;
      if (e->take_n > 0 && base >= e->take_base && base - e->take_base + 4096 < e->take_n) {
        toff = (size_t)(base - e->take_base);
        n_take = e->take_n - toff;
        HIPOK(hipEventSynchronize(e->take_ev));
      } else {
        n_take = (size_t)std::min<uint64_t>(e->cap_take, e->take_guess);
        if (n_take > 0) {
          if (e->words_uploaded < base + n_take + 64) n_take = e->words_uploaded > base + 64 ? (size_t)(e->words_uploaded - base - 64) : 0;
        }
        if (n_take > 0) {
          HIPOK(hipStreamWaitEvent(st, e->words_ev, 0));
          hipLaunchKernelGGL(k_rng_take, dim3(nblk((long long)n_take)), dim3(BLK), 0, st, d.words, (unsigned long long)base,
                             (int)n_take, span, rshift, e->d_take);
          HIPOK(hipMemcpyAsync(e->h_take, e->d_take, n_take, hipMemcpyDeviceToHost, st));
        }
        e->take_base = base; e->take_n = n_take;   // (valid once the sync below has passed)
      }
      const int guess = std::min(n, e->roll_guess);
      HIPOK(hipMemcpyAsync(e->hint + 4, d.cnt->rng_tot, sizeof(unsigned int) * 2, hipMemcpyDeviceToHost, st));
      if (guess > 0) HIPOK(hipMemcpyAsync(e->h_rollD, d.rollD, (size_t)guess * 4, hipMemcpyDeviceToHost, st));
      const double t_w1 = now_ms();
      HIPOK(hipStreamSynchronize(st));
      host_prof(e, PH_WAIT1, now_ms() - t_w1, n);
      const uint32_t Ctot = (uint32_t)e->hint[4];
      const int cnt = e->hint[5];
      if (cnt > guess) {
        HIPOK(hipMemcpyAsync(e->h_rollD + guess, d.rollD + guess, (size_t)(cnt - guess) * 4, hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
      }
      e->roll_guess = cnt + cnt / 8 + 1024;
      // pass 2 (host): the serial chain over the speed rolls.  Roll k starts at base + rollD[k] + (words taken by
      // the rolls before it); the producer thread tabulated how many words a roll takes from any position.
      const double t_scan0 = now_ms();
      uint32_t* Tcum = e->h_Tcum;
      const uint8_t* take_tab = e->h_take + toff;
      const uint32_t* rollD = e->h_rollD;
      Tcum[0] = 0;
      {
        uint64_t T = 0, ensured = 0;
        int k0 = 0;
        {
          // fast path while the walk stays inside the device-built table: nothing but the dependent chain
          // T -> address -> byte load -> T (32-bit arithmetic, four rolls per trip)
          uint32_t T32 = 0;
          const uint32_t lim = (uint32_t)std::min<size_t>(n_take, 0x7FFFFFFFu);
          int k = 0;
          for (; k + 4 <= cnt; k += 4) {
            const uint32_t r0 = rollD[k], r1 = rollD[k + 1], r2 = rollD[k + 2], r3 = rollD[k + 3];
            if ((uint64_t)r3 + T32 + 256 >= lim) break;
            const uint32_t t0 = take_tab[r0 + T32]; const uint32_t a0 = T32 + t0;
            const uint32_t t1 = take_tab[r1 + a0]; const uint32_t a1 = a0 + t1;
            const uint32_t t2 = take_tab[r2 + a1]; const uint32_t a2 = a1 + t2;
            const uint32_t t3 = take_tab[r3 + a2]; const uint32_t a3 = a2 + t3;
            if (__builtin_expect((t0 == 0) | (t1 == 0) | (t2 == 0) | (t3 == 0), 0)) break;   // a run the table does not record
            Tcum[k + 1] = a0; Tcum[k + 2] = a1; Tcum[k + 3] = a2; Tcum[k + 4] = a3;
            T32 = a3;
          }
          k0 = k; T = T32;
        }
        for (int k = k0; k < cnt; k++) {
          const uint64_t pos = base + rollD[k] + T;
          if (__builtin_expect(pos + 8 >= ensured, 0)) {
            const uint64_t want = (pos - r.pos()) + (1u << 18);
            r.need(want + 1248);
            ensured = pos + (1u << 18) - 64;
          }
          const uint64_t rel = pos - base;
          uint32_t t = rel < n_take ? take_tab[rel] : r.take(pos);
          if (__builtin_expect(t == 0, 0)) {  // run longer than the table records: count it here
            uint64_t q = pos;
            for (;;) {
              r.need((q - r.pos()) + 8);
              if ((r.at(q++) >> rshift) < span) break;
            }
            t = (uint32_t)(q - pos);
          }
          T += t;
          Tcum[k + 1] = (uint32_t)T;
        }
        host_prof(e, PH_SCAN, now_ms() - t_scan0, n);
        const uint64_t final_pos = base + Ctot + T;
        e->take_guess = (Ctot + T) + (Ctot + T) / 8 + (1u << 16);
        // the words this pass reads must be on the device (usually prefetched during the previous tick)
        const double t_wu = now_ms();
        int rc = words_upload(e, final_pos + 8);
        if (rc) return rc;
        host_prof(e, PH_WORDS, now_ms() - t_wu, n);
        HIPOK(hipStreamWaitEvent(st, e->words_ev, 0));
        HIPOK(hipMemcpyAsync(d.Tcum, Tcum, ((size_t)cnt + 1) * 4, hipMemcpyHostToDevice, st));
        // pass 3 (device): every vehicle reads its words: malfunction / sideswipe tests, rolled speeds
        {
          int tok = prof_begin(e, PK_RNG, n);
          hipLaunchKernelGGL(k_rng_apply, dim3(nblk(n)), dim3(BLK), 0, st, d, start, n, (unsigned long long)base, T_malf,
                             T_swipe, span, rshift, P.vehicle_min_speed);
          prof_end(e, tok);
        }
        if (seg_end == nA) {  // k_decide_main returns at once if a draw fired (the fix-up below re-runs it)
          LAUNCH(e, PK_DECIDE_MAIN, nA, k_decide_main, dim3(nblk(nA)), dim3(BLK), d, P, nA, rlists);
          HIPOK(hipMemcpyAsync(e->hint + 8, d.cnt->replan_n, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
        }
        HIPOK(hipMemcpyAsync(e->hint + 6, &d.cnt->rng_event, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        const double t_w3 = now_ms();
        HIPOK(hipStreamSynchronize(st));
        host_prof(e, PH_WAIT3, now_ms() - t_w3, n);
        const unsigned int evk = (unsigned int)e->hint[6];
        if (evk == 0xFFFFFFFFu) {
          r.advance_to(final_pos);
          start = seg_end;
          main_done = seg_end == nA;
          continue;
        }
        // rare: a malfunction / sideswipe fired at vehicle ev_at.  Everything before it stands; apply the event,
        // move the stream to just behind its draws and re-derive the draw bytes of the suffix.
        e->C.rng_fixups++;
        const int ev_at = (int)(evk >> 1), ev_coll = (int)(evk & 1u);
        uint32_t cx = 0, rr = 0;
        uint8_t fbyte = 0;
        HIPOK(hipMemcpyAsync(&e->hint[0], d.active + ev_at, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&e->hint[1], d.cand + ev_at, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&cx, d.Cx + ev_at, 4, hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&rr, d.rollrank + ev_at, 4, hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&fbyte, d.F + ev_at, 1, hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        const uint64_t after = base + cx + Tcum[rr] + (ev_coll ? ((fbyte & F_DRAW_MALF) ? 4u : 2u) : 2u);
        r.advance_to(after);
        LAUNCH(e, PK_EVENT, 1, k_apply_event, dim3(1), dim3(64), d, P, e->hint[0], ev_coll, e->hint[1], ev_at);
        start = ev_at + 1;
        if (start < nA) LAUNCH(e, PK_DECIDE_PRE, nA - start, k_decide_pre, dim3(nblk(nA - start)), dim3(BLK), d, P, start, nA);
      }
    }
    if (!main_done) {  // the last pass ended with an event at the very last vehicle (or there was no pass left)
      HIPOK(hipMemsetAsync(&d.cnt->rng_event, 0xFF, sizeof(unsigned int), st));
      LAUNCH(e, PK_DECIDE_MAIN, nA, k_decide_main, dim3(nblk(nA)), dim3(BLK), d, P, nA, rlists);
      HIPOK(hipMemcpyAsync(e->hint + 8, d.cnt->replan_n, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
      HIPOK(hipStreamSynchronize(st));
    }
    // prefetch the part of the stream the next tick will most likely read
    { const double t_wu = now_ms(); int rc = words_upload(e, r.pos() + (uint64_t)nA * 4 + (1u << 16)); if (rc) return rc;
      host_prof(e, PH_WORDS, now_ms() - t_wu, nA); }
    {
      // ... and build the next tick's take table behind that upload, on the copy stream: kernel and download
      // overlap the move phase instead of sitting in front of the next host chain
      const uint64_t nb = r.pos();
      size_t nt = (size_t)std::min<uint64_t>(e->cap_take, e->take_guess);
      if (e->words_uploaded < nb + nt + 64) nt = e->words_uploaded > nb + 64 ? (size_t)(e->words_uploaded - nb - 64) : 0;
      e->take_n = 0;
      static const bool ahead = !getenv("TS_NO_TAKE_AHEAD");
      if (nt > 0 && ahead) {
        hipLaunchKernelGGL(k_rng_take, dim3(nblk((long long)nt)), dim3(BLK), 0, e->copy_stream, d.words, (unsigned long long)nb,
                           (int)nt, span, rshift, e->d_take);
        HIPOK(hipMemcpyAsync(e->h_take, e->d_take, nt, hipMemcpyDeviceToHost, e->copy_stream));
        HIPOK(hipEventRecord(e->take_ev, e->copy_stream));
        e->take_base = nb; e->take_n = nt;
      }
    }
    if (replan_pending(e->hint + 8) > 0) { int rc = run_replans(e); if (rc) return rc; }
    if (svc_on) {
      // on_target_reached inside step_decide for vehicles that stay on the grid (vehicle_base.py:657-661): apply the
      // flag changes now that no decider can see them half-way, then the host part in decide order
      HIPOK(hipMemcpyAsync(e->hint + 2, &d.cnt->arr_n, sizeof(int), hipMemcpyDeviceToHost, st));
      HIPOK(hipStreamSynchronize(st));
      const int n_rec = e->hint[2];
      if (n_rec > 0) {
        int rc = fetch_records(n_rec);
        if (rc) return rc;
        hipLaunchKernelGGL(k_decide_arrive, dim3(nblk(n_rec)), dim3(BLK), 0, st, d, n_rec);
        std::vector<std::pair<int, int>> order;   // (decide index, vehicle)
        for (int k = 0; k < n_rec; k++) if (recs[3 * k + 2] == AR_DECIDE) order.push_back({recs[3 * k], recs[3 * k + 1]});
        std::sort(order.begin(), order.end());
        for (auto& o : order) { int k = svc_find(e, o.second); if (k >= 0) svc_start(e, e->svc[k]); }
      }
    }
  }

  // ---------------- move (schedule.step) ----------------
  const double t_dec1 = now_ms();
  host_prof(e, PH_DECIDE_WALL, t_dec1 - t_tick0, nA);
  shuffle_wait(e);
  joiner.done = true;
  host_prof(e, PH_SHUFFLE_WAIT, now_ms() - t_dec1, nS);
  host_prof(e, PH_SHUFFLE, e->shuffle_ms, nS);
  const double t_move0 = now_ms();
  const uint32_t rank_clock = e->rank_clock_host;
  const int sched_vehicles_at_shuffle = e->n_sched_vehicles;
  const double elapsed0 = e->C.elapsed;
  if (nS > 0) {
    if (e->sh_err) return fail(e, TS_E_DEVICE, "the shuffle thread could not send the permutation to the device");
    HIPOK(hipStreamWaitEvent(st, e->perm_ev, 0));   // the permutation went up on its own stream while the decide phase ran
    hipLaunchKernelGGL(k_rank_invert, dim3(nblk(nS)), dim3(BLK), 0, st, e->d_perm, d.rank, nS);
    HIPOK(hipMemsetAsync(d.resolved, 0, (size_t)nS, st));
    HIPOK(hipMemsetAsync(&d.cnt->resolved, 0, sizeof(int) * 2, st));  // resolved, deaths
    // Round 1 covers every slot; later rounds only the slots that were still blocked (ping-pong lists).
    // With an armed traffic generator the phase runs in two parts: first every agent ranked before it, then the
    // generator's own step on the host (spawns plan on the maps as they are at that point), then the rest.
    const bool split = e->gen.armed && e->clock_slot >= 0;
    // Host-side agents (rain manager, rain clouds) step at their ranks too.  They only touch host state and the
    // global stream, so they need no device synchronisation of their own: the ones ranked before the traffic
    // generator run now, the others after its step.  Clouds created during this tick do not step.
    struct HostEv { uint32_t rank; int hid; int slot; };
    std::vector<HostEv> host_events;
    RainDiscs discs; discs.n = -1;   // -1: the manager has not stepped in this tick
    int host_deaths = 0;
    if (e->rain_manager) {
      const int nh = e->n_host_agents;
      std::vector<int32_t> slots(nh);
      HIPOK(hipMemcpyAsync(slots.data(), d.hslot, (size_t)nh * 4, hipMemcpyDeviceToHost, st));
      HIPOK(hipStreamSynchronize(st));
      std::vector<uint32_t> ranks(nh);
      for (int hdx = 0; hdx < nh; hdx++) {
        if (hdx > 0 && !e->rains_all[(size_t)hdx - 1].alive) continue;
        HIPOK(hipMemcpyAsync(&ranks[hdx], d.rank + slots[hdx], 4, hipMemcpyDeviceToHost, st));
      }
      HIPOK(hipStreamSynchronize(st));
      for (int hdx = 0; hdx < nh; hdx++) {
        if (hdx > 0 && !e->rains_all[(size_t)hdx - 1].alive) continue;
        host_events.push_back(HostEv{ranks[hdx], hdx, slots[hdx]});
      }
      std::sort(host_events.begin(), host_events.end(), [](const HostEv& a, const HostEv& b) { return a.rank < b.rank; });
    }
    // CityBlocks step on the host at their ranks; service vehicles whose load timer runs out in this tick do
    // _finish_service there too (it plans a path on the maps as they are at that point, like the generator's spawns)
    struct Point { uint32_t rank; int kind; int ref; int slot; };   // kind 0 = the clock agent, 1 = finishing service vehicle
    std::vector<Point> points;
    struct StaticEv { uint32_t rank; int kind; int ref; };          // kind 0 = rain event (index), 1 = CityBlock
    std::vector<StaticEv> static_events;
    for (size_t k = 0; k < host_events.size(); k++) static_events.push_back(StaticEv{host_events[k].rank, 0, (int)k});
    if (split) points.push_back(Point{rank_clock, 0, 0, e->clock_slot});
    {
      std::vector<int32_t> ids;
      const int nb = std::min((int)e->blocks.size(), e->blocks_scheduled);
      for (int b = 0; b < nb; b++) ids.push_back(b);
      std::vector<int> fin;   // indices into e->svc
      for (size_t k = 0; k < e->svc.size(); k++) {
        auto& v = e->svc[k];
        if (v.phase != 1) continue;
        if (v.ticks <= 1) { fin.push_back((int)k); ids.push_back(v.vid); }   // service_ticks -= 1; <= 0 -> _finish_service
        else v.ticks -= 1;
      }
      if (!ids.empty()) {
        if ((int)ids.size() > e->cap_ids) {
          const int nc = (int)ids.size() * 2 + 64;
          int rc = regrow(e, &e->d_ids, 0, (size_t)nc); if (rc) return rc;
          rc = regrow(e, &e->d_sr, 0, (size_t)nc * 2); if (rc) return rc;
          e->cap_ids = nc;
        }
        std::vector<int32_t> sr(ids.size() * 2);
        HIPOK(hipMemcpyAsync(e->d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, st));
        if (nb > 0) hipLaunchKernelGGL(k_gather_ranks, dim3(nblk(nb)), dim3(BLK), 0, st, d, e->d_ids, nb, 0, e->d_sr);
        if (!fin.empty())
          hipLaunchKernelGGL(k_gather_ranks, dim3(nblk((long long)fin.size())), dim3(BLK), 0, st, d, e->d_ids + nb, (int)fin.size(), 1,
                             e->d_sr + 2 * nb);
        HIPOK(hipMemcpyAsync(sr.data(), e->d_sr, sr.size() * 4, hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        for (int b = 0; b < nb; b++) static_events.push_back(StaticEv{(uint32_t)sr[2 * b + 1], 1, b});
        for (size_t q = 0; q < fin.size(); q++)
          points.push_back(Point{(uint32_t)sr[2 * (nb + q) + 1], 1, e->svc[fin[q]].vid, sr[2 * (nb + q)]});
      }
    }
    std::sort(static_events.begin(), static_events.end(), [](const StaticEv& a, const StaticEv& b) { return a.rank < b.rank; });
    std::sort(points.begin(), points.end(), [](const Point& a, const Point& b) { return a.rank < b.rank; });
    size_t se_cur = 0;
    // host-side work of every agent ranked below `hi`, in rank order: rain, CityBlocks, and what the device reported
    // about service vehicles (arrivals -> _start_service, despawns)
    auto run_window = [&](uint32_t hi) -> int {
      struct Ev { uint32_t rank; int kind; int ref; };   // kind 0 rain, 1 block, 2 service start, 3 service despawn
      std::vector<Ev> evs;
      while (se_cur < static_events.size() && static_events[se_cur].rank < hi) {
        evs.push_back(Ev{static_events[se_cur].rank, static_events[se_cur].kind, static_events[se_cur].ref});
        se_cur++;
      }
      if (svc_on) {
        const int upto = e->hint[2];
        int rc = fetch_records(upto);
        if (rc) return rc;
        for (size_t k = 0; k + 2 < recs.size(); k += 3) {
          if (recs[k + 2] == AR_START) evs.push_back(Ev{(uint32_t)recs[k], 2, recs[k + 1]});
          else if (recs[k + 2] == AR_DESPAWN) evs.push_back(Ev{(uint32_t)recs[k], 3, recs[k + 1]});
        }
      }
      std::stable_sort(evs.begin(), evs.end(), [](const Ev& a, const Ev& b) { return a.rank < b.rank; });
      if (getenv("TS_DEBUG_EVENTS"))
        for (const Ev& ev : evs)
          fprintf(stderr, "[events] tick %lld rank %u kind %d ref %d%s\n", (long long)e->C.step_count, ev.rank, ev.kind, ev.ref,
                  ev.kind == 0 ? (host_events[ev.ref].hid == 0 ? " (rain manager)" : " (cloud)") : "");
      for (const Ev& ev : evs) {
        if (ev.kind == 0) {
          const HostEv& h = host_events[ev.ref];
          if (h.hid == 0) { int rc = rain_manager_step(e, discs); if (rc) return rc; }
          else if (rain_agent_step(e, h.hid)) {
            const int8_t dead = K_DEAD;   // schedule.remove(self)
            HIPOK(hipMemcpyAsync(d.sched_kind + h.slot, &dead, 1, hipMemcpyHostToDevice, st));
            HIPOK(hipStreamSynchronize(st));
            host_deaths++;
          }
        } else if (ev.kind == 1) {
          block_step(e, ev.ref);
        } else {
          const int k = svc_find(e, ev.ref);
          if (k < 0) continue;
          if (ev.kind == 2) svc_start(e, e->svc[k]);
          else {
            auto& v = e->svc[k];
            if (v.id >= 0) e->sv_live[(size_t)(v.type == TS_TRIP_SERVICE_FOOD ? 0 : e->gen.T.total_service_vehicles_food) + v.id] = 0;
            if (v.type == TS_TRIP_SERVICE_FOOD) e->C.live_service_food--; else e->C.live_service_waste--;
            e->svc.erase(e->svc.begin() + k);
          }
        }
      }
      return TS_OK;
    };
    e->hint[0] = 0; e->hint[1] = 0; e->hint[2] = arr_read; e->hint[3] = 0;
    int done = 0;
    for (size_t pi = 0; pi <= points.size(); pi++) {
      const bool last = pi == points.size();
      const uint32_t rank_limit = last ? NO_RANK : points[pi].rank;
      const int target = last ? nS : (int)points[pi].rank;
      int round_no = 0, pending_bound = nS;
      HIPOK(hipMemsetAsync(d.cnt->pend_n, 0, sizeof(int) * 2, st));
      while (done < target) {
        const int chunk = round_no == 0 ? 1 : 4;
        for (int rr = 0; rr < chunk; rr++, round_no++) {
          if ((e->epoch % EPOCHS) == 0) {  // epoch prefix wrapped: stale keys would win again -> clear once
            size_t n = (size_t)e->N;
            hipLaunchKernelGGL(k_claims_reset, dim3(nblk((long long)n)), dim3(BLK), 0, st, d.cell, (int)n);
            HIPOK(hipMemsetAsync(d.gclaim_r, 0xFF, (size_t)std::max(d.G, 1) * 4, st));
          }
          const uint32_t prefix = (EPOCHS - 1) - (e->epoch % EPOCHS);
          e->epoch++;
          const int in = round_no & 1, out = in ^ 1;   // round r reads list[r & 1] (none in round 0), writes the other
          const int32_t* in_list = round_no == 0 ? nullptr : e->pend_list[in];
          const int grid_items = round_no == 0 ? nS : pending_bound;
          HIPOK(hipMemsetAsync(&d.cnt->pend_n[out], 0, sizeof(int), st));
          const int flat = round_no == 0 && d.gc_n > 0 && e->groups_scheduled == d.G && P.light_algorithm != TS_LIGHTS_DISABLED;
          LAUNCH(e, PK_MOVE_CLAIM, grid_items, k_move_claim, dim3(nblk(grid_items)), dim3(BLK), d, P, nS, prefix, in_list,
                 &d.cnt->pend_n[in], rank_limit, flat);
          if (flat) LAUNCH(e, PK_MOVE_CLAIM, d.gc_n, k_move_claim_groups, dim3(nblk(d.gc_n)), dim3(BLK), d, prefix, rank_limit);
          LAUNCH(e, PK_MOVE_RESOLVE, grid_items, k_move_resolve, dim3(nblk(grid_items)), dim3(BLK), d, P, nS, prefix,
                 rank_clock, elapsed0, in_list, &d.cnt->pend_n[in], e->pend_list[out], &d.cnt->pend_n[out], rank_limit);
          e->C.move_rounds++;
        }
        HIPOK(hipMemcpyAsync(e->hint, &d.cnt->resolved, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        int now = e->hint[0];
        if (now == done && now < target) return fail(e, TS_E_DEVICE, "move phase made no progress (internal error)");
        done = now;
        pending_bound = std::max(1, target - done);
      }
      const int dev_error = e->hint[3], dev_deaths = e->hint[1];
      { int rc = run_window(last ? NO_RANK : rank_limit); if (rc) return rc; }
      if (!last) {
        const Point& pt = points[pi];
        if (pt.kind == 0) {
          // the generator's turn: DynamicTrafficAgent.step on the host
          int rc = generator_step(e);
          if (rc) return rc;
        } else {
          const int k = svc_find(e, pt.ref);
          if (k >= 0) { int rc = svc_finish(e, e->svc[k]); if (rc) return rc; }
        }
        // mark the agent's slot as stepped
        const uint8_t one = 1;
        HIPOK(hipMemcpyAsync(d.resolved + pt.slot, &one, 1, hipMemcpyHostToDevice, st));
        done += 1;
        HIPOK(hipMemcpyAsync(&d.cnt->resolved, &done, sizeof(int), hipMemcpyHostToDevice, st));
        HIPOK(hipStreamSynchronize(st));
      }
      e->hint[3] = dev_error; e->hint[1] = dev_deaths;
    }
    if (e->hint[3]) return fail(e, e->hint[3], "device-side error: a vehicle sits on its target during decide (start == goal is not supported)");
    if (discs.n >= 0)   // RainManager.step ran: rain_map is exactly the union of the discs it saw
    {
      hipLaunchKernelGGL(k_rain_map, dim3(nblk((long long)e->N)), dim3(BLK), 0, st, d.rain, e->W, e->H, e->prev_discs, discs);
      e->prev_discs = discs;
    }
    e->C.agent_steps += sched_vehicles_at_shuffle;
    const int vehicle_deaths = e->hint[1];
    const int deaths = vehicle_deaths + host_deaths;
    if (deaths > 0) {
      int na = 0, ns = 0;
      int rc = compact(e, 0, e->n_active, &na); if (rc) return rc;   // spawns of this tick are part of the lists by now
      rc = compact(e, 1, e->n_sched, &ns); if (rc) return rc;
      e->n_active = na; e->n_sched = ns; e->n_sched_vehicles -= vehicle_deaths;
      if (e->clock_slot >= 0 && e->mixed_order) {
        // the clock never dies, but dead vehicles scheduled before it shift its slot: find it again
        std::vector<int8_t> kinds(ns);
        HIPOK(hipMemcpy(kinds.data(), d.sched_kind, ns, hipMemcpyDeviceToHost));
        e->clock_slot = -1;
        for (int q = 0; q < ns; q++) if (kinds[q] == TS_AGENT_CLOCK) { e->clock_slot = q; break; }
      }
    }
  }
  host_prof(e, PH_MOVE_WALL, now_ms() - t_move0, nS);
  if (e->clock_slot >= 0 && !e->gen.armed) e->C.elapsed += P.time_per_step_seconds;  // an armed generator did it in its step
  e->C.step_count++;
  if (e->prof) { HIPOK(hipStreamSynchronize(st)); prof_collect(e); }
  return TS_OK;
}

}  // namespace

static int g_device = 0;  // ts_set_device
extern "C" {

void ts_default_params(TsParams* p) {
  memset(p, 0, sizeof(*p));
  p->vehicle_min_speed = 1; p->vehicle_max_speed = 5; p->vehicle_awareness_range = 10;
  p->rain_enabled = 1; p->rain_speed_reduction = 2;
  p->pathfinding_cooldown = 5; p->pathfinding_cache = 1;
  p->stuck_recompute_threshold = 30; p->stuck_recompute_threshold_intersection = 1;
  p->contraflow_overtake_active = 1; p->max_contraflow_overtake_steps = 6; p->contraflow_overtake_duration = 30;
  p->stuck_contraflow_enabled = 1; p->stuck_contraflow_threshold = 60; p->stuck_contraflow_threshold_intersection = 10;
  p->max_contraflow_stuck_detour_steps = 20; p->contraflow_stuck_detour_duration = 10;
  p->malfunction_active = 1; p->malfunction_duration = 400; p->malfunction_chance = 1e-7;
  p->sideswipe_active = 1; p->sideswipe_duration = 600; p->sideswipe_chance = 1e-9;
  p->contraflow_penalty = 5000; p->obstacle_penalty_vehicle = 1000; p->obstacle_penalty_stop = 500;
  p->road_type_penalties_enabled = 1; p->turn_penalty_enabled = 1; p->turn_penalty = 10;
  p->dynamic_penalties_enabled = 1;
  p->road_type_penalty_r1 = 0.5; p->road_type_penalty_r2 = 5; p->road_type_penalty_r3 = 50.0;
  p->dynamic_penalty_scale = 4.0;
  p->light_algorithm = TS_LIGHTS_QUEUE_ACTUATED;
  p->transition_duration_enabled = 0; p->transition_clearance_enabled = 1; p->all_red_duration = 2;
  p->green_duration = 20; p->qa_min_green = 5; p->qa_max_green = 30; p->qa_gap = 3;
  p->enable_traffic = 1; p->time_per_step_seconds = 6; p->eager_density = 0;
  p->rain_radius_min = 50; p->rain_radius_max = 100; p->rain_occurrences_max = 3; p->rain_cooldown = 86400;
  p->rain_spawn_offset = 10; p->rain_spawn_chance = 0.1;
}

int ts_create(const TsWorld* w, const TsParams* params, ts_handle* out) {
  if (!w || !params || !out || w->width <= 0 || w->height <= 0 || !w->allowed_dirs_map || !w->is_road_map ||
      !w->road_type_map || !w->intersection_map)
    return TS_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return TS_E_DEVICE;
  if (hipSetDevice(g_device) != hipSuccess) return TS_E_DEVICE;
  E* e = new E();
  memset(&e->d, 0, sizeof(e->d));
  memset(&e->C, 0, sizeof(e->C));
  e->P = *params;
  e->W = w->width; e->H = w->height; e->N = w->width * w->height;
  Dev& d = e->d;
  d.W = e->W; d.H = e->H; d.N = e->N;
  d.w_magic = (!getenv("TS_NO_MAGIC") && e->W < (1 << 14) && (long long)e->N <= (1ll << 26)) ? ((1ull << 40) / (unsigned long long)e->W + 1ull) : 0ull;
  size_t N = e->N;
  auto bail = [&](int code) { ts_destroy(e); return code; };
  if (hipStreamCreate(&e->stream) != hipSuccess) return bail(TS_E_DEVICE);
#define A(ptr, n) if (dalloc(e, &ptr, (size_t)(n)) != hipSuccess) return bail(TS_E_DEVICE);
  A(d.occ, N) A(d.stop, N) A(d.rain, N) A(d.is_road, N) A(d.cell, N) A(d.gclaim_r, 1)
  uint8_t* t_allowed = nullptr;   // static planes only needed to build the cell records
  int8_t *t_road_type = nullptr, *t_inter = nullptr;
  if (hipMalloc((void**)&t_allowed, N) != hipSuccess || hipMalloc((void**)&t_road_type, N) != hipSuccess ||
      hipMalloc((void**)&t_inter, N) != hipSuccess) {
    (void)hipFree(t_allowed); (void)hipFree(t_road_type); (void)hipFree(t_inter);
    return bail(TS_E_DEVICE);
  }
  A(d.cnt, 1) A(e->d_total, 1) A(e->d_crc, 256) A(d.occ_snap, N) A(e->d_status, 4)
#undef A
  hipStream_t st = e->stream;
  bool ok = true;
  ok &= hipMemsetAsync(d.occ, 0, N, st) == hipSuccess;
  ok &= hipMemsetAsync(d.stop, 0, N, st) == hipSuccess;
  ok &= hipMemsetAsync(d.rain, 0, N, st) == hipSuccess;
  ok &= hipMemsetAsync(d.occ_snap, 0, N, st) == hipSuccess;  // _update_density_map() on the fresh, empty model
  ok &= hipMemsetAsync(d.cnt, 0, sizeof(DevCnt), st) == hipSuccess;
  ok &= hipMemcpyAsync(t_allowed, w->allowed_dirs_map, N, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(d.is_road, w->is_road_map, N, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(t_road_type, w->road_type_map, N, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(t_inter, w->intersection_map, N, hipMemcpyHostToDevice, st) == hipSuccess;
  hipLaunchKernelGGL(k_cells_init, dim3(nblk((long long)N)), dim3(BLK), 0, st, d.cell, (int)N, t_allowed, d.is_road, t_road_type, t_inter);
  uint32_t table[256];
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320U ^ (c >> 1)) : (c >> 1);
    table[i] = c;
  }
  ok &= hipMemcpyAsync(e->d_crc, table, sizeof(table), hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipHostMalloc((void**)&e->hcnt, sizeof(DevCnt)) == hipSuccess;
  ok &= hipHostMalloc((void**)&e->hint, sizeof(int) * 16) == hipSuccess;
  ok &= hipStreamSynchronize(st) == hipSuccess;
  (void)hipFree(t_allowed); (void)hipFree(t_road_type); (void)hipFree(t_inter);
  if (!ok) return bail(TS_E_DEVICE);
  ok = hipHostMalloc((void**)&e->h_words, MTPipe::TW_CAP * 4) == hipSuccess;
  ok &= dalloc(e, &d.words, (size_t)MTPipe::TW_CAP) == hipSuccess;
  ok &= hipStreamCreate(&e->copy_stream) == hipSuccess;
  ok &= hipStreamCreateWithFlags(&e->perm_stream, hipStreamNonBlocking) == hipSuccess;
  ok &= hipEventCreateWithFlags(&e->perm_ev, hipEventDisableTiming) == hipSuccess;
  ok &= hipEventCreateWithFlags(&e->take_ev, hipEventDisableTiming) == hipSuccess;
  e->device = g_device;
  ok &= hipEventCreateWithFlags(&e->words_ev, hipEventDisableTiming) == hipSuccess;
  if (!ok) return bail(TS_E_DEVICE);
  e->rng_global.use_storage(e->h_words);
  e->cap_take = 6u << 20;
  if (dalloc(e, &e->d_take, e->cap_take) != hipSuccess) return bail(TS_E_DEVICE);
  if (hipHostMalloc((void**)&e->h_take, e->cap_take) != hipSuccess) return bail(TS_E_DEVICE);
  e->epoch = 0;
  e->rng_global.set_roll((uint32_t)std::max(1, params->vehicle_max_speed - params->vehicle_min_speed + 1));
  if (ensure_vehicle_capacity(e, 1024, 1024) != TS_OK) return bail(TS_E_DEVICE);
  if (ensure_pool(e, 1 << 16) != TS_OK) return bail(TS_E_DEVICE);
  *out = e;
  return TS_OK;
}

int ts_destroy(ts_handle e) {
  if (!e) return TS_OK;
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->sh_thread.joinable()) {
    { std::lock_guard<std::mutex> lk(e->sh_mu); e->sh_quit = true; }
    e->sh_cv.notify_all();
    e->sh_thread.join();
    e->sh2_thread.join();
  }
  if (e->perm_stream) { (void)hipStreamSynchronize(e->perm_stream); (void)hipStreamDestroy(e->perm_stream); }
  if (e->perm_ev) (void)hipEventDestroy(e->perm_ev);
  if (e->take_ev) (void)hipEventDestroy(e->take_ev);
  for (void* p : e->allocs) (void)hipFree(p);
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->hF) (void)hipHostFree(e->hF);
  if (e->hR) (void)hipHostFree(e->hR);
  if (e->hrank) (void)hipHostFree(e->hrank);
  if (e->hcnt) (void)hipHostFree(e->hcnt);
  if (e->h_rollD) (void)hipHostFree(e->h_rollD);
  if (e->h_Tcum) (void)hipHostFree(e->h_Tcum);
  if (e->h_take) (void)hipHostFree(e->h_take);
  if (e->words_ev) (void)hipEventDestroy(e->words_ev);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->hint) (void)hipHostFree(e->hint);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  uint32_t* hw = e->h_words;
  delete e;                       // stops the producer threads before their ring storage goes away
  if (hw) (void)hipHostFree(hw);
  return TS_OK;
}

const char* ts_last_error(ts_handle h) { return h ? h->err.c_str() : "null handle"; }

int ts_set_lights(ts_handle e, const TsLightTables* t) {
  if (!e || !t || t->n_groups < 0 || t->n_lights < 0) return TS_E_INVALID;
  if (e->lights_set) return fail(e, TS_E_STATE, "ts_set_lights may be called once");
  const int W = e->W, H = e->H, G = t->n_groups, L = t->n_lights;
  Dev& d = e->d;
  auto cells = [&](const int32_t* xy, int n, std::vector<int32_t>& out) {
    out.resize(n);
    for (int i = 0; i < n; i++) {
      int x = xy[2 * i], y = xy[2 * i + 1];
      if (x < 0 || x >= W || y < 0 || y >= H) return false;
      out[i] = y * W + x;
    }
    return true;
  };
  auto up = [&](int32_t** dst, const int32_t* src, size_t n) -> int {
    HIPOK(dalloc(e, dst, n));
    if (n) HIPOK(hipMemcpyAsync(*dst, src, n * 4, hipMemcpyHostToDevice, e->stream));
    return TS_OK;
  };
  std::vector<int32_t> tmp;
#define UPOFF(dst, src, n) { int rc = up(&d.dst, t->src, (size_t)(n)); if (rc) return rc; }
#define UPCELLS(dst, src, n) { if (!cells(t->src, (n), tmp)) return fail(e, TS_E_INVALID, #src " out of bounds"); \
    int rc = up(&d.dst, tmp.data(), (size_t)(n)); if (rc) return rc; HIPOK(hipStreamSynchronize(e->stream)); }
  UPOFF(g_light_off, g_light_off, G + 1)
  UPCELLS(light_cell, light_xy, L)
  UPOFF(light_ctrl_off, light_ctrl_off, L + 1)
  UPCELLS(light_ctrl, light_ctrl_xy, t->light_ctrl_off[L])
  UPOFF(g_ns_off, g_ns_off, G + 1) UPOFF(g_ns, g_ns, t->g_ns_off[G])
  UPOFF(g_ew_off, g_ew_off, G + 1) UPOFF(g_ew, g_ew, t->g_ew_off[G])
  for (int k = 0; k < t->g_ns_off[G]; k++) if (t->g_ns[k] < 0 || t->g_ns[k] >= L) return fail(e, TS_E_INVALID, "g_ns light index");
  for (int k = 0; k < t->g_ew_off[G]; k++) if (t->g_ew[k] < 0 || t->g_ew[k] >= L) return fail(e, TS_E_INVALID, "g_ew light index");
  UPOFF(g_icell_off, g_icell_off, G + 1) UPCELLS(g_icell, g_icell_xy, t->g_icell_off[G])
  UPOFF(g_nsin_off, g_ns_in_off, G + 1) UPCELLS(g_nsin, g_ns_in_xy, t->g_ns_in_off[G])
  UPOFF(g_nsout_off, g_ns_out_off, G + 1) UPCELLS(g_nsout, g_ns_out_xy, t->g_ns_out_off[G])
  UPOFF(g_ewin_off, g_ew_in_off, G + 1) UPCELLS(g_ewin, g_ew_in_xy, t->g_ew_in_off[G])
  UPOFF(g_ewout_off, g_ew_out_off, G + 1) UPCELLS(g_ewout, g_ew_out_xy, t->g_ew_out_off[G])
#undef UPOFF
#undef UPCELLS
  {
    // flat (cell, group, plane) list of everything a group claims in the move phase (k_move_claim_groups)
    std::vector<int32_t> fc, fg;
    std::vector<uint8_t> fp;
    std::vector<int32_t> cidx;
    const bool outs = e->P.light_algorithm == TS_LIGHTS_PRESSURE_CONTROL || e->P.light_algorithm == TS_LIGHTS_NEIGHBOR_PRESSURE_CONTROL;
    auto add_range = [&](const int32_t* off, const int32_t* xy, int g, int plane) {
      for (int k = off[g]; k < off[g + 1]; k++) { fc.push_back(xy[2 * k + 1] * W + xy[2 * k]); fg.push_back(g); fp.push_back((uint8_t)plane); }
    };
    for (int g = 0; g < G; g++) {
      add_range(t->g_icell_off, t->g_icell_xy, g, 1);
      add_range(t->g_ns_in_off, t->g_ns_in_xy, g, 1);
      add_range(t->g_ew_in_off, t->g_ew_in_xy, g, 1);
      if (outs) { add_range(t->g_ns_out_off, t->g_ns_out_xy, g, 1); add_range(t->g_ew_out_off, t->g_ew_out_xy, g, 1); }
      for (int l = t->g_light_off[g]; l < t->g_light_off[g + 1]; l++) {
        fc.push_back(t->light_xy[2 * l + 1] * W + t->light_xy[2 * l]); fg.push_back(g); fp.push_back(2);
        add_range(t->light_ctrl_off, t->light_ctrl_xy, l, 2);
        for (size_t q = fg.size() - (size_t)(t->light_ctrl_off[l + 1] - t->light_ctrl_off[l]); q < fg.size(); q++) fg[q] = g;
      }
    }
    d.gc_n = (int)fc.size();
    HIPOK(dalloc(e, &d.gc_cell, fc.size())); HIPOK(dalloc(e, &d.gc_group, fg.size())); HIPOK(dalloc(e, &d.gc_plane, fp.size()));
    if (!fc.empty()) {
      HIPOK(hipMemcpy(d.gc_cell, fc.data(), fc.size() * 4, hipMemcpyHostToDevice));
      HIPOK(hipMemcpy(d.gc_group, fg.data(), fg.size() * 4, hipMemcpyHostToDevice));
      HIPOK(hipMemcpy(d.gc_plane, fp.data(), fp.size(), hipMemcpyHostToDevice));
    }
  }
  std::vector<int32_t> nb((size_t)G * 8, -1), nbc((size_t)G * 8, -1);
  if (t->g_neighbors) nb.assign(t->g_neighbors, t->g_neighbors + (size_t)G * 8);
  const int32_t* nc = t->g_neighbors_ctor ? t->g_neighbors_ctor : t->g_neighbors;
  if (nc) nbc.assign(nc, nc + (size_t)G * 8);
  for (int g = 0; g < G * 4; g++)
    if (nb[g * 2 + 1] >= G || nbc[g * 2 + 1] >= G) return fail(e, TS_E_INVALID, "neighbor group index out of range");
  { int rc = up(&d.g_nb, nb.data(), nb.size()); if (rc) return rc; }
  { int rc = up(&d.g_nb_ctor, nbc.data(), nbc.size()); if (rc) return rc; }
  HIPOK(hipStreamSynchronize(e->stream));
  // state: pending_phase = 0 unless DISABLED (intersection_light_group.py:115-116)
  int32_t** zero_fields[] = {&d.gs_trans, &d.gs_clear, &d.gs_ftphase, &d.gs_fttimer, &d.gs_qtimer, &d.gs_gap,
                             &d.gs_last, &d.gs_nsp, &d.gs_ewp, &d.gs_repop, &d.g_slot};
  for (auto f : zero_fields) {
    HIPOK(dalloc(e, f, (size_t)G));
    HIPOK(hipMemsetAsync(*f, 0, (size_t)std::max(G, 1) * 4, e->stream));
  }
  HIPOK(dalloc(e, &d.gs_cur, (size_t)G));
  HIPOK(dalloc(e, &d.gs_pend, (size_t)G));
  HIPOK(hipMemsetAsync(d.gs_cur, 0xFF, (size_t)std::max(G, 1) * 4, e->stream));
  if (e->P.light_algorithm != TS_LIGHTS_DISABLED) HIPOK(hipMemsetAsync(d.gs_pend, 0, (size_t)std::max(G, 1) * 4, e->stream));
  else HIPOK(hipMemsetAsync(d.gs_pend, 0xFF, (size_t)std::max(G, 1) * 4, e->stream));
  dfree(e, d.gclaim_r);
  d.gclaim_r = nullptr;
  HIPOK(dalloc(e, &d.gclaim_r, (size_t)G));
  HIPOK(hipMemsetAsync(d.gclaim_r, 0xFF, (size_t)std::max(G, 1) * 4, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  d.G = G;
  e->lights_set = true;
  e->groups_scheduled = 0;
  return TS_OK;
}

int ts_schedule_add(ts_handle e, int32_t kind, int32_t count) {
  if (!e || count < 0) return TS_E_INVALID;
  if (kind != TS_AGENT_LIGHT_GROUP && kind != TS_AGENT_NOOP && kind != TS_AGENT_CLOCK && kind != TS_AGENT_RAIN_MANAGER &&
      kind != TS_AGENT_CITY_BLOCK)
    return fail(e, TS_E_INVALID, "bad agent kind");
  if (kind == TS_AGENT_RAIN_MANAGER && (count > 1 || e->rain_manager) && count > 0)
    return fail(e, TS_E_INVALID, "at most one rain manager");
  if (kind == TS_AGENT_LIGHT_GROUP && e->groups_scheduled + count > e->d.G)
    return fail(e, TS_E_INVALID, "more group slots than groups");
  if (kind == TS_AGENT_CLOCK && (count > 1 || e->clock_slot >= 0) && count > 0)
    return fail(e, TS_E_INVALID, "at most one clock agent");
  if (count == 0) return TS_OK;
  if (e->n_sched_vehicles > 0) e->mixed_order = true;
  int rc = ensure_vehicle_capacity(e, e->cap_v, e->n_sched + count);
  if (rc) return rc;
  std::vector<int8_t> kinds(count, (int8_t)kind);
  std::vector<int32_t> refs(count, 0), slots(count);
  for (int i = 0; i < count; i++) {
    if (kind == TS_AGENT_LIGHT_GROUP) refs[i] = e->groups_scheduled + i;
    if (kind == TS_AGENT_CITY_BLOCK) refs[i] = e->blocks_scheduled + i;
    slots[i] = e->n_sched + i;
  }
  if (kind == TS_AGENT_CITY_BLOCK) {   // the n-th CityBlock scheduled is block n of city_blocks (city_model.py:1738)
    if (e->blocks_scheduled + count > e->cap_bslot) {
      const int nc = (e->blocks_scheduled + count) * 2 + 64;
      rc = regrow(e, &e->d.bslot, (size_t)e->blocks_scheduled, (size_t)nc);
      if (rc) return rc;
      e->cap_bslot = nc;
    }
    HIPOK(hipMemcpy(e->d.bslot + e->blocks_scheduled, slots.data(), (size_t)count * 4, hipMemcpyHostToDevice));
    e->blocks_scheduled += count;
  }
  HIPOK(hipMemcpy(e->d.sched_kind + e->n_sched, kinds.data(), count, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(e->d.sched_ref + e->n_sched, refs.data(), (size_t)count * 4, hipMemcpyHostToDevice));
  if (kind == TS_AGENT_LIGHT_GROUP) {
    HIPOK(hipMemcpy(e->d.g_slot + e->groups_scheduled, slots.data(), (size_t)count * 4, hipMemcpyHostToDevice));
    e->groups_scheduled += count;
  }
  if (kind == TS_AGENT_CLOCK) e->clock_slot = e->n_sched;
  if (kind == TS_AGENT_RAIN_MANAGER) {   // host agent id 0
    rc = host_agent_register(e, e->n_sched);
    if (rc) return rc;
    e->rain_manager = true;
  }
  e->n_sched += count;
  return TS_OK;
}

int ts_set_traffic_generator(ts_handle e, const TsTrafficTables* t) {
  if (!e || !t || t->n_blocks < 0 || t->n_zones < 0 || t->n_zones > 8) return TS_E_INVALID;
  if (!e->rng_global.seeded()) return fail(e, TS_E_STATE, "seed the global stream before constructing the traffic generator");
  auto& G = e->gen;
  G.T = *t;
  const int W = e->W, H = e->H;
  auto cellxy = [&](const int32_t* xy, int i, int& out) {
    int x = xy[2 * i], y = xy[2 * i + 1];
    if (x < 0 || x >= W || y < 0 || y >= H) return false;
    out = y * W + x;
    return true;
  };
  G.blk_type.assign(t->blk_type, t->blk_type + t->n_blocks);
  G.blk_entr.assign(t->n_blocks, {});
  for (int b = 0; b < t->n_blocks; b++) {
    for (int k = t->blk_entr_off[b]; k < t->blk_entr_off[b + 1]; k++) {
      int c;
      if (!cellxy(t->blk_entr_xy, k, c)) return fail(e, TS_E_INVALID, "block entrance out of bounds");
      G.blk_entr[b].push_back(c);
    }
    if (G.blk_entr[b].empty())
      return fail(e, TS_E_UNSUPPORTED, "a city block without entrances (random.choice([]) raises in the reference)");
  }
  G.hw_in.clear(); G.hw_out.clear();
  for (int k = 0; k < t->n_highway_entrances; k++) { int c; if (!cellxy(t->highway_entrances_xy, k, c)) return TS_E_INVALID; G.hw_in.push_back(c); }
  for (int k = 0; k < t->n_highway_exits; k++) { int c; if (!cellxy(t->highway_exits_xy, k, c)) return TS_E_INVALID; G.hw_out.push_back(c); }
  if ((G.hw_in.empty() || G.hw_out.empty()) && t->passing_population_per_day > 0)
    return fail(e, TS_E_UNSUPPORTED, "through traffic needs highway entrances and exits");
  for (int z = 0; z < t->n_zones; z++) if (t->zones[z].n_internal < 0 || t->zones[z].n_internal > 8) return TS_E_INVALID;
  e->blocks.clear();
  if (t->blk_inner_cells) {
    if (!t->blk_service_off || !t->blk_service_xy) return fail(e, TS_E_INVALID, "blk_service_* tables missing");
    e->blocks.resize(t->n_blocks);
    for (int b = 0; b < t->n_blocks; b++) {
      auto& B = e->blocks[b];
      B.cells = t->blk_inner_cells[b];
      B.needs_food = (t->needs_food_type_mask >> t->blk_type[b]) & 1;
      B.produces_waste = (t->produces_waste_type_mask >> t->blk_type[b]) & 1;
      B.max_food = (double)B.cells * t->food_capacity_per_cell;
      B.max_waste = (double)B.cells * t->waste_capacity_per_cell;
      B.food = B.max_food; B.waste = 0.0;
      B.food_rate = (double)B.cells / (double)t->food_consumption_ticks;
      B.waste_rate = (double)B.cells / (double)t->waste_production_ticks;
      for (int k = t->blk_service_off[b]; k < t->blk_service_off[b + 1]; k++) {
        int c;
        if (!cellxy(t->blk_service_xy, k, c)) return fail(e, TS_E_INVALID, "service road cell out of bounds");
        B.service_cells.push_back(c);
      }
    }
  }
  const int n_sv = t->total_service_vehicles_food + t->total_service_vehicles_waste;
  if (n_sv < 0 || t->total_service_vehicles_food < 0 || t->total_service_vehicles_waste < 0) return TS_E_INVALID;
  if (n_sv > 0 && (e->blocks.empty() || G.hw_in.empty()))
    return fail(e, TS_E_UNSUPPORTED, "service vehicles need the block tables and highway entrances");
  e->sv_live.assign((size_t)n_sv, 0);
  e->svc.clear(); e->parked_cells.clear();
  if (n_sv > 0 && !e->d.arr) {
    e->d.arr_cap = 1 << 16;
    HIPOK(dalloc(e, &e->d.arr, (size_t)e->d.arr_cap * 3));
  }
  G.pending.clear();
  G.current_day = 0;
  G.armed = true;
  generate_day(e, 0);
  e->words_uploaded = e->rng_global.pos();
  return TS_OK;
}

int ts_seed(ts_handle e, int32_t stream, const uint32_t* mt, uint32_t index) {
  if (!e || !mt || stream < 0 || stream > 1 || index > 624) return TS_E_INVALID;
  (stream == TS_RNG_GLOBAL ? e->rng_global : e->rng_sched).seed(mt, index);
  if (stream == TS_RNG_GLOBAL) { e->words_uploaded = e->rng_global.pos(); e->take_n = 0; }
  return TS_OK;
}
int ts_seed_int(ts_handle e, int32_t stream, uint64_t seed) {
  if (!e || stream < 0 || stream > 1) return TS_E_INVALID;
  (stream == TS_RNG_GLOBAL ? e->rng_global : e->rng_sched).seed_u64(seed);
  if (stream == TS_RNG_GLOBAL) { e->words_uploaded = e->rng_global.pos(); e->take_n = 0; }
  return TS_OK;
}
int ts_rng_state(ts_handle e, int32_t stream, uint32_t* mt_out, uint32_t* index_out) {
  if (!e || stream < 0 || stream > 1 || !mt_out || !index_out) return TS_E_INVALID;
  MTPipe& r = stream == TS_RNG_GLOBAL ? e->rng_global : e->rng_sched;
  if (!r.seeded()) return fail(e, TS_E_STATE, "stream not seeded");
  r.state(mt_out, index_out);
  return TS_OK;
}

// shared tail of the two ts_add_vehicles entries: `enc` holds the 2-bit direction words, every
// vehicle starting on a fresh word; poff[i] is relative to enc.
static int add_vehicles_core(ts_handle e, int n, std::vector<int32_t>& start, std::vector<int32_t>& goal,
                             std::vector<int32_t>& pop, std::vector<int32_t>& plen, std::vector<uint32_t>& poff,
                             std::vector<uint32_t>& enc) {
  const size_t words = enc.size();
  {
    int rc0 = pool_make_room(e, words + 1);
    if (rc0) return rc0;
  }
  for (int i = 0; i < n; i++) poff[i] += (uint32_t)e->pool_used;
  // start cells shared inside the batch must be appended to the cell list in spawn order
  std::vector<uint8_t> serial(n, 0);
  {
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return start[a] < start[b] || (start[a] == start[b] && a < b); });
    for (int q = 1; q < n; q++)
      if (start[order[q]] == start[order[q - 1]]) { serial[order[q]] = 1; serial[order[q - 1]] = 1; }
  }
  int rc = ensure_vehicle_capacity(e, e->n_vehicles_total + n, e->n_sched + n);
  if (rc) return rc;
  rc = ensure_pool(e, e->pool_used + words + 1);
  if (rc) return rc;
  if (n > e->cap_overflow) {
    rc = regrow(e, &e->d_overflow, 0, (size_t)n + 1);
    if (rc) return rc;
    e->cap_overflow = n;
  }
  hipStream_t st = e->stream;
  int32_t *ds = nullptr, *dg = nullptr, *dp = nullptr, *dl = nullptr;
  uint32_t* dof = nullptr;
  uint8_t* dser = nullptr;
  HIPOK(hipMalloc((void**)&ds, (size_t)n * 4)); HIPOK(hipMalloc((void**)&dg, (size_t)n * 4));
  HIPOK(hipMalloc((void**)&dp, (size_t)n * 4)); HIPOK(hipMalloc((void**)&dl, (size_t)n * 4));
  HIPOK(hipMalloc((void**)&dof, (size_t)n * 4)); HIPOK(hipMalloc((void**)&dser, (size_t)n));
  HIPOK(hipMemcpyAsync(ds, start.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPOK(hipMemcpyAsync(dg, goal.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPOK(hipMemcpyAsync(dp, pop.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPOK(hipMemcpyAsync(dl, plen.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPOK(hipMemcpyAsync(dof, poff.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPOK(hipMemcpyAsync(dser, serial.data(), (size_t)n, hipMemcpyHostToDevice, st));
  if (words) HIPOK(hipMemcpyAsync(e->d.pool + e->pool_used, enc.data(), words * 4, hipMemcpyHostToDevice, st));
  HIPOK(hipMemsetAsync(e->d_total, 0, sizeof(int), st));
  SpawnArgs a{ds, dg, dp, dl, dof, dser};
  hipLaunchKernelGGL(k_spawn, dim3(nblk(n)), dim3(BLK), 0, st, e->d, e->P, a, n, e->n_vehicles_total, e->n_active,
                     e->n_sched, e->C.elapsed, e->d_overflow, e->d_total);
  HIPOK(hipMemcpyAsync(e->hint, e->d_total, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPOK(hipStreamSynchronize(st));
  if (e->hint[0] > 0) hipLaunchKernelGGL(k_spawn_serial, dim3(1), dim3(64), 0, st, e->d, e->d_overflow, e->hint[0]);
  HIPOK(hipStreamSynchronize(st));
  (void)hipFree(ds); (void)hipFree(dg); (void)hipFree(dp); (void)hipFree(dl); (void)hipFree(dof); (void)hipFree(dser);
  // live_* counters (city_model.py:1910-1918)
  long long add_int = 0, add_thr = 0;
  for (int i = 0; i < n; i++) { add_int += pop[i] == TS_POP_INTERNAL; add_thr += pop[i] == TS_POP_THROUGH; }
  if (add_int || add_thr) {
    rc = sync_counters(e);
    if (rc) return rc;
    e->hcnt->live_internal += add_int; e->hcnt->live_through += add_thr;
    HIPOK(hipMemcpy(e->d.cnt, e->hcnt, sizeof(DevCnt), hipMemcpyHostToDevice));
  }
  e->pool_used += words;
  rc = pool_to_device(e);
  if (rc) return rc;
  e->n_vehicles_total += n; e->n_active += n; e->n_sched += n; e->n_sched_vehicles += n;
  return TS_OK;
}

// VehicleAgent.__init__ for one vehicle whose path the engine plans itself: place_vehicle, then
// self.path = self._compute_path() with city._path_cache (vehicle_base.py:78-81, 143-167)
static int add_vehicle_planned(ts_handle e, int start, int goal, int pop_type) {
  std::vector<int32_t> s1{start}, g1{goal}, p1{pop_type}, l1{0};
  std::vector<uint32_t> o1{0}, enc;
  int rc = add_vehicles_core(e, 1, s1, g1, p1, l1, o1, enc);
  if (rc) return rc;
  return plan_vehicle(e, e->n_vehicles_total - 1, start, goal);
}

// A search that runs alone gains nothing from a small scratch tier (tiers only buy concurrency), and one that outgrows
// its tier is run again from scratch: start where a search between these two cells very likely fits.
static int first_tier_for(ts_handle e, int start, int goal) {
  const int md = std::abs(start % e->W - goal % e->W) + std::abs(start / e->W - goal / e->W);
  return md < 48 ? 0 : md < 400 ? 1 : md < 2500 ? 2 : 3;
}

// self.path = self._compute_path() for a vehicle standing on `start` with target `goal` (both already on the
// device): city._path_cache first, then the phase 0-4 planner on the maps as they are now
static int plan_vehicle(ts_handle e, int vid, int start, int goal) {
  int rc;
  const uint64_t key = ((uint64_t)(uint32_t)start << 32) | (uint32_t)goal;
  Dev& d = e->d;
  if (e->P.pathfinding_cache) {
    auto it = e->path_cache.find(key);
    if (it != e->path_cache.end()) {
      const auto& cp = it->second;
      rc = pool_make_room(e, cp.words.size() + 1);
      if (rc) return rc;
      uint32_t off = (uint32_t)e->pool_used;
      int len = cp.len, zero = 0;
      if (!cp.words.empty()) HIPOK(hipMemcpy(d.pool + off, cp.words.data(), cp.words.size() * 4, hipMemcpyHostToDevice));
      HIPOK(hipMemcpy(d.path_off + vid, &off, 4, hipMemcpyHostToDevice));
      HIPOK(hipMemcpy(d.path_len + vid, &len, 4, hipMemcpyHostToDevice));
      HIPOK(hipMemcpy(d.path_cur + vid, &zero, 4, hipMemcpyHostToDevice));
      e->pool_used += cp.words.size();
      return pool_to_device(e);
    }
  }
  if (!e->density_valid) { rc = ensure_density(e, d.occ_snap); if (rc) return rc; e->density_valid = true; }
  for (int t = first_tier_for(e, start, goal); t < ts_engine::N_TIERS; t++) {
    rc = ensure_tier(e, t);
    if (rc) return rc;
    for (int attempt = 0; attempt < 3; attempt++) {
      hipLaunchKernelGGL(k_spawn_plan, dim3(1), dim3(64), 0, e->stream, d, e->P, e->tier[t], vid, e->d_status);
      HIPOK(hipMemcpyAsync(e->hint, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
      HIPOK(hipStreamSynchronize(e->stream));
      if (e->hint[0] != -2) break;
      rc = pool_make_room(e, (size_t)e->tier[t].cap + (1u << 16));  // pool full: make room and plan again
      if (rc) return rc;
    }
    if (e->hint[0] == -2) return fail(e, TS_E_CAPACITY, "path pool exhausted while planning a spawn");
    if (e->hint[0] >= 0) {
      const int len = e->hint[0];
      rc = pool_from_device(e);
      if (rc) return rc;
      uint16_t fl = 0;
      HIPOK(hipMemcpy(&fl, d.flags + vid, 2, hipMemcpyDeviceToHost));
      if (e->P.pathfinding_cache && len > 0 && !(fl & (VF_OVER | VF_DETOUR))) {
        ts_engine::CachedPath cp;
        cp.len = len;
        cp.words.resize((size_t)(len + 15) / 16);
        uint32_t off = 0;
        HIPOK(hipMemcpy(&off, d.path_off + vid, 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(cp.words.data(), d.pool + off, cp.words.size() * 4, hipMemcpyDeviceToHost));
        e->path_cache[key] = std::move(cp);
      }
      return TS_OK;
    }
  }
  return fail(e, TS_E_CAPACITY, "an A* search exceeded the largest scratch tier");
}

static int add_vehicles_any(ts_handle e, int32_t n, const int32_t* start_xy, const int32_t* goal_xy,
                            const int32_t* population_type, const int32_t* off32, const int32_t* path_xy,
                            const int64_t* off64, const uint8_t* path_dirs) {
  if (!e || n < 0 || (n > 0 && (!start_xy || !goal_xy))) return TS_E_INVALID;
  if (n == 0) return TS_OK;
  const int W = e->W, H = e->H;
  if (!(off32 && path_xy) && !(off64 && path_dirs)) {
    // the reference's own path: each vehicle is placed, then plans on the maps as they are at that moment
    for (int i = 0; i < n; i++) {
      int sx = start_xy[2 * i], sy = start_xy[2 * i + 1], gx = goal_xy[2 * i], gy = goal_xy[2 * i + 1];
      if (sx < 0 || sx >= W || sy < 0 || sy >= H || gx < 0 || gx >= W || gy < 0 || gy >= H)
        return fail(e, TS_E_INVALID, "vehicle start/goal out of bounds");
      if (sx == gx && sy == gy) return fail(e, TS_E_UNSUPPORTED, "start == goal (vehicle despawns inside the decide phase)");
      if ((long long)e->n_sched + 1 >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^22 agents");
      int rc = add_vehicle_planned(e, sy * W + sx, gy * W + gx, population_type ? population_type[i] : TS_POP_UNDEFINED);
      if (rc) return rc;
    }
    return TS_OK;
  }
  if ((long long)e->n_sched + n >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^22 agents");
  std::vector<int32_t> start(n), goal(n), pop(n), plen(n);
  std::vector<uint32_t> poff(n);
  auto O = [&](int i) -> long long { return off32 ? (long long)off32[i] : (long long)off64[i]; };
  size_t words = 0;
  for (int i = 0; i < n; i++) {
    long long len = O(i + 1) - O(i);
    if (len < 0 || len > 0x7FFFFFFF) return fail(e, TS_E_INVALID, "path_off must be non-decreasing");
    words += ((size_t)len + 15) / 16;
  }
  if (e->pool_used + words >= (1ull << 32)) return fail(e, TS_E_CAPACITY, "path pool exceeds 2^32 words");
  std::vector<uint32_t> enc(words, 0);
  size_t wpos = 0;
  for (int i = 0; i < n; i++) {
    int sx = start_xy[2 * i], sy = start_xy[2 * i + 1], gx = goal_xy[2 * i], gy = goal_xy[2 * i + 1];
    if (sx < 0 || sx >= W || sy < 0 || sy >= H || gx < 0 || gx >= W || gy < 0 || gy >= H)
      return fail(e, TS_E_INVALID, "vehicle start/goal out of bounds");
    if (sx == gx && sy == gy) return fail(e, TS_E_UNSUPPORTED, "start == goal (vehicle despawns inside the decide phase)");
    start[i] = sy * W + sx; goal[i] = gy * W + gx;
    pop[i] = population_type ? population_type[i] : TS_POP_UNDEFINED;
    const long long o = O(i);
    const int len = (int)(O(i + 1) - o);
    plen[i] = len;
    poff[i] = (uint32_t)wpos;
    int px = sx, py = sy;
    for (int k = 0; k < len; k++) {
      int dir;
      if (path_xy && off32) {
        int x = path_xy[2 * (o + k)], y = path_xy[2 * (o + k) + 1];
        int dx = x - px, dy = y - py;
        if (dx == 0 && dy == 1) dir = 0; else if (dx == 1 && dy == 0) dir = 1;
        else if (dx == 0 && dy == -1) dir = 2; else if (dx == -1 && dy == 0) dir = 3;
        else return fail(e, TS_E_INVALID, "explicit path is not a 4-adjacent chain");
        px = x; py = y;
      } else {
        dir = path_dirs[o + k];
        if (dir > 3) return fail(e, TS_E_INVALID, "direction code out of range");
        px += (dir == 1) - (dir == 3); py += (dir == 0) - (dir == 2);
      }
      if (px < 0 || px >= W || py < 0 || py >= H) return fail(e, TS_E_INVALID, "explicit path leaves the grid");
      enc[wpos + (k >> 4)] |= (uint32_t)dir << ((k & 15) * 2);
    }
    wpos += ((size_t)len + 15) / 16;
  }
  return add_vehicles_core(e, n, start, goal, pop, plen, poff, enc);
}

int ts_add_vehicles(ts_handle e, int32_t n, const int32_t* start_xy, const int32_t* goal_xy,
                    const int32_t* population_type, const int32_t* path_off, const int32_t* path_xy) {
  return add_vehicles_any(e, n, start_xy, goal_xy, population_type, path_off, path_xy, nullptr, nullptr);
}
int ts_add_vehicles_dirs(ts_handle e, int32_t n, const int32_t* start_xy, const int32_t* goal_xy,
                         const int32_t* population_type, const int64_t* path_off, const uint8_t* path_dirs) {
  return add_vehicles_any(e, n, start_xy, goal_xy, population_type, nullptr, nullptr, path_off, path_dirs);
}

int ts_upload_map(ts_handle e, int32_t which, const int8_t* src) {
  if (!e || !src) return TS_E_INVALID;
  int8_t* m = which == TS_MAP_STOP ? e->d.stop : which == TS_MAP_RAIN ? e->d.rain : nullptr;
  if (!m) return fail(e, TS_E_INVALID, "only stop_map and rain_map are host-writable");
  HIPOK(hipMemcpy(m, src, e->N, hipMemcpyHostToDevice));
  if (which == TS_MAP_STOP) {
    hipLaunchKernelGGL(k_plane_to_cells, dim3(nblk((long long)e->N)), dim3(BLK), 0, e->stream, e->d.cell, e->N, e->d.stop, 1);
    HIPOK(hipStreamSynchronize(e->stream));
  }
  return TS_OK;
}
int ts_debug_set_occupancy(ts_handle e, const int8_t* src) {
  if (!e || !src) return TS_E_INVALID;
  HIPOK(hipMemcpy(e->d.occ, src, e->N, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_plane_to_cells, dim3(nblk((long long)e->N)), dim3(BLK), 0, e->stream, e->d.cell, e->N, e->d.occ, 0);
  HIPOK(hipStreamSynchronize(e->stream));
  return TS_OK;
}

int ts_step(ts_handle e, int32_t n_ticks) {
  if (!e || n_ticks < 0) return TS_E_INVALID;
  if (!e->rng_global.seeded() || !e->rng_sched.seeded()) return fail(e, TS_E_STATE, "both RNG streams must be seeded before step");
  if (e->groups_scheduled != e->d.G && e->d.G > 0 && e->groups_scheduled != 0)
    return fail(e, TS_E_STATE, "every light group must be scheduled (or none)");
  if (e->fatal) return e->fatal;   // model.step() raised in the reference: the run is over
  for (int t = 0; t < n_ticks; t++) {
    int rc = tick(e);
    if (rc) return rc;
  }
  return TS_OK;
}

int ts_num_vehicles(ts_handle e) { return e ? e->n_active : TS_E_INVALID; }
int ts_num_groups(ts_handle e) { return e ? e->d.G : TS_E_INVALID; }
int ts_num_scheduled(ts_handle e) { return e ? e->n_sched : TS_E_INVALID; }

int ts_download_map(ts_handle e, int32_t which, int8_t* dst) {
  if (!e || !dst) return TS_E_INVALID;
  HIPOK(hipStreamSynchronize(e->stream));
  if (which == TS_MAP_STUCK) {   // lives only in the cell records
    int8_t* tmp = nullptr;
    HIPOK(hipMalloc((void**)&tmp, (size_t)e->N));
    hipLaunchKernelGGL(k_cells_to_plane, dim3(nblk((long long)e->N)), dim3(BLK), 0, e->stream, e->d.cell, e->N, tmp);
    hipError_t r = hipMemcpyAsync(dst, tmp, e->N, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    (void)hipFree(tmp);
    HIPOK(r);
    return TS_OK;
  }
  const int8_t* m = which == TS_MAP_OCCUPANCY ? e->d.occ : which == TS_MAP_STOP ? e->d.stop
                   : which == TS_MAP_RAIN ? e->d.rain : nullptr;
  if (!m) return TS_E_INVALID;
  HIPOK(hipMemcpy(dst, m, e->N, hipMemcpyDeviceToHost));
  return TS_OK;
}

int ts_download_density(ts_handle e, float* dst) {
  if (!e || !dst) return TS_E_INVALID;
  float *t0 = nullptr, *t1 = nullptr, *dn = nullptr;
  size_t N = e->N;
  HIPOK(hipMalloc((void**)&t0, N * 4)); HIPOK(hipMalloc((void**)&t1, N * 4)); HIPOK(hipMalloc((void**)&dn, N * 4));
  const int r = e->P.vehicle_awareness_range;
  hipLaunchKernelGGL(k_density_pass0, dim3(nblk(N)), dim3(BLK), 0, e->stream, e->d.occ, e->d.is_road, e->W, e->H, r, t0, t1);
  hipLaunchKernelGGL(k_density_pass1, dim3(nblk(N)), dim3(BLK), 0, e->stream, t0, t1, e->W, e->H, r, dn);
  HIPOK(hipMemcpyAsync(dst, dn, N * 4, hipMemcpyDeviceToHost, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  (void)hipFree(t0); (void)hipFree(t1); (void)hipFree(dn);
  return TS_OK;
}

int ts_download_vehicles(ts_handle e, int32_t* rows, int32_t cap_rows) {
  if (!e || !rows) return TS_E_INVALID;
  int n = e->n_active;
  if (n > cap_rows) return TS_E_CAPACITY;
  if (n == 0) return 0;
  int32_t* drows = nullptr;
  HIPOK(hipMalloc((void**)&drows, (size_t)n * TS_V_NFIELDS * 4));
  hipLaunchKernelGGL(k_rows, dim3(nblk(n)), dim3(BLK), 0, e->stream, e->d, n, e->d_crc, drows);
  HIPOK(hipMemcpyAsync(rows, drows, (size_t)n * TS_V_NFIELDS * 4, hipMemcpyDeviceToHost, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  (void)hipFree(drows);
  return n;
}

int ts_num_spawned(ts_handle e) { return e ? e->n_vehicles_total : TS_E_INVALID; }
int ts_download_vehicle_meta(ts_handle e, int32_t* rows, int32_t cap_rows) {
  if (!e || !rows) return TS_E_INVALID;
  const int n = e->n_active;
  if (n > cap_rows) return TS_E_CAPACITY;
  if (n == 0) return 0;
  int32_t* drows = nullptr;
  HIPOK(hipMalloc((void**)&drows, (size_t)n * TS_M_NFIELDS * 4));
  hipLaunchKernelGGL(k_meta_rows, dim3(nblk(n)), dim3(BLK), 0, e->stream, e->d, n, drows);
  HIPOK(hipMemcpyAsync(rows, drows, (size_t)n * TS_M_NFIELDS * 4, hipMemcpyDeviceToHost, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  (void)hipFree(drows);
  if (!e->svc.empty()) {
    std::unordered_map<int, int> type_of;
    for (const auto& v : e->svc) type_of[v.vid] = v.type;
    for (int i = 0; i < n; i++) {
      int32_t* r = rows + (size_t)i * TS_M_NFIELDS;
      if (r[TS_M_SERVICE_PHASE] >= 0) { auto it = type_of.find(r[TS_M_SPAWN_IDX]); if (it != type_of.end()) r[TS_M_VEHICLE_TYPE] = it->second; }
    }
  }
  return n;
}
int ts_download_service_vehicles(ts_handle e, int32_t* spawn_idx, double* loads, int32_t* block, int32_t cap) {
  if (!e || !spawn_idx || !loads || !block) return TS_E_INVALID;
  if ((int)e->svc.size() > cap) return TS_E_CAPACITY;
  std::vector<const ts_engine::SvcVeh*> order;
  for (const auto& v : e->svc) order.push_back(&v);
  std::sort(order.begin(), order.end(), [](const ts_engine::SvcVeh* a, const ts_engine::SvcVeh* b) { return a->vid < b->vid; });
  int n = 0;
  for (const auto* v : order) {
    spawn_idx[n] = v->vid; loads[2 * n] = v->load; loads[2 * n + 1] = v->max_load; block[n] = v->block;
    n++;
  }
  return n;
}
int ts_download_path(ts_handle e, int32_t active_pos, int32_t* xy, int32_t cap_cells) {
  if (!e || active_pos < 0 || active_pos >= e->n_active) return TS_E_INVALID;
  int vid, cur, len;
  HIPOK(hipMemcpy(&vid, e->d.active + active_pos, 4, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(&cur, e->d.path_cur + vid, 4, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(&len, e->d.path_len + vid, 4, hipMemcpyDeviceToHost));
  int n = len - cur;
  if (!xy) return n;
  if (n > cap_cells) return TS_E_CAPACITY;
  if (n == 0) return 0;
  int32_t* dxy = nullptr;
  HIPOK(hipMalloc((void**)&dxy, (size_t)n * 8));
  hipLaunchKernelGGL(k_path_cells, dim3(1), dim3(64), 0, e->stream, e->d, vid, dxy);
  HIPOK(hipMemcpyAsync(xy, dxy, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  (void)hipFree(dxy);
  return n;
}

int ts_download_groups(ts_handle e, int32_t* rows) {
  if (!e || !rows) return TS_E_INVALID;
  int G = e->d.G;
  if (G == 0) return 0;
  int32_t* drows = nullptr;
  HIPOK(hipMalloc((void**)&drows, (size_t)G * TS_G_NFIELDS * 4));
  hipLaunchKernelGGL(k_group_rows, dim3(nblk(G)), dim3(BLK), 0, e->stream, e->d, drows);
  HIPOK(hipMemcpyAsync(rows, drows, (size_t)G * TS_G_NFIELDS * 4, hipMemcpyDeviceToHost, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  (void)hipFree(drows);
  return G;
}

int ts_num_blocks(ts_handle e) { return e ? (int)e->blocks.size() : TS_E_INVALID; }
int ts_download_blocks(ts_handle e, double* rows) {
  if (!e || !rows) return TS_E_INVALID;
  for (size_t b = 0; b < e->blocks.size(); b++) { rows[2 * b] = e->blocks[b].food; rows[2 * b + 1] = e->blocks[b].waste; }
  return TS_OK;
}
int ts_group_links(ts_handle e, int32_t group, int32_t repopulate) {
  if (!e || group < 0 || group >= e->d.G) return TS_E_INVALID;
  int v = 0;
  if (repopulate) { v = 1; HIPOK(hipMemcpy(e->d.gs_repop + group, &v, 4, hipMemcpyHostToDevice)); return 1; }
  HIPOK(hipStreamSynchronize(e->stream));
  HIPOK(hipMemcpy(&v, e->d.gs_repop + group, 4, hipMemcpyDeviceToHost));
  return v ? 1 : 0;
}
int ts_add_service_vehicle(ts_handle e, int32_t x, int32_t y, int32_t service_type) {
  if (!e || x < 0 || x >= e->W || y < 0 || y >= e->H) return TS_E_INVALID;
  if (service_type != TS_TRIP_SERVICE_FOOD && service_type != TS_TRIP_SERVICE_WASTE) return fail(e, TS_E_INVALID, "service_type");
  if (e->blocks.empty()) return fail(e, TS_E_STATE, "service vehicles need the block tables (ts_set_traffic_generator)");
  if (e->fatal) return e->fatal;
  return spawn_service_at(e, y * e->W + x, service_type, -1);
}
int ts_rain_info(ts_handle e, TsRainInfo* out) {
  if (!e || !out) return TS_E_INVALID;
  out->has_manager = e->rain_manager; out->n_rains = (int32_t)e->rains.size();
  out->cooldown = e->rain_cooldown_left; out->counter = e->rain_counter;
  return TS_OK;
}
int ts_rain_spawn(ts_handle e) {
  if (!e) return TS_E_INVALID;
  if (!e->rain_manager) return fail(e, TS_E_STATE, "no RainManager is scheduled");
  if (!e->rng_global.seeded()) return fail(e, TS_E_STATE, "seed the global stream first");
  HIPOK(hipStreamSynchronize(e->stream));
  return rain_add_random(e);
}
int ts_counters(ts_handle e, TsCounters* out) {
  if (!e || !out) return TS_E_INVALID;
  int rc = sync_counters(e);
  if (rc) return rc;
  *out = e->C;
  return TS_OK;
}

int ts_set_device(int32_t device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return TS_E_DEVICE;
  g_device = device;
  return hipSetDevice(device) == hipSuccess ? TS_OK : TS_E_DEVICE;
}
int ts_profile_enable(ts_handle e, int32_t on) {
  if (!e) return TS_E_INVALID;
  e->prof = on != 0;
  if (on) { memset(e->prof_ms, 0, sizeof(e->prof_ms)); memset(e->prof_launches, 0, sizeof(e->prof_launches));
            memset(e->prof_items, 0, sizeof(e->prof_items)); }
  return TS_OK;
}
int ts_profile_count(void) { return PK_COUNT; }
const char* ts_profile_name(int32_t id) { return id >= 0 && id < PK_COUNT ? PK_NAMES[id] : ""; }
int ts_profile_get(ts_handle e, int32_t id, double* total_ms, int64_t* launches, int64_t* items) {
  if (!e || id < 0 || id >= PK_COUNT) return TS_E_INVALID;
  if (total_ms) *total_ms = e->prof_ms[id];
  if (launches) *launches = e->prof_launches[id];
  if (items) *items = e->prof_items[id];
  return TS_OK;
}

int ts_astar(ts_handle e, int32_t sx, int32_t sy, int32_t gx, int32_t gy, int32_t soft, int32_t ignore_flow,
             int32_t maximum_steps, int32_t* out_xy, int32_t cap_cells) {
  if (!e) return TS_E_INVALID;
  if (sx < 0 || sx >= e->W || sy < 0 || sy >= e->H || gx < 0 || gx >= e->W || gy < 0 || gy >= e->H)
    return fail(e, TS_E_INVALID, "astar endpoints out of bounds");
  int rc = ensure_density(e, e->d.occ);  // "evaluated on the engine's current maps"
  if (rc) return rc;
  e->density_valid = false;
  for (int t = first_tier_for(e, sy * e->W + sx, gy * e->W + gx); t < ts_engine::N_TIERS; t++) {
    rc = ensure_tier(e, t);
    if (rc) return rc;
    hipLaunchKernelGGL(k_astar_single, dim3(1), dim3(64), 0, e->stream, e->d, e->P, e->tier[t], sy * e->W + sx,
                       gy * e->W + gx, soft, ignore_flow, maximum_steps, e->d_status);
    HIPOK(hipMemcpyAsync(e->hint, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPOK(hipStreamSynchronize(e->stream));
    int len = e->hint[0];
    if (len < 0) continue;
    if (len > cap_cells) return TS_E_CAPACITY;
    if (len > 0) {
      std::vector<int32_t> cells(len);
      HIPOK(hipMemcpy(cells.data(), e->tier[t].cells, (size_t)len * 4, hipMemcpyDeviceToHost));
      for (int k = 0; k < len; k++) { out_xy[2 * k] = cells[k] % e->W; out_xy[2 * k + 1] = cells[k] / e->W; }
    }
    return len;
  }
  return fail(e, TS_E_CAPACITY, "an A* search exceeded the largest scratch tier");
}

}  // extern "C"
