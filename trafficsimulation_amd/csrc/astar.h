// astar.h - GPU A* (one search per wave) and the replanning policy of VehicleAgent.
//
// astar_dev (one lane) and astar_wave (the same search spread over the 64 lanes of a wave; the one the replanning
// kernels use) restate astar_numba.py:87-239 verbatim, quirks included:
//   * binary heap keyed on f only, strict '<' in both sift routines (52-85);
//   * dir_arr lives in heap-SLOT order and is NOT swapped by the sifts, so prev_dir = dir_arr[0] is a
//     stale slot value (139, 147, 235);
//   * `ng` is a double (R1 penalty 0.5) that truncates when stored into the int32 arrays (226-232);
//   * soft-obstacle penalty int(1000 * (1 + 4 * density)) in double arithmetic on the float32 density.
// Instead of the reference's O(W*H) per-call initialisation (119-122) every searcher owns an
// epoch-stamped open-addressing table in HBM (dist / came_from by cell) and a private heap; a search
// that outgrows its tier reports overflow and is re-run, unchanged, on a larger tier.
//
// decide_vehicle restates step_decide (vehicle_base.py:616-663) with _recompute_path_on_stuck 506-517,
// _recompute_path_on_obstacle 454-504, _compute_path 143-167 and _compute_path_internal 199-420
// (phases 0-4).  It is a pure function of the tick-start state until its final commit, so the same code
// runs in k_decide_main (one vehicle per lane, no scratch: bails out as soon as a search is needed) and in
// k_decide_replan (one vehicle per wave, every lane executing the same code on the same values).
#pragma once
#include "dev.h"

namespace {

constexpr int A_INF = 0x3F3F3F3F;
constexpr int MAXB = 64;  // longest contraflow bypass (VEHICLE_MAX_CONTRAFLOW_*_STEPS <= 64)
enum { DV_DONE = 0, DV_DEFER = 1, DV_OVERFLOW = 2, DV_POOL_FULL = 3 };

// 16-byte records so that one probe / one heap level is one dwordx4 access
struct __attribute__((aligned(16))) HEnt { int32_t key, dist, came; uint32_t stamp; };   // dist / came_from by cell
struct __attribute__((aligned(16))) QEnt { int32_t f, g, s, i; };                        // heap entry (f_arr, g_arr, s_arr, i_arr)

// One searcher's scratch (all in HBM, carved from a tier arena).
struct AScratch {
  HEnt* ht;
  uint32_t hmask;
  QEnt* hq;
  int8_t* hd;   // dir_arr: indexed by heap SLOT and deliberately not moved by the sift routines
  int heap_cap;
  int32_t *A, *P, *T, *PO, *PD;  // cap cells each: A* result, current new path, splice target, staged pre-paths
  int32_t *BYP, *OV, *DV;        // MAXB cells each: bypass result, staged overtake / detour paths
  int cap;        // capacity of the cell buffers
  int node_cap;   // distinct cells the table may hold
  uint32_t epoch;
  int nodes;
  int peak_nodes;   // largest table population any search of this vehicle reached (for the tier hint)
  long long calls, expansions, relaxations;
};
struct RLists { int32_t* l[6]; };       // replan work lists, see run_replans (engine.hip)
struct StageCaps { int nodes[5]; };     // node capacity of each replanning stage
// work list / counter a vehicle waiting for stage h is queued on
__device__ __forceinline__ int stage_list(int h) { return h == 0 ? 0 : h == 1 ? 4 : h == 2 ? 1 : h == 3 ? 2 : 5; }
__device__ __forceinline__ int stage_counter(int h) { return h == 0 ? 0 : h == 1 ? 5 : h == 2 ? 1 : h == 3 ? 2 : 6; }
__device__ __forceinline__ void note_tier(const Dev& d, const AScratch& S, int vid, int stage, const StageCaps& caps) {
  if (S.calls == 0) return;
  int h = stage;
  while (h > 0 && 2 * S.peak_nodes <= caps.nodes[h - 1]) h--;
  d.tier_hint[vid] = (uint8_t)h;
}

struct ATier {
  int cap, n_slots, heap_cap;
  uint32_t hsize;  // power of two >= 2 * cap
  HEnt* ht;
  QEnt* hq;
  int8_t* hd;
  int32_t* cells;      // per slot: 5 * cap + 3 * MAXB
  uint32_t* slot_epoch;
};

__device__ __forceinline__ void scratch_bind(const ATier& t, int slot, AScratch& S) {
  S.ht = t.ht + (size_t)slot * t.hsize;
  S.hmask = t.hsize - 1;
  S.hq = t.hq + (size_t)slot * t.heap_cap;
  S.hd = t.hd + (size_t)slot * t.heap_cap;
  S.heap_cap = t.heap_cap;
  int32_t* c = t.cells + (size_t)slot * ((size_t)5 * t.cap + 3 * MAXB);
  S.A = c; S.P = c + t.cap; S.T = c + 2 * (size_t)t.cap; S.PO = c + 3 * (size_t)t.cap; S.PD = c + 4 * (size_t)t.cap;
  S.BYP = c + 5 * (size_t)t.cap; S.OV = S.BYP + MAXB; S.DV = S.OV + MAXB;
  S.cap = t.cap;
  S.node_cap = t.cap;
  S.epoch = t.slot_epoch[slot];
  S.peak_nodes = 0;
  S.nodes = 0; S.calls = 0; S.expansions = 0; S.relaxations = 0;
}

__device__ __forceinline__ uint32_t h_hash(int cell, uint32_t mask) { return ((uint32_t)cell * 2654435761u >> 7) & mask; }

// returns the slot of `cell` (found) or the empty slot where it would go; `ent` = the record read there
__device__ __forceinline__ uint32_t h_probe(const AScratch& S, int cell, bool& found, HEnt& ent) {
  uint32_t h = h_hash(cell, S.hmask);
  for (;;) {
    ent = S.ht[h];
    if (ent.stamp != S.epoch) { found = false; return h; }
    if (ent.key == cell) { found = true; return h; }
    h = (h + 1) & S.hmask;
  }
}

// astar_core.  Writes the path (start excluded, goal included) to out[0..len); returns len >= 0, or -1 on
// tier overflow (table, heap or output capacity).
__device__ int astar_dev(const Dev& d, const TsParams& P, AScratch& S, int start_idx, int goal_idx, bool soft,
                         bool ignore_flow, int maximum_steps, int32_t* out, int out_cap) {
  const int W = d.W, H = d.H;
  S.calls++;
  S.epoch++;
  if (S.epoch == 0) {  // stamp wrapped (once per 2^32 searches): clear the table
    for (uint32_t q = 0; q <= S.hmask; q++) S.ht[q].stamp = 0;
    S.epoch = 1;
  }
  S.nodes = 0;
  const int gx = goal_idx % W, gy = goal_idx / W;
  {
    bool f; HEnt e;
    uint32_t h = h_probe(S, start_idx, f, e);
    S.ht[h] = HEnt{start_idx, 0, -1, S.epoch};
    S.nodes = 1;
  }
  int heap_size = 1;
  {
    int sx = start_idx % W, sy = start_idx / W;
    S.hq[0] = QEnt{abs(sx - gx) + abs(sy - gy), 0, 0, start_idx};
    S.hd[0] = -1;
  }
  while (heap_size > 0) {
    const QEnt top = S.hq[0];
    const int g = top.g, steps = top.s, cur = top.i;
    const int prev_dir = S.hd[0];
    heap_size--;
    if (heap_size > 0) {
      // replace the root with the last entry and sift down (strict '<' on f, left child first)
      QEnt x = S.hq[heap_size];
      S.hd[0] = S.hd[heap_size];
      int idx = 0;
      for (;;) {
        int left = 2 * idx + 1, right = left + 1;
        if (left >= heap_size) break;
        QEnt l = S.hq[left];
        int smallest = idx;
        int fs = x.f;
        QEnt c = x;
        if (l.f < fs) { smallest = left; fs = l.f; c = l; }
        if (right < heap_size) {
          QEnt r = S.hq[right];
          if (r.f < fs) { smallest = right; c = r; }
        }
        if (smallest == idx) break;
        S.hq[idx] = c;     // the child moves up; x keeps sinking
        idx = smallest;
      }
      S.hq[idx] = x;
    }
    if (cur == goal_idx) {
      int len = 0;
      for (int idx = cur; idx != start_idx;) {
        bool f; HEnt e;
        h_probe(S, idx, f, e);
        idx = e.came;
        len++;
      }
      if (len > out_cap) return -1;
      int k = len;
      for (int idx = cur; idx != start_idx;) {
        out[--k] = idx;
        bool f; HEnt e;
        h_probe(S, idx, f, e);
        idx = e.came;
      }
      return len;
    }
    {
      bool f; HEnt e;
      h_probe(S, cur, f, e);
      if (g > (f ? e.dist : A_INF)) continue;
    }
    S.expansions++;
    const int cx = cur % W, cy = cur / W;
    const uint8_t bits = (uint8_t)st_allowed(d.cell[cur].stat);
    for (int dd = 0; dd < 4; dd++) {
      const int nx = cx + (dd == 1) - (dd == 3), ny = cy + (dd == 0) - (dd == 2);
      if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
      const int ns = steps + 1;
      if (ns > maximum_steps) continue;
      const int nidx = ny * W + nx;
      double ng = g + 1;
      if (P.turn_penalty_enabled && prev_dir != -1 && dd != prev_dir) ng += P.turn_penalty;
      const Cell nc = d.cell[nidx];   // one 32-byte load: occupancy, stop and the static byte of the neighbour
      if ((bits & (1 << dd)) == 0) {
        if (ignore_flow && st_is_road(nc.stat) == 1) ng += P.contraflow_penalty;
        else continue;
      }
      if (nc.occ == 1) {
        if (soft && P.dynamic_penalties_enabled) {
          double p = P.obstacle_penalty_vehicle;
          double local_density = (double)d.density[nidx];
          ng += (double)(long long)(p * (1.0 + P.dynamic_penalty_scale * local_density));
        } else if (soft) ng += P.obstacle_penalty_vehicle;
        else continue;
      }
      if (nc.stop == 1) {
        if (soft) ng += P.obstacle_penalty_stop;
        else continue;
      }
      if (P.road_type_penalties_enabled && st_is_road(nc.stat) == 1) {
        int rt = st_road_type(nc.stat);
        if (rt == 1) ng += P.road_type_penalty_r1;
        else if (rt == 2) ng += P.road_type_penalty_r2;
        else if (rt == 3) ng += P.road_type_penalty_r3;
      }
      bool found; HEnt e;
      uint32_t h = h_probe(S, nidx, found, e);
      if (ng < (double)(found ? e.dist : A_INF)) {
        S.relaxations++;
        if (!found) {
          if (S.nodes >= S.node_cap) return -1;
          S.nodes++;
          if (S.nodes > S.peak_nodes) S.peak_nodes = S.nodes;
        }
        S.ht[h] = HEnt{nidx, (int)ng, cur, S.epoch};
        if (heap_size >= S.heap_cap) return -1;
        // heap push at slot heap_size + sift up; dir_arr[slot] is written once and stays with the SLOT
        QEnt x{(int)(ng + (double)(abs(nx - gx) + abs(ny - gy))), (int)ng, ns, nidx};
        int i = heap_size;
        S.hd[i] = (int8_t)dd;
        while (i > 0) {
          int parent = (i - 1) / 2;
          QEnt pe = S.hq[parent];
          if (x.f < pe.f) { S.hq[i] = pe; i = parent; } else break;
        }
        S.hq[i] = x;
        heap_size++;
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// The same search spread over one wavefront.  All 64 lanes call it with identical arguments and get the same
// return value; the scratch (table, heap, dir bytes, output) is the searcher's, as for astar_dev.  The algorithm is
// the sequential one - same heap layout, same comparisons, same order of relaxations - only its memory traffic
// is organised by lanes so that a step costs one round trip instead of one per access:
//   * sift-down: the 62 entries of the next five levels below the hole are fetched at once (one per lane) and the
//     walk down those levels reads them with cross-lane shuffles;
//   * sift-up of a push: the (at most 31) ancestors of the new slot are fetched at once, a ballot finds how far
//     the entry rises, the lanes holding ancestors write them one level down in parallel;
//   * the four neighbours are prepared on lanes 0-3 (cell record, density, table probe), then committed in the
//     reference's order N, E, S, W.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ void wave_mem_sync() { __builtin_amdgcn_wave_barrier(); }
// values every lane holds alike: tell the compiler (scalar registers, scalar branches) / read one lane's copy
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double rl(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ int astar_wave(const Dev& d, const TsParams& P, AScratch& S, int start_idx, int goal_idx, bool soft,
                          bool ignore_flow, int maximum_steps, int32_t* out, int out_cap) {
  const int W = d.W, H = d.H;
  const int lane = lane_id();
  S.calls++;
  S.epoch++;
  if (S.epoch == 0) {
    for (uint32_t q = lane; q <= S.hmask; q += 64) S.ht[q].stamp = 0;
    S.epoch = 1;
    wave_mem_sync();
  }
  S.nodes = 0;
  int gx, gy;
  cell_xy(d, goal_idx, gx, gy);
  {
    bool f; HEnt e;
    uint32_t h = h_probe(S, start_idx, f, e);
    if (lane == 0) S.ht[h] = HEnt{start_idx, 0, -1, S.epoch};
    S.nodes = 1;
    if (S.nodes > S.peak_nodes) S.peak_nodes = S.nodes;
  }
  int heap_size = 1;
  {
    int sx, sy;
    cell_xy(d, start_idx, sx, sy);
    if (lane == 0) { S.hq[0] = QEnt{abs(sx - gx) + abs(sy - gy), 0, 0, start_idx}; S.hd[0] = -1; }
  }
  wave_mem_sync();
  // relative position of this lane inside a 63-entry window hanging below a hole: level and offset in the level
  const int rel = lane;                                  // 0 = the hole itself (unused), 1..62 = five levels below it
  const int rlvl = 31 - __builtin_clz((unsigned)(rel + 1));   // 0 for rel 0, 1 for 1-2, ... 5 for 31-62
  const int roff = (rel + 1) - (1 << rlvl);
  while (heap_size > 0) {
    const QEnt top = S.hq[0];
    const int g = uni(top.g), steps = uni(top.s), cur = uni(top.i);
    const int prev_dir = uni((int)S.hd[0]);
    heap_size--;
    // Everything the expansion of `cur` will read besides the heap is requested now, so that it travels while the
    // sift-down below works on the heap (which it does not touch): the cell record of `cur`, and on lanes 0-3 the
    // record, density and first table probe of neighbour `lane`; on every lane the first probe of `cur` itself.
    int cx, cy;
    cell_xy(d, cur, cx, cy);
    const int dd_l = lane & 3;
    const int nx_l = cx + (dd_l == 1) - (dd_l == 3), ny_l = cy + (dd_l == 0) - (dd_l == 2);
    const bool inb_l = lane < 4 && nx_l >= 0 && nx_l < W && ny_l >= 0 && ny_l < H;
    const int nidx_l = inb_l ? ny_l * W + nx_l : cur;
    const uint32_t cur_dw = *reinterpret_cast<const uint32_t*>(&d.cell[cur].occ);
    const uint32_t dyn_l = *reinterpret_cast<const uint32_t*>(&d.cell[nidx_l].occ);
    const float dens_l = soft && P.dynamic_penalties_enabled ? d.density[nidx_l] : 0.f;
    const uint32_t slot_l = h_hash(nidx_l, S.hmask), slot_c = h_hash(cur, S.hmask);
    const HEnt first_l = S.ht[slot_l], first_c = S.ht[slot_c];
    if (heap_size > 0) {
      QEnt x = S.hq[heap_size];
      x.f = uni(x.f); x.g = uni(x.g); x.s = uni(x.s); x.i = uni(x.i);
      const int8_t xd = (int8_t)uni((int)S.hd[heap_size]);
      wave_mem_sync();                     // every lane has read hd[0] / hq[0] before they are overwritten
      if (lane == 0) S.hd[0] = xd;
      long long idx = 0;                   // the hole; x keeps sinking
      bool placed = false;
      while (!placed) {
        // fetch the window below the hole
        const long long abs_l = ((idx + 1) << rlvl) + roff - 1;
        const bool in_heap = lane >= 1 && lane <= 62 && abs_l < heap_size;
        QEnt mine = in_heap ? S.hq[abs_l] : QEnt{0x7FFFFFFF, 0, 0, 0};
        int cur_rel = 0;
        long long cur_abs = idx;
#pragma unroll
        for (int lv = 0; lv < 5; lv++) {
          const int l_rel = 2 * cur_rel + 1, r_rel = l_rel + 1;
          const long long l_abs = 2 * cur_abs + 1, r_abs = l_abs + 1;
          if (l_abs >= heap_size) { if (lane == 0) S.hq[cur_abs] = x; placed = true; break; }
          const int lf = rl(mine.f, l_rel), rf = rl(mine.f, r_rel);
          int smallest = cur_rel, fs = x.f;
          if (lf < fs) { smallest = l_rel; fs = lf; }
          if (r_abs < heap_size && rf < fs) smallest = r_rel;
          if (smallest == cur_rel) { if (lane == 0) S.hq[cur_abs] = x; placed = true; break; }
          if (lane == smallest) S.hq[cur_abs] = mine;     // the child moves up
          cur_abs = smallest == l_rel ? l_abs : r_abs;
          cur_rel = smallest;
        }
        if (!placed) idx = cur_abs;       // five levels down and still sinking: next window
      }
    }
    wave_mem_sync();
    if (cur == goal_idx) {
      int len = 0;
      for (int idx = cur; idx != start_idx;) {
        bool f; HEnt e;
        h_probe(S, idx, f, e);
        idx = e.came;
        len++;
      }
      if (len > out_cap) return -1;
      int k = len;
      for (int idx = cur; idx != start_idx;) {
        --k;
        if (lane == 0) out[k] = idx;
        bool f; HEnt e;
        h_probe(S, idx, f, e);
        idx = e.came;
      }
      wave_mem_sync();
      return len;
    }
    {
      bool f; HEnt e = first_c;
      uint32_t h = slot_c;
      for (;;) {   // continue the probe of `cur` from the record fetched above
        if (e.stamp != S.epoch) { f = false; break; }
        if (e.key == cur) { f = true; break; }
        h = (h + 1) & S.hmask;
        e = S.ht[h];
      }
      if (uni((int)(g > (f ? e.dist : A_INF)))) continue;
    }
    S.expansions++;
    const uint8_t bits = (uint8_t)st_allowed((uint8_t)(cur_dw >> 24));
    // ---- prepare: lane dd < 4 evaluates neighbour dd from what was fetched before the sift-down ---------------
    bool ok_l = inb_l && steps + 1 <= maximum_steps;
    double ng_l = g + 1;
    bool found_l = false;
    HEnt e_l = first_l;
    uint32_t h_l = slot_l;
    if (ok_l) {
      const int n_occ = (int8_t)(dyn_l & 0xFF), n_stop = (int8_t)((dyn_l >> 8) & 0xFF);
      const uint8_t n_stat = (uint8_t)(dyn_l >> 24);
      if (P.turn_penalty_enabled && prev_dir != -1 && dd_l != prev_dir) ng_l += P.turn_penalty;
      if ((bits & (1 << dd_l)) == 0) {
        if (ignore_flow && st_is_road(n_stat) == 1) ng_l += P.contraflow_penalty;
        else ok_l = false;
      }
      if (ok_l && n_occ == 1) {
        if (soft && P.dynamic_penalties_enabled) {
          double p = P.obstacle_penalty_vehicle;
          double local_density = (double)dens_l;
          ng_l += (double)(long long)(p * (1.0 + P.dynamic_penalty_scale * local_density));
        } else if (soft) ng_l += P.obstacle_penalty_vehicle;
        else ok_l = false;
      }
      if (ok_l && n_stop == 1) {
        if (soft) ng_l += P.obstacle_penalty_stop;
        else ok_l = false;
      }
      if (ok_l && P.road_type_penalties_enabled && st_is_road(n_stat) == 1) {
        int rt = st_road_type(n_stat);
        if (rt == 1) ng_l += P.road_type_penalty_r1;
        else if (rt == 2) ng_l += P.road_type_penalty_r2;
        else if (rt == 3) ng_l += P.road_type_penalty_r3;
      }
      if (ok_l) {   // continue the probe from the record fetched above
        for (;;) {
          if (e_l.stamp != S.epoch) { found_l = false; break; }
          if (e_l.key == nidx_l) { found_l = true; break; }
          h_l = (h_l + 1) & S.hmask;
          e_l = S.ht[h_l];
        }
      }
    }
    // ---- commit in the reference's order --------------------------------------------------------------------
    bool table_grew = false;
#pragma unroll
    for (int dd = 0; dd < 4; dd++) {
      if (!rl((int)ok_l, dd)) continue;
      const int nidx = rl(nidx_l, dd);
      const double ng = rl(ng_l, dd);
      bool found = rl((int)found_l, dd) != 0;
      uint32_t h = (uint32_t)rl((int)h_l, dd);
      int dist_n = rl(e_l.dist, dd);
      if (table_grew) {   // a key went in since the probe: it may sit where this one would have gone
        HEnt e;
        h = (uint32_t)uni((int)h_probe(S, nidx, found, e));
        found = uni((int)found) != 0;
        dist_n = uni(e.dist);
      }
      if (!(ng < (double)(found ? dist_n : A_INF))) continue;
      S.relaxations++;
      if (!found) {
        if (S.nodes >= S.node_cap) return -1;
        S.nodes++;
        if (S.nodes > S.peak_nodes) S.peak_nodes = S.nodes;
        table_grew = true;
      }
      if (heap_size >= S.heap_cap) return -1;
      const int nx = rl(nx_l, dd), ny = rl(ny_l, dd);
      const QEnt x{(int)(ng + (double)(abs(nx - gx) + abs(ny - gy))), (int)ng, steps + 1, nidx};
      const long long i = heap_size;
      // ancestors of slot i: a_k = ((i + 1) >> k) - 1, k = 1 .. depth; lane k - 1 fetches a_k
      const int depth = 63 - __builtin_clzll((unsigned long long)(i + 1));
      const long long a_mine = ((i + 1) >> (lane + 1)) - 1;
      const bool has = lane < depth;
      const QEnt anc = has ? S.hq[a_mine] : QEnt{(int)0x80000000, 0, 0, 0};
      const unsigned long long rises = __ballot(has && x.f < anc.f);
      const int r = rises == ~0ull ? 64 : __builtin_ctzll(~rises);      // leading ancestors the entry passes
      if (lane == 0) { S.ht[h] = HEnt{nidx, (int)ng, cur, S.epoch}; S.hd[i] = (int8_t)dd; }
      if (lane < r) S.hq[((i + 1) >> lane) - 1] = anc;                 // ancestor k moves to where k - 1 was
      if (lane == 0) S.hq[((i + 1) >> r) - 1] = x;
      heap_size++;
      wave_mem_sync();
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// vehicle working state for one step_decide
// ---------------------------------------------------------------------------------------------
struct VW {
  int vid, i, pos, target;
  uint16_t f;
  int base, cur, cooldown, over_dur, det_dur, stuck_ticks;
  // current path: pool-resident direction string, or cells in S.P
  bool newpath;
  int plen, pcur;
  uint32_t off;
  // aux paths k: 0 overtake_path, 1 pre_overtake_path, 2 stuck_detour_path, 3 pre_stuck_detour_path
  bool ax_staged[4];
  int ax_len[4];  // -1 = None
  int d_overtaking, d_detour;
  bool reach_known;  // d.reach[vid] was computed for this tick's maps and this position
};

__device__ __forceinline__ int32_t* ax_buf(const AScratch& S, int k) { return k == 0 ? S.OV : k == 1 ? S.PO : k == 2 ? S.DV : S.PD; }

// sequential reader over an aux path (staged cells or pool-resident directions)
struct AxReader {
  const int32_t* cells;
  const uint32_t* pool;
  uint32_t off;
  int cell, idx, W;
  __device__ void init(const Dev& d, const AScratch& S, const VW& v, int k) {
    idx = 0; W = d.W; pool = d.pool;
    if (v.ax_staged[k]) { cells = ax_buf(S, k); }
    else { cells = nullptr; off = d.ax_off[k][v.vid]; cell = d.ax_start[k][v.vid]; }
  }
  __device__ __forceinline__ int next() {
    if (cells) return cells[idx++];
    cell = step_cell(cell, path_dir(pool, off, idx++), W);
    return cell;
  }
};

__device__ __forceinline__ int vw_path_cell(const Dev& d, const AScratch* S, const VW& v, int k, int& walk) {
  // k-th remaining cell; `walk` carries the running cell for pool-resident paths (call with k ascending)
  if (v.newpath) return S->P[k];
  walk = step_cell(walk, path_dir(d.pool, v.off, v.pcur + k), d.W);
  return walk;
}

// _scan_ahead_for_obstacles (vehicle_base.py:422-452).  The cells are decoded first and their records loaded
// together (independent loads, one memory round trip); the reference's early exit at index 0 only shortens the
// evaluation.
constexpr int SCAN_MAX = 16;
__device__ void scan_ahead_dev(const Dev& d, const TsParams& P, const AScratch* S, const VW& v, int& idx_stop,
                               int& idx_veh, int& first_cell) {
  idx_stop = -1; idx_veh = -1; first_cell = -1;
  const int look = min(P.vehicle_awareness_range, v.plen);
  int walk = v.pos;
  if (look <= SCAN_MAX && !v.newpath) {
    int cells[SCAN_MAX];
    uint32_t dyn[SCAN_MAX];   // the dword holding occ / stop / stuck / stat
    // the next 16 steps are at most 32 bits of the direction string: two pool words, decoded in registers
    const uint32_t wi = (uint32_t)v.pcur >> 4, nwords = ((uint32_t)(v.pcur + v.plen) + 15u) >> 4;
    uint64_t bits = d.pool[v.off + wi];
    if (wi + 1 < nwords) bits |= (uint64_t)d.pool[v.off + wi + 1] << 32;
    bits >>= (v.pcur & 15) * 2;
    {
      int c = v.pos;
#pragma unroll
      for (int k = 0; k < SCAN_MAX; k++) {
        if (k < look) c = step_cell(c, (int)((bits >> (2 * k)) & 3), d.W);
        cells[k] = c;
      }
    }
    if (look > 0) first_cell = cells[0];
    // an obstacle at index 0 ends the scan (the reference's break can only fire there) - and in dense traffic that
    // is the common case, so the first record is fetched alone and the other nine only if it is clear
    dyn[0] = look > 0 ? *reinterpret_cast<const uint32_t*>(&d.cell[cells[0]].occ) : 0u;
    if (look > 0 && (int8_t)((dyn[0] >> 8) & 0xFF) == 1) idx_stop = 0;
    if (look > 0 && (int8_t)(dyn[0] & 0xFF) == 1) idx_veh = 0;
    if (idx_stop == 0 || idx_veh == 0) return;
#pragma unroll
    for (int k = 1; k < SCAN_MAX; k++) dyn[k] = k < look ? *reinterpret_cast<const uint32_t*>(&d.cell[cells[k]].occ) : 0u;
#pragma unroll
    for (int k = 1; k < SCAN_MAX; k++) {
      const bool in = k < look;
      const int occ = (int8_t)(dyn[k] & 0xFF), stop = (int8_t)((dyn[k] >> 8) & 0xFF);
      if (in && idx_stop < 0 && stop == 1) idx_stop = k;
      if (in && idx_veh < 0 && occ == 1) idx_veh = k;
    }
    return;
  }
  for (int k = 0; k < look; k++) {
    int c = vw_path_cell(d, S, v, k, walk);
    if (k == 0) first_cell = c;
    const Cell cc = d.cell[c];
    if (idx_stop < 0 && cc.stop == 1) idx_stop = k;
    if (idx_veh < 0 && cc.occ == 1) idx_veh = k;
    if (idx_stop == 0 || idx_veh == 0) break;
  }
}

__device__ __forceinline__ void swap_ptr(int32_t*& a, int32_t*& b) { int32_t* t = a; a = b; b = t; }

// _compute_path_internal (vehicle_base.py:199-420).  On success the result is in S.P[0..*out_len) (possibly
// empty).  Returns false on tier overflow.
// WAVE: the caller runs with all 64 lanes of its wave executing the same code on the same values (one vehicle per
// wave); plain stores are then harmless duplicates, atomics are issued by lane 0 only, and the searches use
// astar_wave.  !WAVE: one vehicle per lane (k_decide_main), nothing is shared.
template <bool WAVE>
__device__ __forceinline__ int astar_any(const Dev& d, const TsParams& P, AScratch& S, int start_idx, int goal_idx, bool soft,
                                         bool ignore_flow, int maximum_steps, int32_t* out, int out_cap) {
  if (WAVE) return astar_wave(d, P, S, start_idx, goal_idx, soft, ignore_flow, maximum_steps, out, out_cap);
  return astar_dev(d, P, S, start_idx, goal_idx, soft, ignore_flow, maximum_steps, out, out_cap);
}
template <bool WAVE>
__device__ bool compute_path_internal_dev(const Dev& d, const TsParams& P, AScratch& S, VW& v, int& out_len) {
  // ---- phase 0: re-merge with the saved original path (219-277) ----
  for (int which = 0; which < 2; which++) {
    const int kb = which == 0 ? 0 : 2, kp = kb + 1;  // bypass slot, pre-path slot
    const bool active = which == 0 ? (v.f & VF_OVER) != 0 : (v.f & VF_DETOUR) != 0;
    if (!active || v.ax_len[kp] <= 0) continue;
    AxReader r;
    r.init(d, S, v, kp);
    int merge_idx = -1, b = -1;
    for (int q = 0; q < v.ax_len[kp]; q++) {
      int c = r.next();
      if (d.cell[c].occ == 0) { merge_idx = q; b = c; break; }
    }
    if (merge_idx < 0) continue;
    int bl = astar_any<WAVE>(d, P, S, v.pos, b, false, true, P.max_contraflow_overtake_steps, S.BYP, MAXB);
    if (bl < 0) return false;
    if (bl > 0 && S.BYP[bl - 1] == b) {
      int n = 0;
      for (int q = 0; q < bl; q++) S.T[n++] = S.BYP[q];
      int rest = v.ax_len[kp] - (merge_idx + 1);
      if (n + rest > S.cap) return false;
      for (int q = 0; q < rest; q++) S.T[n++] = r.next();
      int32_t* dst = ax_buf(S, kb);
      for (int q = 0; q < bl; q++) dst[q] = S.BYP[q];
      v.ax_staged[kb] = true; v.ax_len[kb] = bl;
      swap_ptr(S.P, S.T);
      out_len = n;
      return true;
    }
  }
  // ---- phase 1: strict; phase 2: soft obstacles (280-306) ----
  const int sx_goal = v.target;
  int la;
  if (v.reach_known && d.reach[v.vid] == 2) {
    // k_reach_strict proved the target unreachable under the strict rules: the search would flood its whole
    // component and return [] (astar_numba.py:239).  Same result, without the flood.
    S.calls++;
    la = 0;
  } else {
    la = astar_any<WAVE>(d, P, S, v.pos, sx_goal, false, false, 0x7FFFFFFF, S.A, S.cap);
    if (la < 0) return false;
  }
  if (la == 0) {
    la = astar_any<WAVE>(d, P, S, v.pos, sx_goal, true, false, 0x7FFFFFFF, S.A, S.cap);
    if (la < 0) return false;
  }
  // ---- phase 3: contraflow overtake of a stranded / parked blocker (309-366) ----
  if (P.contraflow_overtake_active && la > 0) {
    int idx_stop = -1, idx_veh = -1;
    int look = min(P.vehicle_awareness_range, la);
    for (int q = 0; q < look; q++) {
      if (idx_stop < 0 && d.cell[S.A[q]].stop == 1) idx_stop = q;
      if (idx_veh < 0 && d.cell[S.A[q]].occ == 1) idx_veh = q;
      if (idx_stop >= 0 && idx_veh >= 0) break;
    }
    if (idx_veh == 0) {
      int bk = d.cell[S.A[0]].veh;
      if (bk >= 0 && (seen_stranded(d, bk, v.i) || seen_parked(d, bk, v.i))) {
        int bt = -1, idx_bp = -1;
        for (int q = 0; q < la; q++) if (d.cell[S.A[q]].occ == 0) { bt = S.A[q]; idx_bp = q; break; }
        if (bt >= 0) {
          int bl = astar_any<WAVE>(d, P, S, v.pos, bt, false, true, P.max_contraflow_overtake_steps, S.BYP, MAXB);
          if (bl < 0) return false;
          if (bl > 1 && S.BYP[bl - 1] == bt) {
            // idx_bp = first index of bt in path = the index found above (first free cell)
            int n = 0;
            for (int q = 0; q < bl; q++) S.T[n++] = S.BYP[q];
            if (n + (la - idx_bp - 1) > S.cap) return false;
            for (int q = idx_bp + 1; q < la; q++) S.T[n++] = S.A[q];
            for (int q = 0; q < la; q++) S.PO[q] = S.A[q];      // pre_overtake_path = path
            v.ax_staged[1] = true; v.ax_len[1] = la;
            for (int q = 0; q < bl; q++) S.OV[q] = S.BYP[q];    // overtake_path = bypass
            v.ax_staged[0] = true; v.ax_len[0] = bl;
            v.f |= VF_OVER;
            v.d_overtaking++;
            v.over_dur = 0;
            swap_ptr(S.P, S.T);
            out_len = n;
            return true;
          }
        }
      }
    }
  }
  // ---- phase 4: stuck detour (369-418) ----
  if (P.stuck_contraflow_enabled && la > 0) {
    int threshold = st_inter(d.cell[v.pos].stat) == 1 ? P.stuck_contraflow_threshold_intersection : P.stuck_contraflow_threshold;
    if (v.stuck_ticks >= threshold) {
      int bt = -1, merge_idx = -1;
      for (int q = 0; q < la; q++) if (d.cell[S.A[q]].occ == 0) { bt = S.A[q]; merge_idx = q; break; }
      if (bt >= 0) {
        int bl = astar_any<WAVE>(d, P, S, v.pos, bt, true, true, P.max_contraflow_stuck_detour_steps, S.BYP, MAXB);
        if (bl < 0) return false;
        if (bl > 1 && S.BYP[bl - 1] == bt) {
          int n = 0;
          for (int q = 0; q < bl; q++) S.T[n++] = S.BYP[q];
          if (n + (la - merge_idx - 1) > S.cap) return false;
          for (int q = merge_idx + 1; q < la; q++) S.T[n++] = S.A[q];
          for (int q = 0; q < la; q++) S.PD[q] = S.A[q];        // pre_stuck_detour_path = path.copy()
          v.ax_staged[3] = true; v.ax_len[3] = la;
          for (int q = 0; q < bl; q++) S.DV[q] = S.BYP[q];      // stuck_detour_path = bypass
          v.ax_staged[2] = true; v.ax_len[2] = bl;
          v.d_detour++;
          v.f |= VF_DETOUR;
          v.det_dur = 0;
          swap_ptr(S.P, S.T);
          out_len = n;
          return true;
        }
      }
    }
  }
  swap_ptr(S.P, S.A);
  out_len = la;
  return true;
}

// `pos not in aux path k`
__device__ bool ax_contains(const Dev& d, const AScratch* S, const VW& v, int k, int cell) {
  if (v.ax_len[k] <= 0) return false;
  if (!S) {  // no scratch: only pool-resident paths can exist
    int c = d.ax_start[k][v.vid];
    uint32_t off = d.ax_off[k][v.vid];
    for (int q = 0; q < v.ax_len[k]; q++) { c = step_cell(c, path_dir(d.pool, off, q), d.W); if (c == cell) return true; }
    return false;
  }
  AxReader r;
  r.init(d, *S, v, k);
  for (int q = 0; q < v.ax_len[k]; q++) if (r.next() == cell) return true;
  return false;
}

// device-side bump allocation in the path pool; returns false when the pool is exhausted
template <bool WAVE>
__device__ __forceinline__ bool pool_alloc(const Dev& d, int words, uint32_t& off) {
  unsigned long long o = 0;
  if (!WAVE || lane_id() == 0) o = atomicAdd((unsigned long long*)&d.cnt->pool_used, (unsigned long long)words);
  if (WAVE) o = ((unsigned long long)(unsigned)__shfl((int)(o >> 32), 0) << 32) | (unsigned)__shfl((int)(unsigned)o, 0);
  if (o + (unsigned long long)words > (unsigned long long)d.pool_cap_words) return false;
  off = (uint32_t)o;
  return true;
}
__device__ void encode_cells(const Dev& d, uint32_t off, int start_cell, const int32_t* cells, int len) {
  int prev = start_cell;
  uint32_t word = 0;
  for (int k = 0; k < len; k++) {
    int c = cells[k];
    int delta = c - prev;
    int dir = delta == d.W ? 0 : delta == 1 ? 1 : delta == -d.W ? 2 : 3;
    word |= (uint32_t)dir << ((k & 15) * 2);
    if ((k & 15) == 15) { d.pool[off + (k >> 4)] = word; word = 0; }
    prev = c;
  }
  if (len & 15) d.pool[off + (len >> 4)] = word;
}

// step_decide for vehicle number i of active_vehicle_agents.  S == nullptr: run until a search is needed
// (returns DV_DEFER without side effects).  Otherwise completes, unless the tier overflows or the pool is full.
template <bool WAVE>
__device__ int decide_vehicle(const Dev& d, const TsParams& P, int i, AScratch* S) {
  const bool one = !WAVE || lane_id() == 0;   // the lane that issues this vehicle's atomics
  const int vid = d.active[i];
  if (vid < 0) return DV_DONE;
  VW v;
  v.vid = vid; v.i = i; v.pos = d.pos[vid]; v.target = d.target[vid];
  v.reach_known = S != nullptr;
  v.f = d.flags[vid] & ~VF_EARLY;
  const uint8_t ev = d.ev[vid];
  v.base = d.base_speed[vid]; v.cur = d.cur_speed[vid];
  bool early = false;
  int stranded_left = d.stranded_left[vid];
  bool write_stranded = false;
  int dc_coll = 0, dc_malf = 0;
  if (ev == 1) {  // became stranded at its own decide point: state already written by k_apply_event
    v.base = 0; v.cur = 0; early = true;
  } else {
    if (ev != 2 && (v.f & (VF_COLL | VF_MALF))) {  // _tick_stranded (552-565)
      stranded_left -= 1;
      if (stranded_left <= 0) {
        if (v.f & VF_COLL) dc_coll--;
        if (v.f & VF_MALF) dc_malf--;
        v.f &= ~(VF_COLL | VF_MALF);
        stranded_left = 0;
      }
      write_stranded = true;
      if (v.f & (VF_COLL | VF_MALF)) { v.base = 0; v.cur = 0; early = true; }
    }
    if (!early && !P.malfunction_active) {  // `not ACTIVE or ...` (609): malfunction without a draw
      v.f = (v.f | VF_MALF) & ~VF_COLL;
      stranded_left = P.malfunction_duration; write_stranded = true;
      dc_malf++;
      v.base = 0; v.cur = 0; early = true;
    }
    if (!early && d.cell[v.pos].stop == 1) { v.base = 0; v.cur = 0; early = true; }
  }
  int max_steps = d.max_steps[vid];
  bool path_changed = false, reached_body = false, arrived = false;
  v.newpath = false;
  v.d_overtaking = 0; v.d_detour = 0;
  for (int k = 0; k < 4; k++) { v.ax_staged[k] = false; v.ax_len[k] = 0; }
  bool ax_none_set[2] = {false, false};
  if (!early) {
    if (v.base == 0) v.base = d.R[i];  // _choose_new_speed: rolled by the host scan
    int speed = v.base;
    if (P.rain_enabled && d.rain[v.pos] == 1) speed = max(1, speed - P.rain_speed_reduction);
    v.cur = speed;
    v.off = d.path_off[vid]; v.pcur = d.path_cur[vid]; v.plen = d.path_len[vid] - v.pcur;
    v.cooldown = d.cooldown[vid]; v.over_dur = d.over_dur[vid]; v.det_dur = d.det_dur[vid];
    v.stuck_ticks = d.stuck_ticks[vid];
    for (int k = 0; k < 4; k++) { v.ax_staged[k] = false; v.ax_len[k] = d.ax_len[k][vid]; }
    // _recompute_path_on_stuck (506-517): self.path = self._compute_path(use_cache=False)
    const int thresh = st_inter(d.cell[v.pos].stat) == 1 ? P.stuck_recompute_threshold_intersection : P.stuck_recompute_threshold;
    if (v.stuck_ticks >= thresh) {
      if (!S) return DV_DEFER;
      v.cooldown = P.pathfinding_cooldown;
      int len;
      if (!compute_path_internal_dev<WAVE>(d, P, *S, v, len)) return DV_OVERFLOW;
      v.newpath = true; v.plen = len; path_changed = true;
    }
    // _recompute_path_on_obstacle (454-504)
    if ((v.f & VF_OVER) && (v.ax_len[0] <= 0 || !ax_contains(d, S, v, 0, v.pos))) {
      v.ax_len[0] = -1; v.ax_staged[0] = false; ax_none_set[0] = true; v.f &= ~VF_OVER;
    }
    if ((v.f & VF_DETOUR) && (v.ax_len[2] <= 0 || !ax_contains(d, S, v, 2, v.pos))) {
      v.ax_len[2] = -1; v.ax_staged[2] = false; ax_none_set[1] = true; v.f &= ~VF_DETOUR;
    }
    int idx_stop, idx_veh, first_cell;
    scan_ahead_dev(d, P, S, v, idx_stop, idx_veh, first_cell);
    bool done_obst = false;
    if (v.f & VF_OVER) {
      v.over_dur += 1;
      if (v.over_dur <= P.contraflow_overtake_duration) done_obst = true;
    }
    if (!done_obst && (v.f & VF_DETOUR)) {
      v.det_dur += 1;
      if (v.det_dur <= P.contraflow_stuck_detour_duration) done_obst = true;
    }
    if (!done_obst && v.cooldown > 0) {
      if (idx_veh == 0) {
        int b = d.cell[first_cell].veh;
        if (b >= 0 && (seen_stranded(d, b, i) || seen_parked(d, b, i))) {
          // immediate pathfinding
        } else { v.cooldown -= 1; done_obst = true; }
      } else { v.cooldown -= 1; done_obst = true; }
    }
    if (!done_obst && (idx_stop >= 0 || idx_veh >= 0)) {
      if (!S) return DV_DEFER;
      // path = self._compute_path(use_cache=False); adopted only when non-empty (498-502).  The planner
      // never writes through S->P, it only swaps buffer pointers at the end, so a previous result of this
      // tick (stuck replan) survives an empty answer and is swapped back.
      v.cooldown = P.pathfinding_cooldown;
      const bool keep_new = v.newpath;
      int len;
      if (!compute_path_internal_dev<WAVE>(d, P, *S, v, len)) return DV_OVERFLOW;
      if (len > 0) {
        v.newpath = true; v.plen = len; path_changed = true;
        scan_ahead_dev(d, P, S, v, idx_stop, idx_veh, first_cell);
      } else if (keep_new) {
        swap_ptr(S->P, S->A);  // undo the final swap of the empty result
      }
    }
    // _determine_max_steps (719-731)
    int ms = min(v.cur, v.plen);
    bool blocked = false;
    if (idx_stop >= 0) ms = min(ms, idx_stop);
    if (idx_veh >= 0) { if (idx_veh == 0) blocked = true; ms = min(ms, idx_veh); }
    max_steps = ms;
    v.f = blocked ? (v.f | VF_BLOCKED) : (v.f & ~VF_BLOCKED);
    if (ms <= 0) {
      v.base = 0;
      if (v.pos == v.target) arrived = true;   // on_target_reached() inside step_decide (657-661)
      early = true;
    }
    reached_body = true;
  }
  if (ev == 2) { v.base = 0; v.cur = 0; }  // collision inflicted after this vehicle had decided
  // ---------------- commit (first the allocation that can fail, then everything else) ----------------
  if (reached_body && S) {
    int words = path_changed ? (v.plen + 15) / 16 : 0;
    for (int k = 0; k < 4; k++) if (v.ax_staged[k]) words += (v.ax_len[k] + 15) / 16;
    uint32_t off = 0;
    if (words > 0 && !pool_alloc<WAVE>(d, words, off)) return DV_POOL_FULL;
    if (path_changed) {
      encode_cells(d, off, v.pos, S->P, v.plen);
      d.path_off[vid] = off; d.path_len[vid] = v.plen; d.path_cur[vid] = 0;
      off += (v.plen + 15) / 16;
    }
    for (int k = 0; k < 4; k++) {
      if (!v.ax_staged[k]) continue;
      encode_cells(d, off, v.pos, ax_buf(*S, k), v.ax_len[k]);
      d.ax_start[k][vid] = v.pos; d.ax_off[k][vid] = off; d.ax_len[k][vid] = v.ax_len[k];
      off += (v.ax_len[k] + 15) / 16;
    }
  }
  if (reached_body) {
    if (ax_none_set[0] && !v.ax_staged[0]) d.ax_len[0][vid] = -1;
    if (ax_none_set[1] && !v.ax_staged[2]) d.ax_len[2][vid] = -1;
    d.cooldown[vid] = v.cooldown; d.over_dur[vid] = v.over_dur; d.det_dur[vid] = v.det_dur;
    if (one && v.d_overtaking) atomicAdd((unsigned long long*)&d.cnt->overtaking, (unsigned long long)v.d_overtaking);
    if (one && v.d_detour) atomicAdd((unsigned long long*)&d.cnt->in_stuck_detour, (unsigned long long)v.d_detour);
  }
  if (write_stranded) d.stranded_left[vid] = stranded_left;
  if (one && dc_coll) atomicAdd((unsigned long long*)&d.cnt->collisions, (unsigned long long)(long long)dc_coll);
  if (one && dc_malf) atomicAdd((unsigned long long*)&d.cnt->malfunctions, (unsigned long long)(long long)dc_malf);
  d.max_steps[vid] = (int8_t)max_steps;
  d.base_speed[vid] = (int8_t)v.base;
  d.cur_speed[vid] = (int8_t)v.cur;
  d.flags[vid] = early ? (v.f | VF_EARLY) : v.f;
  if (arrived && one) {
    if (!(v.f & VF_KEEP)) atomicExch(&d.cnt->error, TS_E_UNSUPPORTED);  // despawn inside decide (start == goal)
    else if (v.f & VF_TOBLOCK) svc_record(d, i, vid, AR_DECIDE);        // ServiceVehicleAgent._start_service
    else {   // base on_target_reached of a vehicle that stays: trip statistics once more, then _park()
      if (P.enable_traffic && d.pop[vid] == TS_POP_THROUGH) {
        atomicAdd(&d.cnt->dur_through, d.elapsed - d.depart[vid]);
        atomicAdd((unsigned long long*)&d.cnt->dist_through, (unsigned long long)d.steps[vid]);
        atomicAdd((unsigned long long*)&d.cnt->completed_through, 1ULL);
      } else if (P.enable_traffic && d.pop[vid] == TS_POP_INTERNAL) {
        atomicAdd(&d.cnt->dur_internal, d.elapsed - d.depart[vid]);
        atomicAdd((unsigned long long*)&d.cnt->dist_internal, (unsigned long long)d.steps[vid]);
        atomicAdd((unsigned long long*)&d.cnt->completed_internal, 1ULL);
      }
      if (!(v.f & VF_PARKED)) svc_record(d, i, vid, AR_DECIDE);
    }
  }
  return DV_DONE;
}

// every live vehicle: the part of step_decide that needs no search; the others go to the replan list
__global__ void k_decide_main(Dev d, TsParams P, int n_active, RLists lists) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  if (d.cnt->rng_event != 0xFFFFFFFFu) return;  // a malfunction / sideswipe fired: the host re-runs this after the fix-up
  if (decide_vehicle<false>(d, P, i, nullptr) == DV_DEFER) {
    // start where its last search fitted, or where a search over this distance is likely to fit: the searches are
    // Dijkstra-like (the heuristic is far below the penalties), so they touch on the order of md^2 / 2 cells
    const int vid = d.active[i];
    const int p0 = d.pos[vid], p1 = d.target[vid];
    const int md = abs(p0 % d.W - p1 % d.W) + abs(p0 / d.W - p1 / d.W);
    const int by_dist = md < 40 ? 0 : md < 60 ? 1 : md < 240 ? 2 : md < 680 ? 3 : 4;
    const int h = min(max((int)d.tier_hint[vid], by_dist), 4);
    lists.l[stage_list(h)][atomicAdd(&d.cnt->replan_n[stage_counter(h)], 1)] = i;
  }
}

// replanning vehicles: one wave per vehicle, private scratch from tier `t`.  Entries that outgrow the tier go
// to `next_list` (counter replan_n[next_counter]); entries that find the pool full go to `retry_list`.
__global__ void __launch_bounds__(64) k_decide_replan(Dev d, TsParams P, ATier t, const int32_t* list, int begin, int n, int32_t* next_list,
                                int next_counter, int32_t* retry_list, int stage, StageCaps caps) {
  // one wave per vehicle (64 independent searches in one wave would run in lockstep and pay for each other's
  // branches); all 64 lanes run the vehicle's step_decide together and share the work inside the searches
  const int j = blockIdx.x;
  if (j >= n) return;
  AScratch S;
  scratch_bind(t, j, S);
  const int i = list[begin + j];
  int r = decide_vehicle<true>(d, P, i, &S);
  if (threadIdx.x != 0) return;
  t.slot_epoch[j] = S.epoch;
  if (r == DV_DONE) {  // work of attempts that are re-run on a larger tier / after pool growth is not counted twice
    note_tier(d, S, d.active[i], stage, caps);
    atomicAdd((unsigned long long*)&d.cnt->astar_calls, (unsigned long long)S.calls);
    atomicAdd((unsigned long long*)&d.cnt->astar_exp, (unsigned long long)S.expansions);
    atomicAdd((unsigned long long*)&d.cnt->astar_relax, (unsigned long long)S.relaxations);
  }
  if (r == DV_OVERFLOW) next_list[atomicAdd(&d.cnt->replan_n[next_counter], 1)] = i;
  else if (r == DV_POOL_FULL) retry_list[atomicAdd(&d.cnt->replan_n[3], 1)] = i;
}

// The same, with the search structures (dist/came_from table and the heap) in LDS: one wave per vehicle, lane 0
// runs the sequential algorithm.  A* is a chain of dependent accesses; what matters is the latency of each one,
// and LDS answers several times faster than L2.  Searches that outgrow the LDS budget go to the HBM tiers.
constexpr int LDS_NODES = 1024, LDS_HASH = 2048, LDS_HEAP = 2048;
__global__ void __launch_bounds__(64) k_decide_replan_lds(Dev d, TsParams P, ATier cells_tier, const int32_t* list, int begin,
                                                            int n, int32_t* next_list, int next_counter, int32_t* retry_list,
                                                            StageCaps caps) {
  __shared__ HEnt s_ht[LDS_HASH];
  __shared__ QEnt s_hq[LDS_HEAP];
  __shared__ int8_t s_hd[LDS_HEAP];
  const int j = blockIdx.x;
  if (j >= n) return;
  for (int q = threadIdx.x; q < LDS_HASH; q += 64) s_ht[q].stamp = 0;
  __syncthreads();
  AScratch S;
  scratch_bind(cells_tier, j, S);   // cell buffers (paths) from the arena of the first HBM tier
  S.ht = s_ht; S.hmask = LDS_HASH - 1; S.hq = s_hq; S.hd = s_hd; S.heap_cap = LDS_HEAP;
  S.node_cap = LDS_NODES;
  S.peak_nodes = 0;
  S.epoch = 0;
  const int i = list[begin + j];
  int r = decide_vehicle<true>(d, P, i, &S);
  if (threadIdx.x != 0) return;
  if (r == DV_DONE) {
    note_tier(d, S, d.active[i], 0, caps);
    atomicAdd((unsigned long long*)&d.cnt->astar_calls, (unsigned long long)S.calls);
    atomicAdd((unsigned long long*)&d.cnt->astar_exp, (unsigned long long)S.expansions);
    atomicAdd((unsigned long long*)&d.cnt->astar_relax, (unsigned long long)S.relaxations);
  }
  if (r == DV_OVERFLOW) next_list[atomicAdd(&d.cnt->replan_n[next_counter], 1)] = i;
  else if (r == DV_POOL_FULL) retry_list[atomicAdd(&d.cnt->replan_n[3], 1)] = i;
}

// one search on the current maps (the `astar(...)` operator seam, ts_astar) - slot 0 of tier `t`
__global__ void __launch_bounds__(64) k_astar_single(Dev d, TsParams P, ATier t, int start_idx, int goal_idx, int soft,
                                                      int ignore_flow, int maximum_steps, int32_t* out_len) {
  if (blockIdx.x) return;
  AScratch S;
  scratch_bind(t, 0, S);
  int len = astar_wave(d, P, S, start_idx, goal_idx, soft != 0, ignore_flow != 0, maximum_steps, S.A, S.cap);
  if (threadIdx.x) return;
  t.slot_epoch[0] = S.epoch;
  if (len >= 0) {
    atomicAdd((unsigned long long*)&d.cnt->astar_calls, (unsigned long long)S.calls);
    atomicAdd((unsigned long long*)&d.cnt->astar_exp, (unsigned long long)S.expansions);
    atomicAdd((unsigned long long*)&d.cnt->astar_relax, (unsigned long long)S.relaxations);
  }
  *out_len = len;  // -1 = tier overflow; the path cells are in the slot's A buffer
}

// VehicleAgent.__init__ -> self.path = self._compute_path() on a cache miss (vehicle_base.py:80-81, 143-167):
// the phase 0-4 planner for a freshly placed vehicle.  status: path length, or -1 overflow / -2 pool full.
__global__ void __launch_bounds__(64) k_spawn_plan(Dev d, TsParams P, ATier t, int vid, int32_t* status) {
  if (blockIdx.x) return;
  const bool one = threadIdx.x == 0;
  AScratch S;
  scratch_bind(t, 0, S);
  VW v;
  v.vid = vid; v.i = LAST_IDX; v.pos = d.pos[vid]; v.target = d.target[vid];
  v.f = d.flags[vid]; v.base = 0; v.cur = 0; v.cooldown = P.pathfinding_cooldown;
  v.over_dur = d.over_dur[vid]; v.det_dur = d.det_dur[vid]; v.stuck_ticks = d.stuck_ticks[vid];
  v.newpath = false; v.plen = 0; v.pcur = 0; v.off = 0; v.d_overtaking = 0; v.d_detour = 0;
  v.reach_known = false;
  for (int k = 0; k < 4; k++) { v.ax_staged[k] = false; v.ax_len[k] = d.ax_len[k][vid]; }
  int len;
  bool ok = compute_path_internal_dev<true>(d, P, S, v, len);
  t.slot_epoch[0] = S.epoch;
  if (!ok) { *status = -1; return; }
  int words = (len + 15) / 16;
  for (int k = 0; k < 4; k++) if (v.ax_staged[k]) words += (v.ax_len[k] + 15) / 16;
  uint32_t off = 0;
  if (words > 0 && !pool_alloc<true>(d, words, off)) { *status = -2; return; }
  if (one) {
    atomicAdd((unsigned long long*)&d.cnt->astar_calls, (unsigned long long)S.calls);
    atomicAdd((unsigned long long*)&d.cnt->astar_exp, (unsigned long long)S.expansions);
    atomicAdd((unsigned long long*)&d.cnt->astar_relax, (unsigned long long)S.relaxations);
  }
  encode_cells(d, off, v.pos, S.P, len);
  d.path_off[vid] = off; d.path_len[vid] = len; d.path_cur[vid] = 0;
  off += (len + 15) / 16;
  for (int k = 0; k < 4; k++) {
    if (!v.ax_staged[k]) continue;
    encode_cells(d, off, v.pos, ax_buf(S, k), v.ax_len[k]);
    d.ax_start[k][vid] = v.pos; d.ax_off[k][vid] = off; d.ax_len[k][vid] = v.ax_len[k];
    off += (v.ax_len[k] + 15) / 16;
  }
  d.flags[vid] = v.f; d.over_dur[vid] = v.over_dur; d.det_dur[vid] = v.det_dur;
  if (one && v.d_overtaking) atomicAdd((unsigned long long*)&d.cnt->overtaking, (unsigned long long)v.d_overtaking);
  if (one && v.d_detour) atomicAdd((unsigned long long*)&d.cnt->in_stuck_detour, (unsigned long long)v.d_detour);
  *status = len;
}

// path-pool garbage collection: every live vehicle copies the words it still needs into a fresh pool
__global__ void k_pool_gc(Dev d, int n_active, uint32_t* new_pool, unsigned long long* new_used) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  int vid = d.active[i];
  if (vid < 0) return;
  {
    int cur = d.path_cur[vid], len = d.path_len[vid];
    int w0 = cur >> 4, w1 = (len + 15) >> 4;
    int words = w1 - w0;
    uint32_t src = d.path_off[vid] + w0;
    uint32_t dst = words > 0 ? (uint32_t)atomicAdd(new_used, (unsigned long long)words) : 0u;
    for (int q = 0; q < words; q++) new_pool[dst + q] = d.pool[src + q];
    d.path_off[vid] = dst; d.path_cur[vid] = cur & 15; d.path_len[vid] = len - (w0 << 4);
  }
  for (int k = 0; k < 4; k++) {
    int len = d.ax_len[k][vid];
    if (len <= 0) continue;
    int words = (len + 15) >> 4;
    uint32_t src = d.ax_off[k][vid];
    uint32_t dst = (uint32_t)atomicAdd(new_used, (unsigned long long)words);
    for (int q = 0; q < words; q++) new_pool[dst + q] = d.pool[src + q];
    d.ax_off[k][vid] = dst;
  }
}

// Strict reachability of the target, one wave per replanning vehicle: a frontier BFS over the same edges the
// strict A* relaxes (flow bit set, neighbour in bounds, neither occupied nor red).  Unreachable targets are by far
// the most expensive searches (they flood the component before returning []); the flag lets phase 1 skip them.
__global__ void k_reach_strict(Dev d, const int32_t* list, int n_list, uint32_t* visited_all, int32_t* queue_all,
                               size_t words_per, size_t queue_per) {
  const int w = blockIdx.x;
  if (w >= n_list) return;
  const int lane = threadIdx.x;
  const int i = list[w];
  const int vid = d.active[i];
  if (vid < 0) return;
  uint32_t* visited = visited_all + (size_t)w * words_per;
  int32_t* queue = queue_all + (size_t)w * queue_per;
  for (size_t q = lane; q < words_per; q += 64) visited[q] = 0;
  const int start = d.pos[vid], goal = d.target[vid];
  const int W = d.W, H = d.H;
  __syncthreads();
  if (lane == 0) { queue[0] = start; visited[start >> 5] = 1u << (start & 31); }
  __syncthreads();
  int head = 0, tail = 1;
  bool found = false;
  while (head < tail && !found) {
    const int idx = head + lane;
    const int c = idx < tail ? queue[idx] : -1;
    head = min(tail, head + 64);
    const uint8_t bits = c >= 0 ? (uint8_t)st_allowed(d.cell[c].stat) : 0;
    const int cx = c >= 0 ? c % W : 0, cy = c >= 0 ? c / W : 0;
    for (int dd = 0; dd < 4; dd++) {
      int n = -1;
      if (c >= 0 && (bits & (1 << dd))) {
        const int nx = cx + (dd == 1) - (dd == 3), ny = cy + (dd == 0) - (dd == 2);
        if (nx >= 0 && nx < W && ny >= 0 && ny < H) {
          const int nidx = ny * W + nx;
          const Cell nc = d.cell[nidx];
          if (nc.occ != 1 && nc.stop != 1) {
            const uint32_t bit = 1u << (nidx & 31);
            if (!(atomicOr(&visited[nidx >> 5], bit) & bit)) n = nidx;
          }
        }
      }
      const unsigned long long m = __ballot(n >= 0);
      if (n >= 0) {
        const int off = __popcll(m & ((1ULL << lane) - 1));
        if (tail + off < (int)queue_per) queue[tail + off] = n;
      }
      if (__ballot(n == goal && n >= 0)) found = true;
      tail = min((int)queue_per, tail + (int)__popcll(m));
    }
    __syncthreads();  // queue writes of this step are read by the next one
  }
  if (lane == 0) d.reach[vid] = found ? 1 : 2;
}

}  // namespace
