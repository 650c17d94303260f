// astar.h - GPU A* (one search per wavefront, heap in LDS) and the replanning policy of VehicleAgent.
//
// astar_wave restates astar_numba.py:87-239 verbatim, quirks included:
//   * binary heap keyed on f only, strict '<' in both sift routines (52-85);
//   * dir_arr lives in heap-SLOT order and is NOT swapped by the sifts, so prev_dir = dir_arr[0] is a
//     stale slot value (139, 147, 235);
//   * `ng` is a double (R1 penalty 0.5) that truncates when stored into the int32 arrays (226-232);
//   * soft-obstacle penalty int(1000 * (1 + 4 * density)) in double arithmetic on the float32 density.
// What is laid out for the machine instead of restated:
//   * one wavefront = one search; the heap's first LDS_HEAP slots (f, cell: 8 bytes) and their dir bytes live in
//     LDS, deeper levels spill to the searcher's HBM scratch.  g and steps are not carried in the heap: g = f - h(cell)
//     exactly (h is an integer, ng >= 0), and the steps of a non-stale entry are those of the relaxation that wrote the
//     table entry (every relaxation strictly lowers dist, so the entry with g == dist is the last one pushed);
//   * dist / came_from / steps are one 8-byte record per cell in a table indexed DIRECTLY by the cell's position in an
//     8 x 8-tiled order (no hashing, no probing, never outgrown), stamped with the searcher's epoch instead of the
//     reference's O(W*H) initialisation per call (119-122).  A searcher's table is N x 8 bytes; the 288 GB of HBM
//     hold several hundred of them even at 4096^2;
//   * the maps a search reads are a per-tick snapshot in the same tiled order (Dev::amap: static byte + occupied +
//     red, 2 bytes per cell = one 128-byte line per tile), so the neighbours of a cell usually share its line;
//   * everything an expansion needs from HBM (5 map entries, 5 table records, 4 densities) is requested as soon as
//     the popped cell is known and travels while the sift-down works in LDS: one memory round trip per expansion.
//
// decide_vehicle restates step_decide (vehicle_base.py:616-663) with _recompute_path_on_stuck 506-517,
// _recompute_path_on_obstacle 454-504, _compute_path 143-167 and _compute_path_internal 199-420
// (phases 0-4).  It is a pure function of the tick-start state until its final commit, so the same code
// runs in k_decide_main (one vehicle per lane, no scratch: bails out as soon as a search is needed) and in
// k_replan (one vehicle per wave, every lane executing the same code on the same values).
#pragma once
#include <type_traits>
#include "dev.h"

namespace {

constexpr int A_INF = 0x3F3F3F3F;
constexpr int MAXB = 64;  // longest contraflow bypass (VEHICLE_MAX_CONTRAFLOW_*_STEPS <= 64)
enum { DV_DONE = 0, DV_DEFER = 1, DV_OVERFLOW = 2, DV_POOL_FULL = 3, DV_SUSPEND = 4, DV_BAIL = 5 };
// Who runs a vehicle's step_decide: one lane without scratch (k_decide_main: bails out as soon as a search is needed), one
// wavefront (k_replan: every lane the same code on the same values, searches spread over the wave), or one quad of four
// lanes (k_replan_quad, astar_quad.h: sixteen vehicles per wave, their searches advancing in lockstep)
enum { DM_LANE = 0, DM_WAVE = 1, DM_QUAD = 2 };

// heap slots (and dir bytes) a searcher keeps in LDS: 6.2 KB, twenty-four searchers per CU (736 entries already cost occupancy) (the deepest heap seen on
// 1024^2 - 4096^2 runs is ~2100 entries; what does not fit spills to the searcher's HBM scratch)
#ifndef TS_LDS_HEAP
#define TS_LDS_HEAP 704
#endif
constexpr int LDS_HEAP = TS_LDS_HEAP;
// register budget of the replanning kernels and (propagated by the compiler) of the functions they call: at least this
// many waves per SIMD.  Four searchers per SIMD keep its vector ALU ~70 % busy (profiles/r02_sq_replan_2048.json); six
// (the search loop needs 72 vector registers then, spills stay outside it) are 11 % faster at 4096^2 / 10^6 vehicles once
// the queue is ordered in space
#ifndef TS_REPLAN_WAVES
#define TS_REPLAN_WAVES 6
#endif
#define TS_REPLAN_OCC __attribute__((amdgpu_waves_per_eu(TS_REPLAN_WAVES, 8)))
struct __attribute__((aligned(8))) HQ { int32_t f, i; };             // heap entry: f_arr, i_arr (g_arr / s_arr: see above)
struct __attribute__((aligned(8))) TEnt { int32_t dist; uint32_t meta; };   // meta = stamp << 14 | steps << 2 | came-from direction
// (TS_DEBUG_STAMP_MAX: a test build that wraps the searcher tables' epochs every few hundred searches instead of every 262 143)
#ifndef TS_DEBUG_STAMP_MAX
#define TS_DEBUG_STAMP_MAX ((1u << 18) - 1)
#endif
constexpr uint32_t T_STAMP_SHIFT = 14, T_STEPS_MASK = 0xFFF, T_STAMP_MAX = TS_DEBUG_STAMP_MAX;
constexpr int A_STEPS_MAX = (int)T_STEPS_MASK - 1;   // largest binding step limit a search can carry (a limit >= N never binds)

__shared__ unsigned long long g_lq[LDS_HEAP];   // packed HQ: f in the low word, cell in the high word
__shared__ int8_t g_ld[LDS_HEAP];
__shared__ int g_job;   // k_replan: the work-queue entry the wave is on

// One searcher's scratch: LDS heap (above) + its slot of the HBM arena.
struct AScratch {
  HQ* gq;        // heap slots [LDS_HEAP, heap_cap), indexed by slot - LDS_HEAP
  int8_t* gd;    // dir_arr for the same slots: indexed by heap SLOT and deliberately not moved by the sift routines
  int heap_cap;
  TEnt* tab;     // one record per search node (Dev::n_nodes), nodes numbered in tiled order
  uint32_t epoch;
  int32_t *A, *P, *T, *PO, *PD;  // cap cells each: A* result, current new path, splice target, staged pre-paths
  int32_t *BYP, *OV, *DV;        // MAXB cells each: bypass result, staged overtake / detour paths
  int cap;        // capacity of the cell buffers
  int use_reach;  // phase 1 asks reach_strict_wave before it searches (off: TS_NO_REACH, a debugging switch)
  long long calls, expansions, relaxations;
  // DM_QUAD only (astar_quad.h): the policy code is re-run from the top after every search (it is a pure function of the
  // tick-start state and of its searches' results until the final commit), taking finished searches from this log and
  // suspending at the first one that is not in it
  int q_status, q_replay, q_done;          // DV_SUSPEND / DV_BAIL when a planner returns false; searches replayed / finished
  int32_t* q_log;                          // per finished search: path length, expansions, relaxations
  int q_start, q_goal, q_soft, q_cap;      // the search the policy is waiting for
  int32_t* q_out;
};
struct RLists { int32_t* l[6]; };       // replan work lists, see run_replans (engine.hip)

struct ASlots {
  int n_slots, heap_cap, cap, use_reach;
  size_t tab_entries;    // per slot
  TEnt* tab;
  HQ* gq;
  int8_t* gd;
  int32_t* cells;        // per slot: 5 * cap + 3 * MAXB
  uint32_t* slot_epoch;
};

__device__ __forceinline__ void scratch_bind(const ASlots& t, int slot, AScratch& S) {
  S.tab = t.tab + (size_t)slot * t.tab_entries;
  S.gq = t.gq + (size_t)slot * (size_t)(t.heap_cap - LDS_HEAP);
  S.gd = t.gd + (size_t)slot * (size_t)(t.heap_cap - LDS_HEAP);
  S.heap_cap = t.heap_cap;
  int32_t* c = t.cells + (size_t)slot * ((size_t)5 * t.cap + 3 * MAXB);
  S.A = c; S.P = c + t.cap; S.T = c + 2 * (size_t)t.cap; S.PO = c + 3 * (size_t)t.cap; S.PD = c + 4 * (size_t)t.cap;
  S.BYP = c + 5 * (size_t)t.cap; S.OV = S.BYP + MAXB; S.DV = S.OV + MAXB;
  S.cap = t.cap;
  S.use_reach = t.use_reach;
  S.epoch = t.slot_epoch[slot];
  S.calls = 0; S.expansions = 0; S.relaxations = 0;
  S.q_status = 0; S.q_replay = 0; S.q_done = 0; S.q_log = nullptr; S.q_start = 0; S.q_goal = 0; S.q_soft = 0; S.q_cap = 0; S.q_out = nullptr;
}
// Work-queue order of the replans (largest first).  Per vehicle the bit length of the expansions its last replan took is
// kept (Dev::tier_hint); the four classes are ranges of it (< 2 048, < 32 768, < 262 144 expansions, more), and inside a
// class the queue is sorted by it again (run_replans), so that the longest search of a tick starts first.
__device__ __forceinline__ int cost_bits(long long expansions) {
  return 64 - __builtin_clzll((unsigned long long)max(expansions, 0ll) | 1ull);
}
__device__ __forceinline__ int cost_class_of_bits(int b) { return b < 12 ? 0 : b < 16 ? 1 : b < 19 ? 2 : 3; }
// ... and for a vehicle without history, from the distance to its target: the searches are Dijkstra-like (the heuristic is
// far below the penalties), so they touch on the order of md^2 / 2 cells
__device__ __forceinline__ int cost_bits_of_distance(int md) { return md < 60 ? 8 : md < 240 ? 13 : md < 680 ? 17 : 19; }
__device__ __forceinline__ int replan_cost_bits(const Dev& d, int vid) {
  int x0, y0, x1, y1;
  cell_xy(d, d.pos[vid], x0, y0); cell_xy(d, d.target[vid], x1, y1);
  return max((int)d.tier_hint[vid], cost_bits_of_distance(abs(x0 - x1) + abs(y0 - y1)));
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ void wave_mem_sync() { __builtin_amdgcn_wave_barrier(); }
// values every lane holds alike: tell the compiler (scalar registers, scalar branches) / read one lane's copy
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// __ballot() takes an int: a lane predicate would be turned into 0 / 1 and compared against 0 again (two vector
// instructions and a hazard nop per ballot); this form hands the compare's own lane mask over
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double rl(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Heap entries and table records travel as packed 64-bit words (HQ: f low, cell high; TEnt: dist low, meta high), and
// the searcher's HBM arrays are addressed through global-address-space pointers: pointers that reach a function inside
// a struct are generic to the compiler, and generic ("flat") loads count against the LDS wait counter as well, which
// would make every LDS wait of the sift-down also wait for the expansion's loads in flight.
#define TS_GLOBAL __attribute__((address_space(1)))
typedef unsigned long long u64;
typedef TS_GLOBAL u64* gu64p;
typedef TS_GLOBAL int8_t* gi8p;
typedef TS_GLOBAL int32_t* gi32p;
__device__ __forceinline__ u64 hq_pack(int f, int i) { return (u64)(uint32_t)f | ((u64)(uint32_t)i << 32); }
__device__ __forceinline__ int hq_f(u64 e) { return (int)(uint32_t)e; }
__device__ __forceinline__ int hq_i(u64 e) { return (int)(uint32_t)(e >> 32); }
__device__ __forceinline__ u64 rl(u64 v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 uni64(u64 v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((u64)hi << 32) | lo;
}
// Element `idx` of an array of 8-byte records whose byte size stays below 4 GiB (map snapshot, searcher tables: at most
// 2^26 records): the 32-bit byte offset lets the access use its scalar-base + 32-bit-offset form instead of 64-bit
// vector address arithmetic.
__device__ __forceinline__ u64 ld8(const TS_GLOBAL u64* base, uint32_t idx) {
  return *(const TS_GLOBAL u64*)((const TS_GLOBAL char*)base + (uint32_t)(idx << 3));
}
__device__ __forceinline__ void st8(TS_GLOBAL u64* base, uint32_t idx, u64 v) {
  *(TS_GLOBAL u64*)((TS_GLOBAL char*)base + (uint32_t)(idx << 3)) = v;
}
// heap slot k / its dir byte: LDS below LDS_HEAP, the searcher's HBM spill above (gq / gd are the slot's spill arrays).
// SPILL = false: the caller guarantees k < LDS_HEAP - straight LDS accesses, no branch (and with it no conservative
// wait for the expansion's loads in flight) in the hot loop.
template <bool SPILL> __device__ __forceinline__ u64 hq_get(gu64p gq, int k) {
  if constexpr (SPILL) return k < LDS_HEAP ? g_lq[k] : gq[k - LDS_HEAP];
  else return g_lq[k];
}
template <bool SPILL> __device__ __forceinline__ void hq_put(gu64p gq, int k, u64 v) {
  if constexpr (SPILL) { if (k < LDS_HEAP) g_lq[k] = v; else gq[k - LDS_HEAP] = v; }
  else g_lq[k] = v;
}
template <bool SPILL> __device__ __forceinline__ int hd_get(gi8p gd, int k) {
  if constexpr (SPILL) return k < LDS_HEAP ? (int)g_ld[k] : (int)gd[k - LDS_HEAP];
  else return (int)g_ld[k];
}
template <bool SPILL> __device__ __forceinline__ void hd_put(gi8p gd, int k, int v) {
  if constexpr (SPILL) { if (k < LDS_HEAP) g_ld[k] = (int8_t)v; else gd[k - LDS_HEAP] = (int8_t)v; }
  else g_ld[k] = (int8_t)v;
}

// a fresh epoch for the searcher's table (cleared by the wave when the 18-bit stamp wraps)
__device__ __forceinline__ uint32_t next_epoch(const Dev& d, AScratch& S) {
  if (S.epoch >= T_STAMP_MAX) {
    const size_t n = (size_t)max(d.n_nodes, 1);
    for (size_t q = lane_id(); q < n; q += 64) S.tab[q] = TEnt{0, 0u};
    S.epoch = 0;
    __syncthreads();
  }
  return ++S.epoch;
}

// what a search keeps in registers while its loop runs (d, P and S themselves live in scratch memory behind references)
// Half units carry the search costs exactly iff every penalty is a non-negative multiple of 0.5 of moderate size (the
// reference's defaults are).  Host (k_amap_build's penalty bits) and device (astar_wave) decide with the same function.
__host__ __device__ inline bool astar_half_units(const TsParams& P) {
  const double pp[7] = {(double)P.turn_penalty, (double)P.contraflow_penalty, (double)P.obstacle_penalty_vehicle,
                        (double)P.obstacle_penalty_stop, (double)P.road_type_penalty_r1, (double)P.road_type_penalty_r2,
                        (double)P.road_type_penalty_r3};
  bool half = true;
  for (int k = 0; k < 7; k++) half = half && pp[k] >= 0.0 && pp[k] < 1048576.0 && (double)(int)(pp[k] * 2.0) == pp[k] * 2.0;
  half = half && P.dynamic_penalty_scale >= 0.0 && P.dynamic_penalty_scale <= 64.0;   // (2 (g + 1) + penalties stays below 2^31 for every g < INF)
  // the per-cell vehicle penalty travels in 22 bits of the map snapshot (density <= 1)
  half = half && (double)P.obstacle_penalty_vehicle * (1.0 + (double)P.dynamic_penalty_scale) * 2.0 < 2097152.0;
  return half;
}
// obstacle penalty of an occupied cell under VEHICLE_DYNAMIC_PENALTIES (astar_numba.py:203-206): int(veh_pen * (1 + scale * density))
__device__ __forceinline__ double occ_penalty_dyn(double veh_pen, double dyn_scale, float dens) {
#pragma clang fp contract(off)
  return __builtin_trunc(veh_pen * (1.0 + dyn_scale * (double)dens));
}
constexpr int AMAP_PEN_SHIFT = 10;   // Dev::amap low word: bits 0-9 flags, bits 10-31 the vehicle penalty in half units (HALF searches)

struct ACtx {
  int W, H, W8, N, lane;
  u64 w_magic;
  gu64p gq, tab;
  gi8p gd;
  const TS_GLOBAL u64* amap;
  const TS_GLOBAL u64* fovrun;
  const TS_GLOBAL float* density;
  gi32p outg;
  int heap_cap, out_cap, start_idx, goal_idx, gx, gy, maximum_steps, sx, sy, aw;
  bool fov;
  uint32_t epoch;
  bool soft, ignore_flow, limited, turn_on, rt_on, dens_on;
  double turn_pen, contra_pen, veh_pen, stop_pen, dyn_scale, rt1, rt2, rt3;
  int turn2, contra2, veh2, stop2, rt2_1, rt2_2, rt2_3;     // the same in half units (valid when `half`)
  bool half;
  long long n_exp, n_relax, n_exp_spill;   // n_exp_spill: expansions made while part of the heap sat in HBM
  int max_heap;
  long long prof[8], pt;
  __device__ __forceinline__ void xy_of(int cell, int& x, int& y) const {
    if (w_magic) { y = (int)(((u64)(unsigned)cell * w_magic) >> 40); x = cell - y * W; }
    else { y = cell / W; x = cell - y * W; }
  }
  __device__ __forceinline__ uint32_t tile_ix(int x, int y) const {
    // (tile rows and tiles per row are far below 2^24: the full-rate 24-bit multiply)
    return (((__umul24((uint32_t)(y >> 3), (uint32_t)W8) + (uint32_t)(x >> 3)) << 6) | (uint32_t)((y & 7) << 3) | (uint32_t)(x & 7));
  }
};
#ifdef TS_KPROF
#define KP(k) do { const long long _t = clock64(); C.prof[k] += _t - C.pt; C.pt = _t; } while (0)
#else
#define KP(k) do { } while (0)
#endif
enum { AL_EMPTY = -2, AL_OVERFLOW = -1, AL_SWITCH = -3 };   // astar_loop results besides a path length >= 0

// ---------------------------------------------------------------------------------------------
// astar_core's main loop, one search spread over one wavefront.  The algorithm is the sequential one - same heap
// layout, same comparisons, same order of relaxations - only its work is organised by lanes:
//   * everything the expansion of the popped cell reads from HBM is requested as soon as the cell is known (lanes
//     0-3: map entry, table record and density of neighbour `lane`; the other lanes: the cell's own) and travels while
//     the sift-down works on the heap;
//   * sift-down: lanes 2..63 fetch the 62 entries of the five levels below the hole at once (lane L's children sit
//     on lanes 2L and 2L + 1, siblings on an even / odd lane pair); every lane decides with its sibling's key (one DPP
//     swap) whether its entry would move up if its parent were the hole, one ballot collects that, the walk down
//     the five levels is scalar bit tests on the ballot, and the entries on the path move up with one masked write;
//   * sift-up of a push: the ancestors of the new slot are fetched at once, a ballot finds how far the entry rises,
//     the lanes holding ancestors write them one level down in parallel;
//   * the four neighbours are evaluated on lanes 0-3, then committed in the reference's order N, E, S, W.
// SPILL = false runs while the whole heap fits LDS (straight LDS accesses); SPILL = true is the general form.  Either
// returns AL_SWITCH when the other one should take over.
// ---------------------------------------------------------------------------------------------
template <bool SPILL, bool HALF, bool FOV>
__device__ __forceinline__ int astar_loop(ACtx& C, int& heap_size) {
  // A lone wave issues one instruction every four cycles whatever its kind, so this loop is written for instruction
  // count: lane-parallel vector work and one ballot in place of scalar walks, no scalar <-> vector round trips that
  // can be avoided.
  const int lane = C.lane;
  const int W = C.W, H = C.H;
  const gu64p gq = C.gq;
  const gi8p gd = C.gd;
  const gu64p tab = C.tab;
  const uint32_t epoch = C.epoch, stamp = C.epoch << T_STAMP_SHIFT;
  const int gx = C.gx, gy = C.gy;
  // window geometry of this lane (lanes 2..63 = the 62 entries of five levels below a hole that sits on "lane 1")
  const int wlvl = 31 - __builtin_clz((unsigned)max(lane, 1));   // 1 for lanes 2-3, ... 5 for 32-63
  const int woff = lane - (1 << wlvl);
  const bool wlane = lane >= 2;
  const int wadj = (lane & 1) == 0 ? 1 : 0;                       // left children (even lanes) win ties against their sibling
  unsigned long long wanc = 0;                                     // this lane's ancestors-or-self inside the window
  for (int j = 0; j < wlvl; j++) wanc |= 1ull << (lane >> j);
  if (!wlane) wanc = ~0ull;                                        // lanes 0-1 hold no entry: their path test can never pass (bits 0-1 of a winner mask are never set)
  {  // pin the mask in registers: left alone, the compiler re-derives it (a six-step loop) in every turn of the main loop
    unsigned lo = (unsigned)wanc, hi = (unsigned)(wanc >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    wanc = ((unsigned long long)hi << 32) | lo;
  }
  const int dd_l = lane & 3;
  const int dx_l = lane < 4 ? (dd_l == 1) - (dd_l == 3) : 0, dy_l = lane < 4 ? (dd_l == 0) - (dd_l == 2) : 0;
  const unsigned below_l = (1u << lane) - 1u;                      // (lanes 0-3 use it)
  while (heap_size > 0) {
    if (!SPILL && heap_size > LDS_HEAP - 4) return AL_SWITCH;        // this turn's pushes might not fit LDS
    if (SPILL && heap_size < LDS_HEAP - 96) return AL_SWITCH;      // (heaps breathe by a few entries per turn: a band of ~90 keeps the switches rare)
    KP(7);
    // one LDS read serves the root (lane 0; every lane gets it through readfirstlane) and the first sift-down window
    // (lanes 2..63: slots 1..62 - the window below a hole at the root): nothing has written to them in this turn yet
    const u64 w0 = g_lq[max(lane - 1, 0)];
    const int prev_dir = uni((int)g_ld[0]);
    const u64 x = uni64(hq_get<SPILL>(gq, heap_size - 1));            // the last entry: it takes the root's place
    const int xd = uni(hd_get<SPILL>(gd, heap_size - 1));
    const int f_top = uni(hq_f(w0)), cur = uni(hq_i(w0));     // (readfirstlane: lane 0's word, the root)
    heap_size--;
    KP(0);
    int cx, cy;
    C.xy_of(cur, cx, cy);
    const int nx_l = cx + dx_l, ny_l = cy + dy_l;
    const bool inb_l = (unsigned)nx_l < (unsigned)W && (unsigned)ny_l < (unsigned)H;
    const int nidx_l = inb_l ? ny_l * W + nx_l : cur;
    const uint32_t t_l = inb_l ? C.tile_ix(nx_l, ny_l) : C.tile_ix(cx, cy);
    // round 1: the map entries (flags + search-node number); round 2, issued half-way through the sift-down: the table
    // records of those nodes
    const u64 am_l = ld8(C.amap, t_l);
    float dens_l = 0.0f;      // (HALF searches find the cell's vehicle penalty in the map entry itself)
    if constexpr (!HALF) dens_l = C.density[nidx_l];      // (read whether or not the search is soft: no branch around a load)
    u64 fr_l = 0;
    if constexpr (FOV) fr_l = ld8(C.fovrun, t_l);
    u64 e_l = 0;
    bool e_loaded = false;
    KP(1);
    if (heap_size > 0) {
      const int xf = hq_f(x);
      wave_mem_sync();                     // every lane has read slot 0 before it is overwritten
      if (lane == 0) g_ld[0] = (int8_t)xd;
      int idx = 0;                         // the hole; x keeps sinking
      for (;;) {
        const int abs_l = ((idx + 1) << wlvl) + woff - 1;
        const bool valid = wlane & (abs_l < heap_size);
        const u64 mine = idx == 0 ? w0 : hq_get<SPILL>(gq, valid ? abs_l : 0);   // (idx == 0: abs_l = lane - 1, read above)
        const int mf = valid ? hq_f(mine) : 0x7FFFFFFF;
        const int sf = __builtin_amdgcn_mov_dpp(mf, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: the sibling's key
        // smallest of (x, left, right) with ties going to x, then left (heap_sift_down, astar_numba.py:67-85):
        // "my entry moves up if my parent is the hole" - at most one of two siblings.  Left: mine <= sibling's, right:
        // mine < sibling's; keys are >= 0, so mine <= s is (mine - 1) < s.
        // (one compare per ballot: the backend hands a compare's lane mask over as it is, anything else is turned into
        // 0 / 1 and compared again.)  mf < xf and mf - wadj < sf  <=>  mf < min(xf, sf + wadj), in unsigned arithmetic
        // (keys are >= 0, an empty slot is 0x7FFFFFFF: the sum cannot wrap)
        const unsigned long long wmask = ballot((unsigned)mf < min((unsigned)xf, (unsigned)sf + (unsigned)wadj));
        // A lane's entry is on the sift path iff it and all its ancestors inside the window are winners: one mask
        // test per lane (wanc = the lane's ancestors-or-self).  Exactly one lane per level can pass, so a second
        // ballot yields both the number of levels the hole sinks (k) and where it ends up (L).
        const bool onp = (wmask & wanc) == wanc;
        const unsigned long long pmask = ballot(onp);
        const int k = __builtin_popcountll(pmask);
        const int L = 63 - __builtin_clzll(pmask | 2ull);             // deepest lane on the path; the hole itself (1) if none
        if (onp) hq_put<SPILL>(gq, (abs_l - 1) >> 1, mine);          // the entries on the path move up one level
        const int abs_end = uni(((idx + 1) << k) + (L - (1 << k)) - 1);
        const bool done = k < 5;
        if (done & (lane == 0)) hq_put<SPILL>(gq, abs_end, x);
        if (!e_loaded) {                   // the first window is done: the map entries have had time to arrive
          const uint32_t r_l = (uint32_t)(am_l >> 32);
          e_l = ld8(tab, r_l != 0xFFFFFFFFu ? r_l : 0u);
          e_loaded = true;
        }
        if (done) break;
        idx = abs_end;                     // five levels down and still sinking: next window
      }
    }
    wave_mem_sync();
    KP(2);
    const uint32_t a_l = (uint32_t)am_l, r_l = (uint32_t)(am_l >> 32);
    const bool node_l = r_l != 0xFFFFFFFFu;
    if (!e_loaded) e_l = ld8(tab, node_l ? r_l : 0u);     // (the heap held a single entry: no sift-down happened)
    if (cur == C.goal_idx) {
      // walk came_from back to the start, filling the output from its far end, then slide it to the front
      const gi32p outg = C.outg;
      const int out_cap = C.out_cap;
      int len = 0, px = cx, py = cy;
      for (int c = cur; c != C.start_idx;) {
        if (len >= out_cap) return AL_OVERFLOW;
        if (lane == 0) outg[out_cap - 1 - len] = c;
        len++;
        const uint32_t pr = (uint32_t)(C.amap[C.tile_ix(px, py)] >> 32);
        const int dd = (int)((tab[pr] >> 32) & 3u);
        px -= (dd == 1) - (dd == 3); py -= (dd == 0) - (dd == 2);
        c = py * W + px;
      }
      __syncthreads();
      const int shift = out_cap - len;
      if (shift > 0)
        for (int k0 = 0; k0 < len; k0 += 64) {
          const int k = k0 + lane;
          const int v = k < len ? outg[shift + k] : 0;
          if (k < len) outg[k] = v;
        }
      __syncthreads();
      return len;
    }
    const int g = f_top - (abs(cx - gx) + abs(cy - gy));
    const u64 e_c = rl(e_l, 4);
    const bool node_c = rl((int)node_l, 4) != 0;     // (only a start cell can be off the node set: dist 0, no steps)
    const uint32_t m_c = node_c ? (uint32_t)(e_c >> 32) : (epoch << T_STAMP_SHIFT);
    {
      const int dist_c = !node_c ? 0 : (m_c >> T_STAMP_SHIFT) == epoch ? (int)(uint32_t)e_c : A_INF;
      if (g > dist_c) continue;
    }
    KP(3);
    C.n_exp++;
    if constexpr (SPILL) C.n_exp_spill++;
    const int steps = C.limited ? (int)((m_c >> 2) & T_STEPS_MASK) : 0;
    const uint32_t bits = (uint32_t)rl((int)a_l, 4) & 15u;
    // ---- lane dd < 4 evaluates neighbour dd from what was fetched before the sift-down -------------------------
    // HALF: every penalty is a non-negative multiple of 0.5 (the reference's defaults are), so the reference's float `ng`
    // is carried exactly as an integer count of half units: ng < dist  <=>  ng2 < 2 dist, int(ng) = ng2 >> 1,
    // int(ng + h) = (ng2 >> 1) + h.  Otherwise the same in doubles.
    double ng_l = 0.0;
    int ng2_l = 0;
    bool ok_l;
    {
      const uint32_t m_l = (uint32_t)(e_l >> 32);
      const int dist_l = (m_l >> T_STAMP_SHIFT) == epoch ? (int)(uint32_t)e_l : A_INF;
      bool n_occ = ((a_l >> 8) & 1u) != 0u, n_stop = ((a_l >> 9) & 1u) != 0u;
      const bool n_road = ((a_l >> 4) & 1u) != 0u;
      if constexpr (FOV) {
        // compute_fov_inplace (astar_numba.py:29-50): the neighbour is seen iff a straight run of road cells joins it to
        // the line of 2 * awareness - 1 cells through the START cell perpendicular to that run's direction
        const int dxs = nx_l - C.sx, dys = ny_l - C.sy, aw = C.aw;
        const int run_dn = (int)(fr_l & 0xFFFF), run_up = (int)((fr_l >> 16) & 0xFFFF), run_lf = (int)((fr_l >> 32) & 0xFFFF), run_rt = (int)(fr_l >> 48);
        const bool band_x = (dxs < aw) & (dxs > -aw), band_y = (dys < aw) & (dys > -aw);
        const bool seen = (band_x & (dys >= 0) & (run_dn > dys)) | (band_x & (dys <= 0) & (run_up > -dys)) |
                          (band_y & (dxs >= 0) & (run_lf > dxs)) | (band_y & (dxs <= 0) & (run_rt > -dxs));
        n_occ &= seen; n_stop &= seen;
      }
      const uint32_t rt = (a_l >> 6) & 3u;
      const bool flow = ((bits >> dd_l) & 1u) != 0u;
      const bool turn = C.turn_on & (prev_dir != -1) & (dd_l != prev_dir);
      bool cheaper;
      // (selects, not branches: a killed neighbour's cost is simply not used)
      if constexpr (HALF) {
        int n2 = 2 * (g + 1);
        n2 += turn ? C.turn2 : 0;
        n2 += flow ? 0 : C.contra2;
        const int occ2 = (int)(a_l >> AMAP_PEN_SHIFT);     // k_amap_build: veh2, or the density-dependent penalty of this cell
        n2 += n_occ ? occ2 : 0;
        n2 += n_stop ? C.stop2 : 0;
        const int rtp = rt == 1u ? C.rt2_1 : rt == 2u ? C.rt2_2 : rt == 3u ? C.rt2_3 : 0;
        n2 += (C.rt_on & n_road) ? rtp : 0;
        ng2_l = n2;
        cheaper = n2 < 2 * dist_l;
      } else {
        ng_l = (double)(g + 1);
        ng_l += turn ? C.turn_pen : 0.0;
        ng_l += flow ? 0.0 : C.contra_pen;
        const double occ_pen = C.dens_on ? occ_penalty_dyn(C.veh_pen, C.dyn_scale, dens_l) : C.veh_pen;
        ng_l += n_occ ? occ_pen : 0.0;
        ng_l += n_stop ? C.stop_pen : 0.0;
        ng_l += (C.rt_on & n_road) ? (rt == 1u ? C.rt1 : rt == 2u ? C.rt2 : rt == 3u ? C.rt3 : 0.0) : 0.0;
        cheaper = ng_l < (double)dist_l;
      }
      ok_l = (lane < 4) & inb_l & node_l & (steps + 1 <= C.maximum_steps) & (flow | (C.ignore_flow & n_road)) & (C.soft | !(n_occ | n_stop)) & cheaper;
    }
    // ---- commit.  The four neighbours are distinct cells, so no relaxation changes another one's test: the table
    // records and the dir bytes of all of them go out with one masked store each; only the heap pushes are made one
    // after the other, in the reference's order N, E, S, W.
    unsigned relax = (unsigned)(ballot(ok_l) & 15ull);
    KP(4);
    if (relax == 0u) continue;
    const int n_new = __builtin_popcount(relax);
    C.n_relax += n_new;
    if (heap_size + n_new > C.heap_cap) return AL_OVERFLOW;
    C.max_heap = max(C.max_heap, heap_size + n_new);
    const int h_l = abs(nx_l - gx) + abs(ny_l - gy);
    const int ngi_l = HALF ? (ng2_l >> 1) : (int)ng_l;
    const u64 ent_l = hq_pack(HALF ? ngi_l + h_l : (int)(ng_l + (double)h_l), nidx_l);
    if (ok_l) {
      st8(tab, r_l, (u64)(uint32_t)ngi_l | ((u64)(stamp | (C.limited ? (uint32_t)(steps + 1) << 2 : 0u) | (uint32_t)dd_l) << 32));
      hd_put<SPILL>(gd, heap_size + __builtin_popcount(relax & below_l), dd_l);
    }
    KP(5);
    while (relax) {
      const int dd = __builtin_ctz(relax);
      relax &= relax - 1;
      const u64 nx64 = rl(ent_l, dd);
      const int nf = hq_f(nx64);
      const int i = heap_size;
      // ancestors of slot i: a_k = ((i + 1) >> k) - 1, k = 1 .. depth; lane k - 1 fetches a_k
      const int depth = 31 - __builtin_clz((unsigned)(i + 1));
      const bool has = lane < depth;                                  // depth <= 31: lanes beyond it fetch nothing
      const int a_mine = (int)(((unsigned)(i + 1) >> ((lane + 1) & 31)) - 1u) & (has ? -1 : 0);
      const u64 anc = hq_get<SPILL>(gq, a_mine);
      const unsigned long long rises = ballot(has & (nf < hq_f(anc)));
      const int r = __builtin_ctzll(~rises);                          // leading ancestors the entry passes (lanes >= 31 never rise)
      if (lane < r) hq_put<SPILL>(gq, (int)((unsigned)(i + 1) >> (lane & 31)) - 1, anc);   // ancestor k moves to where k - 1 was
      if (lane == 0) hq_put<SPILL>(gq, ((i + 1) >> r) - 1, nx64);
      heap_size++;
      wave_mem_sync();
    }
    KP(6);
  }
  return AL_EMPTY;
}

// astar_core (astar_numba.py:87-239).  All 64 lanes call it with identical arguments and get the same return value.
// Writes the path (start excluded, goal included) to out[0..len); returns len >= 0, or -1 when the heap or the output
// buffer is too small.
__device__ int astar_wave(const Dev& d, const TsParams& P, AScratch& S, int start_idx, int goal_idx, bool soft,
                          bool ignore_flow, int maximum_steps, int32_t* out, int out_cap) {
  ACtx C;
  C.lane = lane_id();
  // the arguments arrive in vector registers: tell the compiler they are wave-uniform (scalar loop control)
  C.start_idx = uni(start_idx); C.goal_idx = uni(goal_idx); C.maximum_steps = uni(maximum_steps); C.out_cap = uni(out_cap);
  C.soft = uni((int)soft) != 0; C.ignore_flow = uni((int)ignore_flow) != 0;
  S.calls++;
  C.epoch = (uint32_t)uni((int)next_epoch(d, S));
  C.W = uni(d.W); C.H = uni(d.H); C.W8 = uni(d.W8); C.N = uni(d.N);
  C.w_magic = uni64(d.w_magic);
  C.gq = (gu64p)(uintptr_t)uni64((u64)(uintptr_t)S.gq);
  C.gd = (gi8p)(uintptr_t)uni64((u64)(uintptr_t)S.gd);
  C.tab = (gu64p)(uintptr_t)uni64((u64)(uintptr_t)S.tab);
  C.amap = (const TS_GLOBAL u64*)(uintptr_t)uni64((u64)(uintptr_t)d.amap);
  C.fovrun = (const TS_GLOBAL u64*)(uintptr_t)uni64((u64)(uintptr_t)d.fovrun);
  C.fov = P.respect_awareness != 0 && d.fovrun != nullptr;
  C.aw = uni(P.vehicle_awareness_range);
  C.density = (const TS_GLOBAL float*)(uintptr_t)uni64((u64)(uintptr_t)d.density);
  C.outg = (gi32p)(uintptr_t)uni64((u64)(uintptr_t)out);
  C.heap_cap = uni(S.heap_cap);
  C.turn_on = P.turn_penalty_enabled != 0; C.rt_on = P.road_type_penalties_enabled != 0;
  C.dens_on = C.soft && P.dynamic_penalties_enabled;
  C.turn_pen = P.turn_penalty; C.contra_pen = P.contraflow_penalty; C.veh_pen = P.obstacle_penalty_vehicle;
  C.stop_pen = P.obstacle_penalty_stop; C.dyn_scale = P.dynamic_penalty_scale; C.rt1 = P.road_type_penalty_r1;
  C.rt2 = P.road_type_penalty_r2; C.rt3 = P.road_type_penalty_r3;
  {
    // half units carry the costs exactly iff every penalty is a non-negative multiple of 0.5 of moderate size
    C.half = astar_half_units(P);
    C.turn2 = (int)(C.turn_pen * 2.0); C.contra2 = (int)(C.contra_pen * 2.0); C.veh2 = (int)(C.veh_pen * 2.0); C.stop2 = (int)(C.stop_pen * 2.0);
    C.rt2_1 = (int)(C.rt1 * 2.0); C.rt2_2 = (int)(C.rt2 * 2.0); C.rt2_3 = (int)(C.rt3 * 2.0);
  }
  C.n_exp = 0; C.n_relax = 0; C.n_exp_spill = 0; C.max_heap = 0;
  for (int k = 0; k < 8; k++) C.prof[k] = 0;
  C.pt = clock64();
  // the chain of relaxations behind a heap entry never revisits a cell (dist strictly falls), so it is shorter than
  // N: a limit of N or more never binds and the steps need not be carried
  C.limited = C.maximum_steps < C.N;
  C.xy_of(C.goal_idx, C.gx, C.gy);
  C.xy_of(C.start_idx, C.sx, C.sy);
  const int sx = C.sx, sy = C.sy;
  if (C.lane == 0) {
    const uint32_t sr = (uint32_t)(C.amap[C.tile_ix(sx, sy)] >> 32);
    if (sr != 0xFFFFFFFFu) C.tab[sr] = (u64)0u | ((u64)(C.epoch << T_STAMP_SHIFT) << 32);    // dist 0, steps 0
    g_lq[0] = hq_pack(abs(sx - C.gx) + abs(sy - C.gy), C.start_idx);
    g_ld[0] = -1;
  }
  int heap_size = 1;
  wave_mem_sync();
  int r;
  // (the default policy is the <HALF, no FOV> pair; the other instantiations serve fractional penalties and
  // VEHICLE_RESPECT_AWARENESS)
  auto run = [&](auto half_c, auto fov_c) {
    constexpr bool HF = decltype(half_c)::value, FV = decltype(fov_c)::value;
    for (;;) {
      int q = astar_loop<false, HF, FV>(C, heap_size);
      if (q != AL_SWITCH) return q;
      q = astar_loop<true, HF, FV>(C, heap_size);
      if (q != AL_SWITCH) return q;
    }
  };
  if (!C.fov) r = C.half ? run(std::true_type{}, std::false_type{}) : run(std::false_type{}, std::false_type{});
  else r = C.half ? run(std::true_type{}, std::true_type{}) : run(std::false_type{}, std::true_type{});
  S.expansions += C.n_exp; S.relaxations += C.n_relax;
  if (C.lane == 0) {   // profiling aid: deepest heap / longest search any searcher has seen (ts_debug_read words 4, 5)
    atomicMax(&d.cnt->dbg[4], C.max_heap);
    atomicMax(&d.cnt->dbg[5], (int)min(C.n_exp, (long long)0x7FFFFFFF));
    if (C.n_exp_spill) atomicAdd((unsigned long long*)&d.cnt->dbg[6], (unsigned long long)C.n_exp_spill);   // (dbg[6..7] as one 64-bit count)
#ifdef TS_KPROF
    for (int k = 0; k < 8; k++) d.cnt->prof[k] = C.prof[k];
#endif
  }
  return r == AL_EMPTY ? 0 : r;
}

// Strict reachability of `goal` from `start`: a frontier BFS by the wave over the same edges the strict A* relaxes
// (flow bit set, neighbour in bounds, neither occupied nor red), 64 cells per step.  Unreachable targets are by far
// the most expensive searches (the sequential search floods the whole component before returning []); knowing the
// answer lets phase 1 skip them.  Visited marks are stamps of a fresh epoch in the searcher's own table, the queue is
// a ring in its heap spill area.  Returns 1 reachable, 2 not, 0 unknown (ring overflow).
__device__ int reach_strict_wave(const Dev& d, AScratch& S, int start, int goal) {
  const int lane = lane_id();
  start = uni(start); goal = uni(goal);
  const uint32_t stamp = (uint32_t)uni((int)next_epoch(d, S)) << T_STAMP_SHIFT;
  const int W = uni(d.W), H = uni(d.H), W8 = uni(d.W8);
  const u64 w_magic = uni64(d.w_magic);
  const gi32p ring = (gi32p)(uintptr_t)uni64((u64)(uintptr_t)S.gq);
  TS_GLOBAL uint32_t* const tabw = (TS_GLOBAL uint32_t*)(uintptr_t)uni64((u64)(uintptr_t)S.tab);   // record t: dist at 2t, meta at 2t + 1
  const TS_GLOBAL u64* amap = (const TS_GLOBAL u64*)(uintptr_t)uni64((u64)(uintptr_t)d.amap);
  const unsigned qcap = (unsigned)uni(S.heap_cap - LDS_HEAP) * 2u;
  if (qcap < 256u) return 0;
  auto xy_of = [&](int cell, int& x, int& y) {
    if (w_magic) { y = (int)(((u64)(unsigned)cell * w_magic) >> 40); x = cell - y * W; }
    else { y = cell / W; x = cell - y * W; }
  };
  auto tile_ix = [&](int x, int y) -> uint32_t {
    return ((((uint32_t)(y >> 3) * (uint32_t)W8 + (uint32_t)(x >> 3)) << 6) | (uint32_t)((y & 7) << 3) | (uint32_t)(x & 7));
  };
  if (lane == 0) {
    int sx, sy;
    xy_of(start, sx, sy);
    ring[0] = start;
    const uint32_t sr = (uint32_t)(amap[tile_ix(sx, sy)] >> 32);
    if (sr != 0xFFFFFFFFu) tabw[2 * (size_t)sr + 1] = stamp;
  }
  __syncthreads();
  unsigned head = 0, tail = 1;
  bool found = false;
  while (head < tail && !found) {
    const unsigned idx = head + (unsigned)lane;
    const int c = idx < tail ? ring[idx % qcap] : -1;
    head = min(tail, head + 64u);
    int cx = 0, cy = 0;
    if (c >= 0) xy_of(c, cx, cy);
    const uint32_t bits = c >= 0 ? (uint32_t)amap[tile_ix(cx, cy)] & 15u : 0u;
    for (int dd = 0; dd < 4; dd++) {
      int n = -1;
      if (bits & (1u << dd)) {
        const int nx = cx + (dd == 1) - (dd == 3), ny = cy + (dd == 0) - (dd == 2);
        if (nx >= 0 && nx < W && ny >= 0 && ny < H) {
          const u64 an = amap[tile_ix(nx, ny)];
          const uint32_t rn = (uint32_t)(an >> 32);
          if (((uint32_t)an & 0x300u) == 0u && rn != 0xFFFFFFFFu &&
              __hip_atomic_exchange(&tabw[2 * (size_t)rn + 1], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != stamp)
            n = ny * W + nx;
        }
      }
      const unsigned long long m = __ballot(n >= 0);
      const unsigned cnt = (unsigned)__popcll(m);
      if (tail + cnt - head > qcap) return 0;
      if (n >= 0) ring[(tail + (unsigned)__popcll(m & ((1ULL << lane) - 1))) % qcap] = n;
      if (__ballot(n == goal && n >= 0)) found = true;
      tail += cnt;
    }
    __syncthreads();  // ring writes of this step are read by the next one
  }
  return found ? 1 : 2;
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
  bool reach_known;  // a replan inside the decide phase: phase 1 asks reach_strict_wave first
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
// DM_WAVE: the caller runs with all 64 lanes of its wave executing the same code on the same values (one vehicle per
// wave); plain stores are then harmless duplicates, atomics are issued by lane 0 only, and the searches use
// astar_wave.  DM_QUAD: the same with the four lanes of a quad.  DM_LANE: one vehicle per lane (k_decide_main), nothing is shared.
constexpr int QLOG = 8;   // searches one step_decide can make in quad mode (beyond that the vehicle goes to k_replan)
template <int MODE>
__device__ __forceinline__ int astar_any(const Dev& d, const TsParams& P, AScratch& S, int start_idx, int goal_idx, bool soft,
                                         bool ignore_flow, int maximum_steps, int32_t* out, int out_cap) {
  if constexpr (MODE == DM_WAVE) return astar_wave(d, P, S, start_idx, goal_idx, soft, ignore_flow, maximum_steps, out, out_cap);
  else if constexpr (MODE == DM_QUAD) {
    const int k = S.q_replay++;
    if (k < S.q_done) {   // finished in an earlier pass: its path already sits in `out`
      S.calls++; S.expansions += S.q_log[3 * k + 1]; S.relaxations += S.q_log[3 * k + 2];
      return S.q_log[3 * k];
    }
    // the quad searcher carries neither step limits nor contraflow (bypass searches are rare and small): such a vehicle is
    // handed to k_replan, as is one that searches more often than the log is long
    if (maximum_steps < d.N || ignore_flow || k >= QLOG) { S.q_status = DV_BAIL; return -1; }
    S.q_start = start_idx; S.q_goal = goal_idx; S.q_soft = soft ? 1 : 0; S.q_out = out; S.q_cap = out_cap;
    S.q_status = DV_SUSPEND;
    return -1;
  }
  else return -1;   // (one vehicle per lane never searches: decide_vehicle<DM_LANE> defers before it gets here)
}
template <int MODE>
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
    int bl = astar_any<MODE>(d, P, S, v.pos, b, false, true, P.max_contraflow_overtake_steps, S.BYP, MAXB);
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
  bool unreachable = false;
  if constexpr (MODE == DM_WAVE) unreachable = v.reach_known && reach_strict_wave(d, S, v.pos, sx_goal) == 2;
  if (unreachable) {
    // the frontier BFS proved the target unreachable under the strict rules: the search would flood its whole
    // component and return [] (astar_numba.py:239).  Same result, without the flood.
    S.calls++;
    la = 0;
  } else {
    la = astar_any<MODE>(d, P, S, v.pos, sx_goal, false, false, 0x7FFFFFFF, S.A, S.cap);
    if (la < 0) return false;
  }
  if (la == 0) {
    la = astar_any<MODE>(d, P, S, v.pos, sx_goal, true, false, 0x7FFFFFFF, S.A, S.cap);
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
          int bl = astar_any<MODE>(d, P, S, v.pos, bt, false, true, P.max_contraflow_overtake_steps, S.BYP, MAXB);
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
        int bl = astar_any<MODE>(d, P, S, v.pos, bt, true, true, P.max_contraflow_stuck_detour_steps, S.BYP, MAXB);
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
// (quad_perm [0,0,0,0]: every lane of a quad reads its lane 0)
__device__ __forceinline__ int quad_first(int v) { return __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, true); }
template <int MODE>
__device__ __forceinline__ bool pool_alloc(const Dev& d, int words, uint32_t& off) {
  unsigned long long o = 0;
  const bool one = MODE == DM_LANE || (MODE == DM_WAVE ? lane_id() == 0 : (lane_id() & 3) == 0);
  if (one) o = atomicAdd((unsigned long long*)&d.cnt->pool_used, (unsigned long long)words);
  if (MODE == DM_WAVE) o = ((unsigned long long)(unsigned)__shfl((int)(o >> 32), 0) << 32) | (unsigned)__shfl((int)(unsigned)o, 0);
  if (MODE == DM_QUAD) o = ((unsigned long long)(unsigned)quad_first((int)(o >> 32)) << 32) | (unsigned)quad_first((int)(unsigned)o);
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
template <int MODE>
__device__ int decide_vehicle(const Dev& d, const TsParams& P, int i, AScratch* S) {
  const bool one = MODE == DM_LANE || (MODE == DM_WAVE ? lane_id() == 0 : (lane_id() & 3) == 0);   // the lane that issues this vehicle's atomics
  const int vid = d.active[i];
  if (vid < 0) return DV_DONE;
  VW v;
  v.vid = vid; v.i = i; v.pos = d.pos[vid]; v.target = d.target[vid];
  v.reach_known = S != nullptr && S->use_reach;
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
      if (!compute_path_internal_dev<MODE>(d, P, *S, v, len)) return MODE == DM_QUAD ? S->q_status : DV_OVERFLOW;
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
      if (!compute_path_internal_dev<MODE>(d, P, *S, v, len)) return MODE == DM_QUAD ? S->q_status : DV_OVERFLOW;
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
    if (words > 0 && !pool_alloc<MODE>(d, words, off)) return DV_POOL_FULL;
    uint8_t chg = 0;
    if (path_changed) {
      encode_cells(d, off, v.pos, S->P, v.plen);
      d.path_off[vid] = off; d.path_len[vid] = v.plen; d.path_cur[vid] = 0;
      off += (v.plen + 15) / 16;
      chg |= 1;
    }
    for (int k = 0; k < 4; k++) {
      if (!v.ax_staged[k]) continue;
      encode_cells(d, off, v.pos, ax_buf(*S, k), v.ax_len[k]);
      d.ax_start[k][vid] = v.pos; d.ax_off[k][vid] = off; d.ax_len[k][vid] = v.ax_len[k];
      off += (v.ax_len[k] + 15) / 16;
      chg |= (uint8_t)(2 << k);
    }
    if (chg) d.chg[vid] = chg;
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
    if (!(v.f & VF_KEEP)) {   // a trip that ends where it starts: _despawn inside step_decide.  The host ends the stretch of
      // the decide order at such a vehicle and takes it off the maps once everybody before it is through (tick())
      if (d.dec_expect == i + 1) atomicExch(&d.cnt->dec_arrived, i + 1);
      else atomicExch(&d.cnt->error, TS_E_DEVICE);
    }
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
__global__ void k_decide_main(Dev d, TsParams P, int lo, int n_active, RLists lists) {
  int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  if (d.cnt->rng_event != 0xFFFFFFFFu) return;  // a malfunction / sideswipe fired: the host re-runs this after the fix-up
  if (decide_vehicle<DM_LANE>(d, P, i, nullptr) == DV_DEFER) {
    // work-queue class (largest first): what the vehicle's last replan cost, or what a search over this distance is
    // likely to cost
    const int h = cost_class_of_bits(replan_cost_bits(d, d.active[i]));
    lists.l[h][atomicAdd(&d.cnt->replan_n[h], 1)] = i;
  }
}

// sort key of a replanning entry (run_replans): expected cost, largest first (bit length of the expansions, see cost_bits),
// then the Morton index of the 32 x 32-cell block its vehicle stands in
constexpr int REPLAN_KEY_BITS = 21;
// The same with the entry itself (its decide-order index) below the key: a total order, the same on every rank of a sharded
// run whatever order k_decide_main's atomics left the list in, so that ranks can split the queue by POSITION (entry j of
// the sorted queue belongs to rank j % world: every rank gets every world-th search of every cost class and every
// neighbourhood - the longest searches are dealt out one by one instead of falling where index % world puts them).
__global__ void k_replan_keys64(Dev d, const int32_t* list, int n, unsigned long long* keys) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int i = list[j];
  const int vid = d.active[i];
  int x = 0, y = 0, bits = 0;
  if (vid >= 0) { cell_xy(d, d.pos[vid], x, y); bits = min(replan_cost_bits(d, vid), 31); }
  uint32_t bx = (uint32_t)x >> 5, by = (uint32_t)y >> 5, k = 0;
  for (int b = 0; b < 8; b++) k |= ((bx >> b) & 1u) << (2 * b) | ((by >> b) & 1u) << (2 * b + 1);
  keys[j] = ((unsigned long long)(((uint32_t)(31 - max(bits, 16)) << 16) | k) << 32) | (unsigned long long)(uint32_t)i;
}
__global__ void k_replan_unkey64(const unsigned long long* keys, int n, int32_t* list) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) list[j] = (int32_t)(uint32_t)keys[j];
}
__global__ void k_replan_keys(Dev d, const int32_t* list, int n, uint32_t* keys) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int vid = d.active[list[j]];
  int x = 0, y = 0, bits = 0;
  if (vid >= 0) { cell_xy(d, d.pos[vid], x, y); bits = min(replan_cost_bits(d, vid), 31); }
  uint32_t bx = (uint32_t)x >> 5, by = (uint32_t)y >> 5, k = 0;
  for (int b = 0; b < 8; b++) k |= ((bx >> b) & 1u) << (2 * b) | ((by >> b) & 1u) << (2 * b + 1);
  // (only the long searches are ordered by cost - 65 536 expansions and more, bit by bit; the bulk stays in plain spatial order)
  keys[j] = ((uint32_t)(31 - max(bits, 16)) << 16) | k;
}

// One turn of a searcher wave at the replanning work queue: take the next entry, run the vehicle's step_decide with
// all 64 lanes, account for it.  Returns 0 once the queue is empty.  Kept out of line on purpose: inlined into
// k_replan's loop, hipcc 7.2 threaded the lane-0-only parts (queue pop, accounting) of consecutive turns together and
// let lane 0 run the loop on a path of its own, apart from the other 63 lanes - wrong for code whose lanes cooperate
// through readlane / ballot.  A call boundary is a point where the wave is whole again.
#ifdef TS_TRACE_REPLAN
// profiling builds (profiles/replan_trace.py): per queue entry (start, end: low words of the 100 MHz clock; expansions;
// predicted cost bits | searcher slot << 8)
__device__ int4* g_rtrace = nullptr;
__device__ int g_rtrace_cap = 0;
#endif
struct RQueue {
  int32_t* l[4]; int n[4]; int32_t *retry_list, *owned_list; int rank, world;
  // vehicles k_replan_quad (astar_quad.h) hands back while both kernels run: entries appear in fb_list (-1 = not yet
  // written; tickets from quad_n[2]) until all fb_waves quad waves have counted themselves out in quad_n[3]
  int32_t* fb_list; int fb_waves, fb_cap;   // fb_cap: entries the quads were given = the most they can hand back
};
__device__ __attribute__((noinline)) int replan_turn(const Dev& d, const TsParams& P, AScratch* S, const RQueue& q) {
  const int n3 = uni(q.n[3]), n2 = uni(q.n[2]), n1 = uni(q.n[1]), n0 = uni(q.n[0]);
  if (threadIdx.x == 0) g_job = atomicAdd(&d.cnt->replan_n[5], 1);
  __syncthreads();
  const int j = uni(g_job);
  __syncthreads();
  int i;
  if (j >= n3 + n2 + n1 + n0) {
    if (q.fb_waves == 0) return 0;
    // this launch's own lists are done: serve the hand-back list.  An entry is only ever claimed once it has been produced
    // (claimed <= produced at all times), so whatever is left when this wave gives up is a suffix the host can queue again.
    // Giving up: all quad waves have counted themselves out, or nothing has moved for about three seconds (the two kernels
    // were not run side by side - a profiler or debugger serialising launches; k_replan_quad is then yet to run).
    if (threadIdx.x == 0) {
      int job = -1;
      long long t_last = wall_clock64();
      int seen = -1;
      for (;;) {
        const int produced = __hip_atomic_load(&d.cnt->quad_n[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        int claimed = __hip_atomic_load(&d.cnt->quad_n[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (claimed < produced) {
          if (__hip_atomic_compare_exchange_strong(&d.cnt->quad_n[2], &claimed, claimed + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            // (the producer stores the entry right after counting it: a running wave, a few hundred cycles at most)
            do job = __hip_atomic_load(&q.fb_list[claimed], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); while (job < 0);
            break;
          }
          continue;
        }
        const int done = __hip_atomic_load(&d.cnt->quad_n[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (done >= q.fb_waves) {
          if (__hip_atomic_load(&d.cnt->quad_n[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == produced) break;   // nothing more can come
          continue;
        }
        const int mark = produced + done + __hip_atomic_load(&d.cnt->quad_n[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long now = wall_clock64();
        if (mark != seen) { seen = mark; t_last = now; }
        else if (now - t_last > 300000000ll) break;          // 3 s of the 100 MHz clock
        __builtin_amdgcn_s_sleep(64);
      }
      g_job = job;
    }
    __syncthreads();
    i = uni(g_job);
    __syncthreads();
    if (i < 0) return 0;
  }
  else if (j < n3) i = q.l[3][j];
  else if (j < n3 + n2) i = q.l[2][j - n3];
  else if (j < n3 + n2 + n1) i = q.l[1][j - n3 - n2];
  else i = q.l[0][j - n3 - n2 - n1];
  i = uni(i);
  // (sharded mode: the queue is in the same total order on every rank - run_replans sorts it by (key, index) - and entry j
  // of it is rank j % world's; hand-backs of this rank's own quads are this rank's)
  if (q.world > 1 && j < n3 + n2 + n1 + n0 && (j % q.world) != q.rank) return 1;
  // the most expensive classes are a tick's critical path (its longest search bounds it): their waves take the issue slots
  // of their SIMD first, the five waves beside them fill in behind (`s_setprio`; TS_NO_PRIO: a build without it)
#ifndef TS_NO_PRIO
  if (j < n3) __builtin_amdgcn_s_setprio(3);
  else if (j < n3 + n2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(0);
#endif
  const long long c0 = S->calls, e0 = S->expansions, r0 = S->relaxations;
#ifdef TS_TRACE_REPLAN
  const long long tr0 = wall_clock64();
  const int tr_bits = replan_cost_bits(d, max(d.active[i], 0));
#endif
  const int r = uni(decide_vehicle<DM_WAVE>(d, P, i, S));
#ifdef TS_TRACE_REPLAN
  if (threadIdx.x == 0 && g_rtrace && j < g_rtrace_cap && j < n3 + n2 + n1 + n0)
    g_rtrace[j] = make_int4((int)(unsigned)tr0, (int)(unsigned)wall_clock64(), (int)(S->expansions - e0), tr_bits | ((int)blockIdx.x << 8));
#endif
  if (threadIdx.x == 0) {
    if (r == DV_DONE) {  // work of attempts that are re-run after pool growth is not counted twice
      const int vid = d.active[i];
      if (S->calls > c0) d.tier_hint[vid] = (uint8_t)cost_bits(S->expansions - e0);
      atomicAdd((unsigned long long*)&d.cnt->astar_calls, (unsigned long long)(S->calls - c0));
      atomicAdd((unsigned long long*)&d.cnt->astar_exp, (unsigned long long)(S->expansions - e0));
      atomicAdd((unsigned long long*)&d.cnt->astar_relax, (unsigned long long)(S->relaxations - r0));
      if (q.owned_list) q.owned_list[atomicAdd(&d.cnt->replan_n[6], 1)] = i;
    } else if (r == DV_OVERFLOW) atomicExch(&d.cnt->error, TS_E_CAPACITY);
    else if (r == DV_POOL_FULL) q.retry_list[atomicAdd(&d.cnt->replan_n[4], 1)] = i;
  }
  return 1;
}

// Replanning vehicles: a work queue served by one wave per searcher slot.  The queue is the four class lists, the
// most expensive class first (a tick's replanning time is bounded below by its longest search: start those first and
// let the short ones fill in behind).  Every wave takes the next entry until the queue is empty; all 64 lanes run
// the vehicle's step_decide together and share the work inside the searches.  Entries that find the path pool
// full go to `retry_list` (counter replan_n[4]); replan_n[5] is the queue cursor.
// `world` > 1: the replicated-state multi-GPU mode - this rank plans only the vehicles whose decide-order index is
// congruent to `rank`; the results travel through ts_replan_export / ts_replan_import.
// `class_mask`: the class lists this launch serves (the others belong to k_replan_quad, astar_quad.h).
TS_REPLAN_OCC __global__ void __launch_bounds__(64) k_replan(Dev d, TsParams P, ASlots sl, RLists lists, int32_t* retry_list, int rank, int world,
                                               int32_t* owned_list, int class_mask, int32_t* fb_list, int fb_waves, int fb_cap) {
  AScratch S;
  scratch_bind(sl, blockIdx.x, S);
  RQueue q;
  for (int c = 0; c < 4; c++) { q.l[c] = lists.l[c]; q.n[c] = ((class_mask >> c) & 1) ? d.cnt->replan_n[c] : 0; }
  q.retry_list = retry_list; q.owned_list = owned_list; q.rank = rank; q.world = world;
  q.fb_list = fb_list; q.fb_waves = fb_waves; q.fb_cap = fb_cap;
  while (uni(replan_turn(d, P, &S, q))) {}
  if (threadIdx.x == 0) sl.slot_epoch[blockIdx.x] = S.epoch;
}

// ---- replicated-state multi-GPU mode (ts_set_replan_sharding) ------------------------------------------------------
// What step_decide changed about a vehicle this rank planned, for the ranks that did not: one fixed record plus the
// 2-bit direction words of whatever paths the replan rewrote.
struct ReplanRec {
  int32_t i, vid, flags, base, cur, max_steps, cooldown, over_dur, det_dur, stranded_left, hint;
  int32_t path_len, path_woff;          // path_woff < 0: the path was left as it is
  int32_t ax_len[4], ax_start[4], ax_woff[4];
  int32_t pad_[3];
};
static_assert(sizeof(ReplanRec) == 112, "ReplanRec is exchanged as 28 ints");
// `count_only`: add up the words the export will need and touch nothing else (the host sizes the word buffer with it: the
// pool's growth over the phase is no bound once a garbage collection ran inside it).
__global__ void k_replan_export(Dev d, const int32_t* owned, int n, ReplanRec* recs, uint32_t* words, unsigned long long* words_n,
                                int count_only) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = owned[t];
  const int vid = d.active[i];
  if (count_only) {
    if (vid < 0) return;
    const uint8_t chg = d.chg[vid];
    unsigned long long nw = (chg & 1) ? (unsigned long long)((d.path_len[vid] + 15) >> 4) : 0ull;
    for (int k = 0; k < 4; k++) if ((chg >> (1 + k)) & 1) nw += (unsigned long long)((d.ax_len[k][vid] + 15) >> 4);
    if (nw) atomicAdd(words_n, nw);
    return;
  }
  ReplanRec r;
  r.i = i; r.vid = vid;
  r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
  if (vid < 0) { recs[t] = r; return; }
  r.flags = d.flags[vid]; r.base = d.base_speed[vid]; r.cur = d.cur_speed[vid]; r.max_steps = d.max_steps[vid];
  r.cooldown = d.cooldown[vid]; r.over_dur = d.over_dur[vid]; r.det_dur = d.det_dur[vid];
  r.stranded_left = d.stranded_left[vid]; r.hint = d.tier_hint[vid];
  const uint8_t chg = d.chg[vid];
  d.chg[vid] = 0;
  r.path_len = d.path_len[vid]; r.path_woff = -1;
  if (chg & 1) {
    const int nw = (r.path_len + 15) >> 4;
    const unsigned long long w = atomicAdd(words_n, (unsigned long long)nw);
    const uint32_t src = d.path_off[vid];
    for (int q = 0; q < nw; q++) words[w + q] = d.pool[src + q];
    r.path_woff = (int32_t)w;
  }
  for (int k = 0; k < 4; k++) {
    r.ax_len[k] = d.ax_len[k][vid]; r.ax_start[k] = d.ax_start[k][vid]; r.ax_woff[k] = -1;
    if ((chg >> (1 + k)) & 1) {
      const int nw = (r.ax_len[k] + 15) >> 4;
      const unsigned long long w = atomicAdd(words_n, (unsigned long long)nw);
      const uint32_t src = d.ax_off[k][vid];
      for (int q = 0; q < nw; q++) words[w + q] = d.pool[src + q];
      r.ax_woff[k] = (int32_t)w;
    }
  }
  recs[t] = r;
}
// the same in the other direction: records of vehicles another rank planned (pool capacity ensured by the host)
__global__ void k_replan_import(Dev d, const ReplanRec* __restrict__ recs, int n, const uint32_t* __restrict__ words) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const ReplanRec r = recs[t];
  const int vid = r.vid;
  if (vid < 0) return;
  d.flags[vid] = (uint16_t)r.flags; d.base_speed[vid] = (int8_t)r.base; d.cur_speed[vid] = (int8_t)r.cur;
  d.max_steps[vid] = (int8_t)r.max_steps; d.cooldown[vid] = r.cooldown; d.over_dur[vid] = r.over_dur; d.det_dur[vid] = r.det_dur;
  d.stranded_left[vid] = r.stranded_left; d.tier_hint[vid] = (uint8_t)r.hint;
  if (r.path_woff >= 0) {
    const int nw = (r.path_len + 15) >> 4;
    const uint32_t o = (uint32_t)atomicAdd((unsigned long long*)&d.cnt->pool_used, (unsigned long long)nw);
    for (int q = 0; q < nw; q++) d.pool[o + q] = words[r.path_woff + q];
    d.path_off[vid] = o; d.path_len[vid] = r.path_len; d.path_cur[vid] = 0;
  }
  for (int k = 0; k < 4; k++) {
    d.ax_len[k][vid] = r.ax_len[k];
    if (r.ax_woff[k] >= 0) {
      const int nw = (r.ax_len[k] + 15) >> 4;
      const uint32_t o = (uint32_t)atomicAdd((unsigned long long*)&d.cnt->pool_used, (unsigned long long)nw);
      for (int q = 0; q < nw; q++) d.pool[o + q] = words[r.ax_woff[k] + q];
      d.ax_start[k][vid] = r.ax_start[k]; d.ax_off[k][vid] = o;
    }
  }
}

// one search on the current maps (the `astar(...)` operator seam, ts_astar) - searcher slot 0
TS_REPLAN_OCC __global__ void __launch_bounds__(64) k_astar_single(Dev d, TsParams P, ASlots sl, int start_idx, int goal_idx, int soft,
                                                      int ignore_flow, int maximum_steps, int32_t* out_len) {
  if (blockIdx.x) return;
  AScratch S;
  scratch_bind(sl, 0, S);
  const long long c0 = clock64(), w0 = wall_clock64();
  int len = astar_wave(d, P, S, start_idx, goal_idx, soft != 0, ignore_flow != 0, maximum_steps, S.A, S.cap);
  if (threadIdx.x) return;
  {  // probe figures (profiles/astar_probe.py): shader cycles and 100 MHz wall ticks the search took
    const long long dc = clock64() - c0, dw = wall_clock64() - w0;
    d.cnt->dbg[0] = (int)(dc & 0xFFFFFFFF); d.cnt->dbg[1] = (int)(dc >> 32); d.cnt->dbg[2] = (int)(dw & 0xFFFFFFFF); d.cnt->dbg[3] = (int)(dw >> 32);
  }
  sl.slot_epoch[0] = S.epoch;
  if (len >= 0) {
    atomicAdd((unsigned long long*)&d.cnt->astar_calls, (unsigned long long)S.calls);
    atomicAdd((unsigned long long*)&d.cnt->astar_exp, (unsigned long long)S.expansions);
    atomicAdd((unsigned long long*)&d.cnt->astar_relax, (unsigned long long)S.relaxations);
  }
  *out_len = len;  // -1 = heap / output capacity exceeded; the path cells are in the slot's A buffer
}

// VehicleAgent.__init__ -> self.path = self._compute_path() on a cache miss (vehicle_base.py:80-81, 143-167):
// the phase 0-4 planner for a freshly placed vehicle.  status: path length, or -1 overflow / -2 pool full.
TS_REPLAN_OCC __global__ void __launch_bounds__(64) k_spawn_plan(Dev d, TsParams P, ASlots sl, int vid, int32_t* status) {
  if (blockIdx.x) return;
  const bool one = threadIdx.x == 0;
  AScratch S;
  scratch_bind(sl, 0, S);
  VW v;
  v.vid = vid; v.i = LAST_IDX; v.pos = d.pos[vid]; v.target = d.target[vid];
  v.f = d.flags[vid]; v.base = 0; v.cur = 0; v.cooldown = P.pathfinding_cooldown;
  v.over_dur = d.over_dur[vid]; v.det_dur = d.det_dur[vid]; v.stuck_ticks = d.stuck_ticks[vid];
  v.newpath = false; v.plen = 0; v.pcur = 0; v.off = 0; v.d_overtaking = 0; v.d_detour = 0;
  v.reach_known = false;
  for (int k = 0; k < 4; k++) { v.ax_staged[k] = false; v.ax_len[k] = d.ax_len[k][vid]; }
  int len;
  bool ok = compute_path_internal_dev<DM_WAVE>(d, P, S, v, len);
  if (one) sl.slot_epoch[0] = S.epoch;
  if (!ok) { if (one) *status = -1; return; }
  int words = (len + 15) / 16;
  for (int k = 0; k < 4; k++) if (v.ax_staged[k]) words += (v.ax_len[k] + 15) / 16;
  uint32_t off = 0;
  if (words > 0 && !pool_alloc<DM_WAVE>(d, words, off)) { if (one) *status = -2; return; }
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
  if (one) *status = len;
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

}  // namespace
