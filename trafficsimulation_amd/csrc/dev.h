// dev.h - device-side state (structure-of-arrays in HBM) and small helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/trafficsim.h"

#define BLK 256

namespace {

// vehicle flag bits: TS_F_* (1..256) plus engine-private ones
constexpr uint16_t VF_EARLY = TS_F_EARLY_EXIT, VF_STUCK = TS_F_STUCK, VF_PARKED = TS_F_PARKED,
                   VF_COLL = TS_F_COLLISION, VF_MALF = TS_F_MALFUNCTION, VF_OVER = TS_F_OVERTAKING,
                   VF_DETOUR = TS_F_DETOUR, VF_BLOCKED = TS_F_BLOCKED, VF_HASPREV = TS_F_HAS_PREV,
                   VF_KEEP = 512 /* remove_on_arrival == False */, VF_ALIVE = 1024,
                   // ServiceVehicleAgent (vehicle_service.py): phase "to_block" = SVC|TOBLOCK, "servicing" = SVC|SERVICING,
                   // "to_exit" = SVC alone
                   VF_SVC = 2048, VF_SERVICING = 4096, VF_TOBLOCK = 8192,
                   // PATHFINDING_BATCHING=False only: _start_service ran inside this vehicle's own step_decide, i.e. inside its step():
                   // the rest of that step() still runs (tick_stuck, the second on_target_reached: vehicle_base.py:678-685) before
                   // the vehicle counts as "servicing" (whose step() returns at its top, vehicle_service.py:43-49)
                   VF_SVCNEW = 16384;
constexpr int LAST_IDX = 0x7FFFFFFF;  // "decides after everyone": planners that run outside the decide phase
// service records the kernels hand to the host (key, vehicle, what)
constexpr int AR_START = 0 /* _start_service in the move phase, key = rank */, AR_DECIDE = 1 /* on_target_reached inside
                   step_decide, key = decide-order index: applied by k_decide_arrive */, AR_DESPAWN = 2 /* service vehicle left */;
constexpr int8_t K_VEHICLE = 100, K_DEAD = -1, K_RAIN = 5;  // K_RAIN: a RainAgent's schedule entry
constexpr uint32_t RANK_BITS = 24, RANK_MASK = (1u << RANK_BITS) - 1, EPOCHS = 1u << (32 - RANK_BITS);
constexpr uint32_t NO_RANK = 0xFFFFFFFFu;

// decide-phase flag byte F (k_decide_pre -> host scan)
constexpr uint8_t F_DRAW_MALF = 1, F_DRAW_SWIPE = 2, F_DRAW_SPEED = 4;
constexpr uint32_t WORDS_MASK = (1u << 23) - 1;  // = MTPipe::TW_CAP - 1

struct DevCnt {
  long long stuck, collisions, malfunctions, overtaking, in_stuck_detour, parked, live_internal, live_through,
      completed_internal, completed_through, dist_internal, dist_through;
  double dur_internal, dur_through;
  int resolved;    // agents stepped so far in this move phase
  int deaths;      // vehicles removed this tick
  int arr_n;       // service records written this tick (Dev::arr)
  int error;       // sticky device-side error
  int replan_n[8]; // replanning work queue: [0..3] class list lengths, [4] pool-full retries, [5] queue cursor, [6] entries this rank planned
  int quad_n[4];   // k_replan_quad: [0] vehicles handed back to k_replan, [1] its queue cursor
  unsigned long long pool_used;  // words handed out from the path pool (device-side bump allocator)
  long long astar_calls, astar_exp, astar_relax;
  long long errored_internal, errored_through;   // _despawn_check removals
  int pend_n[2];   // lengths of the two ping-pong lists of still-unresolved schedule slots
  unsigned int rng_event;   // first (vehicle index * 2 + is_collision) whose draw fired this pass, 0xFFFFFFFF = none
  unsigned int rng_tot[2];  // pass 1 totals: fixed words, number of speed rolls
  int dec_arrived; // 1 + decide index of the vehicle that despawned inside this stretch of the decide phase (0 = none)
  long long qprof[8];  // TS_QUAD_PROF builds: cycles per segment of k_replan_quad's turn (lane 0 of every wave)
  long long prof[8];   // TS_KPROF builds: cycles per segment of the last search's loop
  int dbg[8];      // debugging aid: first watchdog that fired inside a replanning kernel (code, vehicle index, values)
};

// Everything the hot kernels read or write about one grid cell, in one 32-byte sector: the four move-phase claim
// words (plane 0 = writers of occupancy, 1 = readers of occupancy, 2 = writers of stop, 3 = readers of stop), the head
// of the cell's MultiGrid vehicle list, the dynamic bytes and the packed static byte.  Every field is written with
// its own byte / dword store (agents that run in the same round touch different fields), never read-modify-write.
struct __attribute__((aligned(32))) Cell {
  uint32_t claim[4];
  int32_t veh;                 // first vehicle in the cell's list, -1 = none
  int8_t occ, stop, stuck;     // occupancy_map / stop_map / stuck_map
  uint8_t stat;                // allowed_dirs (bits 0-3) | is_road << 4 | intersection << 5 | road_type << 6
  uint32_t pad_[2];
};
__host__ __device__ __forceinline__ int st_allowed(uint8_t s) { return s & 15; }
__host__ __device__ __forceinline__ int st_is_road(uint8_t s) { return (s >> 4) & 1; }
__host__ __device__ __forceinline__ int st_inter(uint8_t s) { return (s >> 5) & 1; }
__host__ __device__ __forceinline__ int st_road_type(uint8_t s) { return (s >> 6) & 3; }

struct Dev {
  int W, H, N;
  int W8, H8;                   // the map in 8 x 8 tiles (tiled order of the A* snapshot and tables, see tix)
  unsigned long long w_magic;   // floor(2^40 / W) + 1: y = (cell * w_magic) >> 40 is exact for cell < 2^26, W < 2^14
  double elapsed;   // DynamicTrafficAgent.elapsed as the decide phase sees it (before the clock agent steps)
  int dec_expect;   // 1 + decide index of the one vehicle that may despawn inside this stretch of the decide phase (0 = none)
  int seq;          // PATHFINDING_BATCHING=False: every vehicle decides alone, at its turn of the shuffled order - what is stored about
                    // the other vehicles IS their state at that point (none of them is half-way through a decide phase)
  Cell* cell;
  // dense byte planes kept next to the records: occ and stop are written through (light groups sum lanes of
  // occupancy, the density map and the host read whole planes), rain is only ever read at a vehicle's own cell
  int8_t *occ, *stop, *rain;
  int8_t* is_road;  // static plane for the density map
  // vehicles (indexed by vehicle id = spawn index)
  int32_t *pos, *target, *path_len, *path_cur, *stuck_ticks, *cooldown, *stranded_left, *steps, *over_dur, *det_dur,
      *next_in_cell, *active_idx, *sched_slot;
  uint32_t* path_off;
  int8_t *base_speed, *cur_speed, *max_steps, *dir, *pop;
  uint16_t* flags;
  double* depart;
  uint8_t *ev, *st_before, *st_after;
  int32_t* ev_idx;  // decide-order index of the event that stranded this vehicle this tick (valid when ev != 0)
  uint32_t* pool;
  unsigned long long pool_cap_words;
  // contraflow aux paths, k: 0 overtake_path, 1 pre_overtake_path, 2 stuck_detour_path, 3 pre_stuck_detour_path
  // (vehicle_base.py:45-54) as (cell before the first element, pool offset, length; -1 = None)
  int32_t* ax_start[4];
  uint32_t* ax_off[4];
  int32_t* ax_len[4];
  uint8_t* tier_hint;  // per vehicle: bit length of the expansions its last replan took (astar.h: cost_bits), orders the replanning work queue
  uint8_t* chg;        // per vehicle: what its replan of this tick rewrote (bit 0 path, bits 1-4 aux paths) - read and cleared by
                       // k_replan_export in the multi-GPU mode
  // what a search reads about a cell, as of the last tick start (or the last ensure_amap), in tiled order: low word =
  // static byte | occupied << 8 | red << 9, high word = the cell's search-node number (0xFFFFFFFF: not a node).  Nodes
  // are the cells a search can ever stand on (roads, cells with flow bits, cells a flow bit points at), numbered in
  // tiled order: a searcher's dist / came_from table has one record per NODE, a quarter of one record per cell.
  unsigned long long* amap;
  int n_nodes;
  // VEHICLE_RESPECT_AWARENESS only: per cell (tiled order) the lengths of the straight runs of road cells that end in it,
  // four 16-bit counts (going -y, +y, -x, +x, the cell included): what the field-of-view rays of astar_numba.py:29-50 ask
  const unsigned long long* fovrun;
  float* density;    // _update_density_map (city_model.py:1764-1778), materialised on demand
  int8_t* occ_snap;  // occupancy at the last tick start (what density_map is a function of)
  // ordered lists
  int32_t* active;    // active_vehicle_agents (vehicle ids, -1 = removed this tick)
  int8_t* sched_kind;
  int32_t* sched_ref;
  int32_t* hslot;     // schedule slot of every host-side agent (rain manager, rain clouds), kept by compaction
  int32_t* bslot;     // schedule slot of every CityBlock, kept by compaction
  int32_t* arr;       // service records, 3 ints each (see AR_*)
  int arr_cap;
  uint32_t* rank;     // per schedule slot
  uint8_t* resolved;  // per schedule slot, this move phase
  // light groups (CSR tables + state)
  int G;
  int32_t *g_light_off, *light_cell, *light_ctrl_off, *light_ctrl, *g_ns_off, *g_ns, *g_ew_off, *g_ew, *g_icell_off,
      *g_icell, *g_nsin_off, *g_nsin, *g_nsout_off, *g_nsout, *g_ewin_off, *g_ewin, *g_ewout_off, *g_ewout, *g_nb,
      *g_nb_ctor, *g_slot;
  // every (cell, group, claim plane) pair a light group announces in the move phase, flattened so that the first
  // round of a move phase claims them with one thread per pair instead of one thread walking a whole group
  int32_t *gc_cell, *gc_group;
  uint8_t* gc_plane;
  int gc_n;
  int32_t *gs_cur, *gs_pend, *gs_trans, *gs_clear, *gs_ftphase, *gs_fttimer, *gs_qtimer, *gs_gap, *gs_last, *gs_nsp,
      *gs_ewp, *gs_repop;
  // per-cell min-rank claims for the move phase live in Cell::claim (epoch-tagged so they never need clearing)
  uint32_t* gclaim_r;
  // decide-phase exchange buffers
  uint8_t *F, *R;
  int32_t* cand;
  // decide-phase RNG bookkeeping on the device (see k_rng_* in engine.hip)
  uint32_t* words;      // ring mirror of the global MT19937 stream (tempered words), index = absolute & WORDS_MASK
  uint32_t *Cx, *rollrank, *rollD, *Tcum;
  DevCnt* cnt;
};

__device__ __forceinline__ int path_dir(const uint32_t* pool, uint32_t off, int k) {
  return (pool[off + ((uint32_t)k >> 4)] >> ((k & 15) * 2)) & 3;
}
__device__ __forceinline__ int step_cell(int cell, int dir, int W) {
  return dir == 0 ? cell + W : dir == 1 ? cell + 1 : dir == 2 ? cell - W : cell - 1;
}
__device__ __forceinline__ void set_occ(const Dev& d, int c, int8_t v) { d.cell[c].occ = v; d.occ[c] = v; }
__device__ __forceinline__ void set_stop(const Dev& d, int c, int8_t v) { d.cell[c].stop = v; d.stop[c] = v; }
// cell -> (x, y) without an integer division (falls back to one when the map is beyond the magic number's range)
__device__ __forceinline__ void cell_xy(const Dev& d, int cell, int& x, int& y) {
  if (d.w_magic) { y = (int)(((unsigned long long)(unsigned)cell * d.w_magic) >> 40); x = cell - y * d.W; }
  else { y = cell / d.W; x = cell - y * d.W; }
}
// position of cell (x, y) in the 8 x 8-tiled order
__device__ __forceinline__ uint32_t tix(const Dev& d, int x, int y) {
  return ((((uint32_t)(y >> 3) * (uint32_t)d.W8 + (uint32_t)(x >> 3)) << 6) | (uint32_t)((y & 7) << 3) | (uint32_t)(x & 7));
}
__device__ __forceinline__ uint32_t claim_rank(uint32_t v, uint32_t prefix) {
  return (v >> RANK_BITS) == prefix ? (v & RANK_MASK) : NO_RANK;
}

// "is ag stranded, as vehicle number my_idx of the decide order sees it" - earlier vehicles have already
// run their step_decide this tick (countdown applied, events visible), later ones have not.
__device__ __forceinline__ bool seen_stranded(const Dev& d, int ag, int my_idx) {
  if (d.seq) return (d.flags[ag] & (VF_COLL | VF_MALF)) != 0;
  const uint8_t e = d.ev[ag];
  if (e) {  // stranded by a malfunction / sideswipe found during this tick's decide phase, at order index j
    const int j = d.ev_idx[ag];
    if (e == 1) return j < my_idx ? true : (d.st_before[ag] != 0);  // the vehicle that drew the event (j = its own index)
    if (j < my_idx) return true;                                      // its partner, seen after the collision
    // ... seen before the collision.  A later partner (3) was a valid candidate when it was hit, i.e. not stranded
    // before its own turn (and its st_before has been re-derived from the post-collision flags since: not usable).
    if (e == 3) return false;
    // An earlier partner (2) was not stranded after its own step_decide, but may have carried an old stranding into
    // it that only expired there: observers ahead of it still saw that (its st_before / st_after predate the event).
  }
  if (d.active_idx[ag] < my_idx) return d.st_after[ag] != 0;
  return d.st_before[ag] != 0;
}

// blocker.is_parked as vehicle number my_idx of the decide order sees it: a remove_on_arrival=False vehicle that sits
// on its target parks inside its own step_decide (vehicle_base.py:657-661 -> on_target_reached -> _park); the flag
// itself is only written after the decide kernels (k_decide_arrive), so earlier deciders are recognised here.
__device__ __forceinline__ bool seen_parked(const Dev& d, int ag, int my_idx) {
  const uint16_t af = d.flags[ag];
  if (af & VF_PARKED) return true;
  if (d.seq) return false;
  if (!(af & VF_KEEP) || my_idx == LAST_IDX || d.active_idx[ag] >= my_idx) return false;
  const int p = d.pos[ag];
  if (p != d.target[ag]) return false;
  const uint8_t e = d.ev[ag];
  if (e == 1 || e == 3 || d.st_after[ag]) return false;   // stranded at its own decide point: it returned early
  return d.stop[p] != 1;
}

__device__ __forceinline__ void svc_record(const Dev& d, int key, int vid, int what) {
  const int k = atomicAdd(&d.cnt->arr_n, 1);
  if (k < d.arr_cap) { d.arr[3 * k] = key; d.arr[3 * k + 1] = vid; d.arr[3 * k + 2] = what; }
}

}  // namespace
