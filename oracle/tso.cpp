/*
 * tso.cpp - CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A single-threaded C++ restatement of the reference's sequential per-tick semantics
 * (SURVEY.md §8(a) A1-A16), exporting the same C-ABI as include/trafficsim.h under the
 * prefix `tso_`.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libtrafficsim_hip.so) never links or calls it.
 *
 * Parity pinning: checked against the golden vectors of tests/golden/ (MT19937 streams from
 * CPython's own `random`; density from the real scipy; A* KATs and per-tick traces from the
 * reference's own source files executed with stand-ins for the absent mesa/numba/tensorflow
 * packages - see tests/golden/standins/README.md).  The Mesa scheduler contract (A4) is not
 * pinned by the reference itself (no lock file): "parity unpinned" at that boundary.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/Simulation).
 */
#include "../include/trafficsim.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// MT19937 exactly as CPython's _randommodule.c (random.random / getrandbits / _randbelow / randint /
// shuffle; Lib/random.py 3.10).  Pinned by tests/golden/mt_kats.npz.
// ---------------------------------------------------------------------------------------------
struct MT {
  uint32_t mt[624];
  uint32_t idx = 625;
  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void init_by_array(const uint32_t* key, size_t len) {
    init_genrand(19650218U);
    size_t i = 1, j = 0;
    size_t k = (624 > len ? 624 : len);
    for (; k; k--) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
      i++; j++;
      if (i >= 624) { mt[0] = mt[623]; i = 1; }
      if (j >= len) j = 0;
    }
    for (k = 623; k; k--) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
      i++;
      if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000U;
  }
  void seed_int(uint64_t s) {  // random.seed(int): key = 32-bit little-endian chunks of abs(s)
    uint32_t key[2] = {(uint32_t)(s & 0xffffffffU), (uint32_t)(s >> 32)};
    init_by_array(key, key[1] ? 2 : 1);
  }
  uint32_t next() {
    if (idx >= 624) {
      static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
      int kk;
      uint32_t y;
      for (kk = 0; kk < 624 - 397; kk++) {
        y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1U];
      }
      for (; kk < 623; kk++) {
        y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1U];
      }
      y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
      mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1U];
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
  }
  double random() {
    uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
  uint32_t getrandbits(int k) { return next() >> (32 - k); }  // 1 <= k <= 32
  uint32_t randbelow(uint32_t n) {                             // Random._randbelow_with_getrandbits
    int k = 32 - __builtin_clz(n);                             // n.bit_length(), n >= 1
    uint32_t r = getrandbits(k);
    while (r >= n) r = getrandbits(k);
    return r;
  }
  int randint(int a, int b) { return a + (int)randbelow((uint32_t)(b - a + 1)); }
};

// crc32 (zlib polynomial) of int32 (x,y) pairs - identical to zlib.crc32(np.int32 array bytes)
uint32_t crc_table[256];
bool crc_init_done = false;
void crc_init() {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320U ^ (c >> 1)) : (c >> 1);
    crc_table[i] = c;
  }
  crc_init_done = true;
}
inline uint32_t crc_update(uint32_t crc, const void* buf, size_t len) {
  const uint8_t* p = (const uint8_t*)buf;
  for (size_t i = 0; i < len; i++) crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return crc;
}

// N, E, S, W deltas (astar_numba.py:9; numba_utilities.py:5-10; config.py:64)
const int DX[4] = {0, 1, 0, -1};
const int DY[4] = {1, 0, -1, 0};
const int INF = 0x3F3F3F3F;

struct Vehicle {
  int spawn_idx;
  int pos;     // cell index, -1 once removed
  int target;  // cell index
  int pop_type;
  std::vector<int> path;  // remaining path = path[head..]
  size_t head = 0;
  int base_speed = 0, current_speed = 0, max_steps = 0;
  bool blocked_by_vehicle = false;
  int direction = -1;
  int cooldown = 0;
  bool is_overtaking = false;
  std::vector<int> overtake_path, pre_overtake_path;
  bool overtake_path_none = false;  // `self.overtake_path = None` (vehicle_base.py:461)
  int overtaking_duration = -1;
  bool is_in_stuck_detour = false;
  std::vector<int> pre_stuck_detour_path, stuck_detour_path;
  bool stuck_detour_path_none = false;
  int stuck_detour_duration = -1;
  int stuck_ticks = 0;
  bool early_exit = false, is_stuck = false, is_parked = false, is_in_collision = false,
       is_in_malfunction = false;
  bool remove_on_arrival = true;
  bool has_prev = false;  // previous_pos == pos
  int stranded_left = 0;
  double depart_time = 0.0;
  int steps_traveled = 0;
  bool alive = true;
  // ServiceVehicleAgent (vehicle_service.py:13-41)
  int svc_type = 0;        // 0 = plain VehicleAgent, TS_TRIP_SERVICE_FOOD / TS_TRIP_SERVICE_WASTE
  int svc_phase = 0;       // 0 "to_block", 1 "servicing", 2 "to_exit"
  int service_ticks = 0;
  int svc_id = -1;         // index into sv_food_ids / sv_waste_ids
  int current_block = -1;
  double max_load = 0.0, current_load = 0.0;
  int next_in_cell = -1;  // MultiGrid cell list (vehicles only, arrival order)
  size_t plen() const { return path.size() - head; }
  bool stranded() const { return is_in_collision || is_in_malfunction; }
};

struct Group {
  std::vector<int> lights;                 // global light indices (traffic_lights order)
  std::vector<int> ns_lights, ew_lights;   // opposite_pairs
  std::vector<int> icells, ns_in, ns_out, ew_in, ew_out;  // cell indices
  int nb_dir[4], nb_grp[4];            // neighbor_groups once re-populated
  int nb_dir_ctor[4], nb_grp_ctor[4];  // neighbor_groups as left by the constructor
  bool links_repopulated = false;      // get_opposite_traffic_lights() re-ran populate_links()
  int current_phase = -1, pending_phase = -1;
  int transition_timer = 0, clearance_timer = 0;
  int ft_phase = 0, fixed_time_timer = 0;
  int queue_timer = 0, gap_timer = 0, last_arrival = 0;
  int ns_pressure = 0, ew_pressure = 0;
};

struct Trip { int origin, dest; double depart; int kind; int day; };  // kind: TS_POP_INTERNAL / TS_POP_THROUGH / TS_TRIP_SERVICE_*; day: pending_by_day's key
struct Block {   // CityBlock (city_block.py:39-77)
  int cells = 0;
  bool needs_food = false, produces_waste = false;
  double max_food = 0, max_waste = 0, food = 0, waste = 0;
  double food_rate = 0, waste_rate = 0, food_rem = 0, waste_rem = 0;
  int ticks_since_food = 0, ticks_since_waste = 0;
  std::vector<int> service_cells;   // get_service_road_cell's ranked candidates
};
struct Generator {
  std::vector<Block> blocks;
  int blocks_scheduled = 0;
  std::vector<char> sv_live;        // service ids currently in the scheduler: [food ids..., waste ids...]
  bool armed = false;
  TsTrafficTables T;
  std::vector<int> blk_type;
  std::vector<std::vector<int>> blk_entr;  // entrance cells per block
  std::vector<int> hw_in, hw_out;
  std::vector<Trip> pending;
  int current_day = 0;
  // _update_cached_stats (dynamic_traffic_generator.py:525-648) and what feeds it
  std::vector<long long> daily_difference_history;     // (165-167)
  long long completed_at_day_start = 0;                // count_completed_internal + _through when the day began: daily_finished_* = now - this
  int ticks_since_stats = 0;
  TsCachedStats cs{};
};

struct Rain {     // RainAgent (rain.py:18-84)
  double x, y, dx, dy;
  int radius;
  std::vector<int> covered;   // covered_cells as of its last step
  bool alive = true;
  int sched_idx = -1;
};

struct SchedEntry {
  int kind;  // TS_AGENT_* or 100 = vehicle
  int ref;
  bool alive;
};

}  // namespace

struct ts_engine {
  int W = 0, H = 0, N = 0;
  TsParams P;
  std::vector<uint8_t> allowed;
  std::vector<int8_t> is_road, road_type, intersection, occ, stop, stuck, rain;
  std::vector<float> density;
  std::vector<int8_t> occ_snap;  // occupancy at the last _update_density_map() call site
  bool density_valid = false;
  std::vector<int> cell_head;  // first vehicle in cell (MultiGrid list minus the CellAgent)
  std::vector<Vehicle> veh;
  std::vector<int> active;     // active_vehicle_agents (vehicle ids; -1 = removed, compacted per tick)
  std::vector<SchedEntry> sched;
  std::vector<int> veh_sched;  // vehicle id -> schedule entry
  std::vector<Group> groups;
  std::vector<int> light_cell;
  std::vector<std::vector<int>> light_ctrl;
  int groups_scheduled = 0;
  MT rng_global, rng_sched;
  bool seeded[2] = {false, false};
  TsCounters C;
  std::unordered_map<uint64_t, std::vector<int>> path_cache;  // city._path_cache
  Generator gen;
  std::vector<Rain> rains_all;     // every RainAgent ever created
  std::vector<int> rains;          // city_model.rains (indices, list order)
  std::vector<int> prev_raining;   // RainManager._prev_raining
  int rain_counter = 0, rain_cooldown_left = 0;
  std::string err;
  int svc_pending_type = 0;   // vehicle_type of the vehicle tso_add_vehicles is placing
  int fatal = 0;   // an exception the reference would have raised inside model.step()
  // A* scratch (epoch-stamped so the O(N) init of astar_numba.py:119-122 is not repeated)
  std::vector<int> a_dist, a_came, a_epoch;
  int epoch = 0;
  std::vector<int> fov_epoch;      // compute_fov_inplace's mask (astar_numba.py:29-50), stamped instead of reset
  int fov_ep = 0;
  std::vector<int> hf, hg, hs, hi;
  std::vector<int8_t> hdir;
};

namespace {

typedef ts_engine E;

int fail(E* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  return code;
}

// --------------------------- MultiGrid cell lists (vehicles only) ----------------------------
void cell_append(E* e, int cell, int vid) {
  e->veh[vid].next_in_cell = -1;
  int h = e->cell_head[cell];
  if (h < 0) { e->cell_head[cell] = vid; return; }
  while (e->veh[h].next_in_cell >= 0) h = e->veh[h].next_in_cell;
  e->veh[h].next_in_cell = vid;
}
void cell_remove(E* e, int cell, int vid) {
  int h = e->cell_head[cell];
  if (h == vid) { e->cell_head[cell] = e->veh[vid].next_in_cell; return; }
  while (h >= 0 && e->veh[h].next_in_cell != vid) h = e->veh[h].next_in_cell;
  if (h >= 0) e->veh[h].next_in_cell = e->veh[vid].next_in_cell;
}

// ------------------------------ density (city_model.py:1764-1778) -----------------------------
// Exact recipe of scipy.ndimage.uniform_filter on float32 input (two uniform_filter1d passes with
// double accumulators and a float32 intermediate), `* 441` in float32, float32 division.
void box_sum_f32(const int8_t* src, int W, int H, int r, std::vector<float>& out) {
  const int size = 2 * r + 1;
  std::vector<float> t((size_t)W * H);
  // axis 0 (y): running window count (exact), /size in double, store float32
  for (int x = 0; x < W; x++) {
    for (int y = 0; y < H; y++) {
      int c = 0;
      int y0 = std::max(0, y - r), y1 = std::min(H - 1, y + r);
      for (int yy = y0; yy <= y1; yy++) c += src[(size_t)yy * W + x];
      t[(size_t)y * W + x] = (float)((double)c / (double)size);
    }
  }
  out.assign((size_t)W * H, 0.f);
  for (int y = 0; y < H; y++) {
    for (int x = 0; x < W; x++) {
      double s = 0.0;
      int x0 = std::max(0, x - r), x1 = std::min(W - 1, x + r);
      for (int xx = x0; xx <= x1; xx++) s += (double)t[(size_t)y * W + xx];
      float v = (float)(s / (double)size);
      out[(size_t)y * W + x] = v * (float)(size * size);
    }
  }
}
void update_density(E* e) {
  const int r = e->P.vehicle_awareness_range;
  std::vector<float> so, sr;
  box_sum_f32(e->occ_snap.data(), e->W, e->H, r, so);
  box_sum_f32(e->is_road.data(), e->W, e->H, r, sr);
  e->density.resize(e->N);
  for (int i = 0; i < e->N; i++) e->density[i] = sr[i] > 0.f ? so[i] / sr[i] : 0.f;
  e->density_valid = true;
}
// faster sliding-window variant used by the per-tick eager path (bitwise identical: window counts
// are exact integers; the axis-1 double sums of <= 21 float32 values are exact in any order)
void update_density_fast(E* e) {
  const int r = e->P.vehicle_awareness_range, W = e->W, H = e->H, size = 2 * r + 1;
  std::vector<float> t0((size_t)W * H), t1((size_t)W * H);
  const int8_t* srcs[2] = {e->occ_snap.data(), e->is_road.data()};
  std::vector<float>* ts[2] = {&t0, &t1};
  for (int m = 0; m < 2; m++) {
    const int8_t* src = srcs[m];
    std::vector<float>& t = *ts[m];
    std::vector<int> col(W, 0);
    for (int yy = 0; yy <= std::min(H - 1, r); yy++)
      for (int x = 0; x < W; x++) col[x] += src[(size_t)yy * W + x];
    for (int y = 0; y < H; y++) {
      for (int x = 0; x < W; x++) t[(size_t)y * W + x] = (float)((double)col[x] / (double)size);
      int add = y + r + 1, sub = y - r;
      if (add < H) for (int x = 0; x < W; x++) col[x] += src[(size_t)add * W + x];
      if (sub >= 0) for (int x = 0; x < W; x++) col[x] -= src[(size_t)sub * W + x];
    }
  }
  e->density.resize(e->N);
  for (int y = 0; y < H; y++) {
    const float* a = &t0[(size_t)y * W];
    const float* b = &t1[(size_t)y * W];
    // running double sums: every partial sum is an exact multiple of 2^-29 below 2^5, so adds and
    // subtracts are exact and the result equals scipy's per-line running sum bit for bit
    double s0 = 0.0, s1 = 0.0;
    for (int xx = 0; xx <= std::min(W - 1, r); xx++) { s0 += (double)a[xx]; s1 += (double)b[xx]; }
    for (int x = 0; x < W; x++) {
      float v0 = (float)(s0 / (double)size) * (float)(size * size);
      float v1 = (float)(s1 / (double)size) * (float)(size * size);
      e->density[(size_t)y * W + x] = v1 > 0.f ? v0 / v1 : 0.f;
      int add = x + r + 1, sub = x - r;
      if (add < W) { s0 += (double)a[add]; s1 += (double)b[add]; }
      if (sub >= 0) { s0 -= (double)a[sub]; s1 -= (double)b[sub]; }
    }
  }
  e->density_valid = true;
}

// ------------------------------ A* (astar_numba.py:87-239) ------------------------------------
// Verbatim semantics incl. both quirks of SURVEY §8(a) A13: (1) dir_arr is indexed by heap SLOT and
// is not swapped by the sift routines; (2) `ng` is a float (R1 penalty 0.5) and truncates on store.
// respect_awareness (FOV, astar_numba.py:29-50; off by default, config.py:278) is restated inside astar().
#ifdef TSO_STATS
// Profiling aid (profiles/heap_stats.py; built into a separate library, never into libtso.so): what the heaps of the
// searches look like - heap size at every pop, levels the hole sinks / a push rises, searches by expansions.
struct AStats { long long pop_size[64], sink[40], rise[40], relax_n[5], search_exp[40], pops, stale;
                long long ext_n[32], ext_exp[32], empty_n[4], empty_exp[4], kind_n[4], kind_exp[4], maxheap_n[32], maxheap_exp[32], fspread[32]; };
static AStats g_astats;
static inline int bitlen(long long v) { int b = 0; while (v > 0) { b++; v >>= 1; } return b; }
#define STAT(x) do { x; } while (0)
#else
#define STAT(x) do { } while (0)
#endif
inline void heap_sift_up(E* e, int i) {
  int rise_ = 0; (void)rise_;
  while (i > 0) {
    int parent = (i - 1) / 2;
    if (e->hf[i] < e->hf[parent]) {
      std::swap(e->hf[i], e->hf[parent]);
      std::swap(e->hg[i], e->hg[parent]);
      std::swap(e->hs[i], e->hs[parent]);
      std::swap(e->hi[i], e->hi[parent]);
      i = parent;
      STAT(rise_++);
    } else break;
  }
  STAT(g_astats.rise[rise_ < 39 ? rise_ : 39]++);
}
inline void heap_sift_down(E* e, int size) {
  int idx = 0, sink_ = 0; (void)sink_;
  for (;;) {
    int left = 2 * idx + 1, right = left + 1, smallest = idx;
    if (left < size && e->hf[left] < e->hf[smallest]) smallest = left;
    if (right < size && e->hf[right] < e->hf[smallest]) smallest = right;
    if (smallest != idx) {
      std::swap(e->hf[idx], e->hf[smallest]);
      std::swap(e->hg[idx], e->hg[smallest]);
      std::swap(e->hs[idx], e->hs[smallest]);
      std::swap(e->hi[idx], e->hi[smallest]);
      idx = smallest;
      STAT(sink_++);
    } else break;
  }
  STAT(g_astats.sink[sink_ < 39 ? sink_ : 39]++);
}
inline void heap_reserve(E* e, int n) {
  if ((int)e->hf.size() < n) {
    size_t m = std::max<size_t>(n, e->hf.size() * 2 + 64);
    e->hf.resize(m); e->hg.resize(m); e->hs.resize(m); e->hi.resize(m); e->hdir.resize(m);
  }
}

void astar(E* e, int sx, int sy, int gx, int gy, bool soft, bool ignore_flow, int maximum_steps,
           std::vector<int>& out) {
  out.clear();
  e->C.astar_calls++;
  const int W = e->W, H = e->H;
  const TsParams& P = e->P;
  if ((int)e->a_dist.size() != e->N) {
    e->a_dist.assign(e->N, INF); e->a_came.assign(e->N, -1); e->a_epoch.assign(e->N, 0); e->epoch = 0;
  }
  if (soft && !e->density_valid) update_density_fast(e);
  const int ep = ++e->epoch;
  auto dist_get = [&](int i) { return e->a_epoch[i] == ep ? e->a_dist[i] : INF; };
  auto dist_set = [&](int i, int d, int from) { e->a_epoch[i] = ep; e->a_dist[i] = d; e->a_came[i] = from; };
  const int start_idx = sy * W + sx, goal_idx = gy * W + gx;
  // compute_fov_inplace(start_x, start_y, ...) (astar_numba.py:29-50, 113-115): from a line of 2 * awareness - 1 cells through
  // the start, perpendicular to each of the four directions, rays run along the direction for as long as they stay on road
  const bool fov_on = P.respect_awareness != 0;
  if (fov_on) {
    if ((int)e->fov_epoch.size() != e->N) { e->fov_epoch.assign(e->N, 0); e->fov_ep = 0; }
    const int fe = ++e->fov_ep, aw = P.vehicle_awareness_range;
    for (int d = 0; d < 4; d++) {
      const int dx = DX[d], dy = DY[d], px = -dy, py = dx;
      for (int offset = -aw + 1; offset < aw; offset++) {
        const int x0 = sx + offset * px, y0 = sy + offset * py;
        int step = 0, x = x0, y = y0;
        while (x >= 0 && x < W && y >= 0 && y < H && e->is_road[y * W + x] == 1) {
          e->fov_epoch[y * W + x] = fe;
          step++;
          x = x0 + dx * step; y = y0 + dy * step;
        }
      }
    }
  }
  auto seen = [&](int i) { return !fov_on || e->fov_epoch[i] == e->fov_ep; };
  dist_set(start_idx, 0, -1);
  heap_reserve(e, 8);
  int heap_size = 1;
  e->hf[0] = std::abs(sx - gx) + std::abs(sy - gy);
  e->hg[0] = 0; e->hs[0] = 0; e->hi[0] = start_idx; e->hdir[0] = -1;
  long long exp0_ = e->C.astar_expansions; (void)exp0_;
  int ext_ = 0, maxheap_ = 0; (void)ext_; (void)maxheap_;
  struct StatEnd { int64_t& c; long long c0; int& ext; int& mh; std::vector<int>& out; int kind;
    ~StatEnd() { STAT(const long long n = c - c0; g_astats.search_exp[bitlen(n)]++; const int b = ext / 32 < 31 ? ext / 32 : 31;
                      g_astats.ext_n[b]++; g_astats.ext_exp[b] += n; g_astats.kind_n[kind]++; g_astats.kind_exp[kind] += n;
                      const int hb = mh / 64 < 31 ? mh / 64 : 31; g_astats.maxheap_n[hb]++; g_astats.maxheap_exp[hb] += n;
                      if (out.empty()) { g_astats.empty_n[kind]++; g_astats.empty_exp[kind] += n; }); } }
    stat_end_{e->C.astar_expansions, exp0_, ext_, maxheap_, out, (soft ? 1 : 0) + (ignore_flow ? 2 : 0)};
  while (heap_size > 0) {
    int g = e->hg[0], steps = e->hs[0], cur = e->hi[0];
    int prev_dir = e->hdir[0];
    STAT(g_astats.pops++; g_astats.pop_size[heap_size < 64 * 64 ? heap_size / 64 : 63]++);
    heap_size--;
    if (heap_size > 0) {
      e->hf[0] = e->hf[heap_size]; e->hg[0] = e->hg[heap_size]; e->hs[0] = e->hs[heap_size];
      e->hi[0] = e->hi[heap_size]; e->hdir[0] = e->hdir[heap_size];
      heap_sift_down(e, heap_size);
    }
    if (cur == goal_idx) {
      int idx = cur;
      while (idx != start_idx) { out.push_back(idx); idx = e->a_came[idx]; }
      std::reverse(out.begin(), out.end());
      return;
    }
    if (g > dist_get(cur)) { STAT(g_astats.stale++); continue; }
    e->C.astar_expansions++;
    const int cx = cur % W, cy = cur / W;
    int nrel_ = 0; (void)nrel_;
    struct StatRel { int& n; ~StatRel() { STAT(g_astats.relax_n[n]++); } } stat_rel_{nrel_};
    for (int d = 0; d < 4; d++) {
      int nx = cx + DX[d], ny = cy + DY[d];
      if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
      int ns = steps + 1;
      if (ns > maximum_steps) continue;
      int nidx = ny * W + nx;
      double ng = g + 1;
      if (P.turn_penalty_enabled && prev_dir != -1 && d != prev_dir) ng += P.turn_penalty;
      uint8_t bits = e->allowed[cur];
      if ((bits & (1 << d)) == 0) {
        if (ignore_flow && e->is_road[nidx] == 1) ng += P.contraflow_penalty;
        else continue;
      }
      if (e->occ[nidx] == 1 && seen(nidx)) {
        if (soft && P.dynamic_penalties_enabled) {
          double p = P.obstacle_penalty_vehicle;
          double local_density = (double)e->density[nidx];
          ng += (double)(long long)(p * (1.0 + P.dynamic_penalty_scale * local_density));
        } else if (soft) ng += P.obstacle_penalty_vehicle;
        else continue;
      }
      if (e->stop[nidx] == 1 && seen(nidx)) {
        if (soft) ng += P.obstacle_penalty_stop;
        else continue;
      }
      if (P.road_type_penalties_enabled && e->is_road[nidx] == 1) {
        int rt = e->road_type[nidx];
        if (rt == 1) ng += P.road_type_penalty_r1;
        else if (rt == 2) ng += P.road_type_penalty_r2;
        else if (rt == 3) ng += P.road_type_penalty_r3;
      }
      if (ng < (double)dist_get(nidx)) {
        e->C.astar_relaxations++;
        STAT(nrel_++; ext_ = std::max(ext_, std::max(std::abs(nx - sx), std::abs(ny - sy))); maxheap_ = std::max(maxheap_, heap_size + 1);
             { int fs = (int)(ng + (std::abs(nx - gx) + std::abs(ny - gy))) - e->hf[0]; if (heap_size > 0) g_astats.fspread[bitlen(fs < 0 ? 0 : fs)]++; });
        dist_set(nidx, (int)ng, cur);
        int h = std::abs(nx - gx) + std::abs(ny - gy);
        int i = heap_size;
        heap_reserve(e, i + 1);
        e->hf[i] = (int)(ng + h); e->hg[i] = (int)ng; e->hs[i] = ns; e->hi[i] = nidx; e->hdir[i] = (int8_t)d;
        heap_sift_up(e, i);
        heap_size++;
      }
    }
  }
}

// ------------------------------ vehicle helpers -----------------------------------------------
inline int first_vehicle_on_cell(E* e, int cell) { return e->cell_head[cell]; }  // _vehicle_on_cell (711-717)

void set_collision(E* e, Vehicle& v, int ticks) {  // vehicle_base.py:534-541
  v.is_in_collision = true; v.is_in_malfunction = false; v.stranded_left = ticks;
  v.base_speed = 0; v.current_speed = 0; e->C.collisions++;
}
void set_malfunction(E* e, Vehicle& v, int ticks) {  // vehicle_base.py:543-550
  v.is_in_malfunction = true; v.is_in_collision = false; v.stranded_left = ticks;
  v.base_speed = 0; v.current_speed = 0; e->C.malfunctions++;
}
bool tick_stranded(E* e, Vehicle& v) {  // vehicle_base.py:552-565
  if (!v.stranded()) return false;
  v.stranded_left -= 1;
  if (v.stranded_left <= 0) {
    if (v.is_in_collision) e->C.collisions--;
    if (v.is_in_malfunction) e->C.malfunctions--;
    v.is_in_collision = false; v.is_in_malfunction = false; v.stranded_left = 0;
  }
  return v.stranded();
}

void remove_vehicle(E* e, int vid, int pop_arg = -1);

void start_service(E* e, int vid);

void on_target_reached(E* e, int vid) {  // vehicle_base.py:755-775; vehicle_service.py:54-60
  Vehicle& v = e->veh[vid];
  if (v.svc_type && v.svc_phase == 0) { start_service(e, vid); return; }
  if (e->P.enable_traffic) {
    double duration = e->C.elapsed - v.depart_time;
    if (v.pop_type == TS_POP_INTERNAL) {
      e->C.total_duration_internal += duration; e->C.total_distance_internal += v.steps_traveled;
      e->C.count_completed_internal++;
    } else if (v.pop_type == TS_POP_THROUGH) {
      e->C.total_duration_through += duration; e->C.total_distance_through += v.steps_traveled;
      e->C.count_completed_through++;
    }
  }
  if (v.remove_on_arrival) remove_vehicle(e, vid);
  else if (!v.is_parked) { v.is_parked = true; e->C.parked++; }
}

void remove_vehicle(E* e, int vid, int pop_arg) {  // city_model.py:1920-1941 (pop_arg: the caller's population_type argument; -1 = the vehicle's own, as _despawn passes it)
  Vehicle& v = e->veh[vid];
  e->occ[v.pos] = 0; e->stuck[v.pos] = 0;
  for (size_t i = 0; i < e->active.size(); i++) if (e->active[i] == vid) { e->active[i] = -1; break; }
  cell_remove(e, v.pos, vid);
  e->sched[e->veh_sched[vid]].alive = false;
  const int pop = pop_arg >= 0 ? pop_arg : v.pop_type;
  if (pop == TS_POP_INTERNAL) e->C.live_internal--;
  else if (pop == TS_POP_THROUGH) {
    e->C.live_through--;
    if (v.svc_type == TS_TRIP_SERVICE_FOOD) e->C.live_service_food--;
    else if (v.svc_type == TS_TRIP_SERVICE_WASTE) e->C.live_service_waste--;
  }
  if (v.svc_type && v.svc_id >= 0) e->gen.sv_live[(v.svc_type == TS_TRIP_SERVICE_FOOD ? 0 : e->gen.T.total_service_vehicles_food) + v.svc_id] = 0;
  v.alive = false; v.pos = -1;
}

// _scan_ahead_for_obstacles (vehicle_base.py:422-452) incl. its index-0-only break
void scan_ahead(E* e, const Vehicle& v, int& idx_stop, int& idx_vehicle) {
  idx_stop = -1; idx_vehicle = -1;
  size_t n = v.plen();
  if (n == 0) return;
  int look = (int)std::min<size_t>((size_t)e->P.vehicle_awareness_range, n);
  for (int i = 0; i < look; i++) {
    int c = v.path[v.head + i];
    if (idx_stop < 0 && e->stop[c] == 1) idx_stop = i;
    if (idx_vehicle < 0 && e->occ[c] == 1) idx_vehicle = i;
    if (idx_stop == 0 || idx_vehicle == 0) break;
  }
}

inline bool contains(const std::vector<int>& p, int cell) {
  return std::find(p.begin(), p.end(), cell) != p.end();
}

// _compute_path_internal (vehicle_base.py:199-420)
void compute_path_internal(E* e, int vid, std::vector<int>& out) {
  Vehicle& v = e->veh[vid];
  const TsParams& P = e->P;
  const int W = e->W;
  const int sx = v.pos % W, sy = v.pos / W, gx = v.target % W, gy = v.target / W;
  std::vector<int> bypass;
  // Phase 0a: re-merge after overtaking (219-247)
  if (v.is_overtaking && !v.pre_overtake_path.empty()) {
    int merge_idx = -1;
    for (size_t i = 0; i < v.pre_overtake_path.size(); i++)
      if (e->occ[v.pre_overtake_path[i]] == 0) { merge_idx = (int)i; break; }
    if (merge_idx >= 0) {
      int b = v.pre_overtake_path[merge_idx];
      astar(e, sx, sy, b % W, b / W, false, true, P.max_contraflow_overtake_steps, bypass);
      if (!bypass.empty() && bypass.back() == b) {
        v.overtake_path = bypass; v.overtake_path_none = false;
        out = bypass;
        out.insert(out.end(), v.pre_overtake_path.begin() + merge_idx + 1, v.pre_overtake_path.end());
        return;
      }
    }
  }
  // Phase 0b: re-merge after a stuck detour (249-277)
  if (v.is_in_stuck_detour && !v.pre_stuck_detour_path.empty()) {
    int merge_idx = -1;
    for (size_t i = 0; i < v.pre_stuck_detour_path.size(); i++)
      if (e->occ[v.pre_stuck_detour_path[i]] == 0) { merge_idx = (int)i; break; }
    if (merge_idx >= 0) {
      int b = v.pre_stuck_detour_path[merge_idx];
      astar(e, sx, sy, b % W, b / W, false, true, P.max_contraflow_overtake_steps, bypass);
      if (!bypass.empty() && bypass.back() == b) {
        v.stuck_detour_path = bypass; v.stuck_detour_path_none = false;
        out = bypass;
        out.insert(out.end(), v.pre_stuck_detour_path.begin() + merge_idx + 1, v.pre_stuck_detour_path.end());
        return;
      }
    }
  }
  // Phase 1: strict (280-291); Phase 2: soft obstacles (294-306)
  std::vector<int> path;
  astar(e, sx, sy, gx, gy, false, false, 0x7FFFFFFF, path);
  if (path.empty()) astar(e, sx, sy, gx, gy, true, false, 0x7FFFFFFF, path);
  // Phase 3: contraflow overtake of a stranded/parked blocker (309-366)
  if (P.contraflow_overtake_active && !path.empty()) {
    int aw = P.vehicle_awareness_range;
    int idx_stop = -1, idx_vehicle = -1;
    int look = (int)std::min<size_t>((size_t)aw, path.size());
    for (int i = 0; i < look; i++) {
      if (idx_stop < 0 && e->stop[path[i]] == 1) idx_stop = i;
      if (idx_vehicle < 0 && e->occ[path[i]] == 1) idx_vehicle = i;
      if (idx_stop >= 0 && idx_vehicle >= 0) break;
    }
    if (idx_vehicle == 0) {
      int b = first_vehicle_on_cell(e, path[0]);
      if (b >= 0 && (e->veh[b].stranded() || e->veh[b].is_parked)) {
        int bt = -1;
        for (int c : path) if (e->occ[c] == 0) { bt = c; break; }
        if (bt >= 0) {
          astar(e, sx, sy, bt % W, bt / W, false, true, P.max_contraflow_overtake_steps, bypass);
          if (!bypass.empty() && bypass.back() == bt && bypass.size() > 1) {
            int idx_bp = -1;
            for (size_t i = 0; i < path.size(); i++) if (path[i] == bt) { idx_bp = (int)i; break; }
            if (idx_bp >= 0) {
              v.pre_overtake_path = path;
              v.overtake_path = bypass; v.overtake_path_none = false;
              out = bypass;
              out.insert(out.end(), path.begin() + idx_bp + 1, path.end());
              v.is_overtaking = true;
              e->C.overtaking++;
              v.overtaking_duration = 0;
              return;
            }
          }
        }
      }
    }
  }
  // Phase 4: stuck detour (369-418)
  if (P.stuck_contraflow_enabled && !path.empty()) {
    int threshold = e->intersection[v.pos] == 1 ? P.stuck_contraflow_threshold_intersection
                                                : P.stuck_contraflow_threshold;
    if (v.stuck_ticks >= threshold) {
      int bt = -1;
      for (int c : path) if (e->occ[c] == 0) { bt = c; break; }
      if (bt >= 0) {
        astar(e, sx, sy, bt % W, bt / W, true, true, P.max_contraflow_stuck_detour_steps, bypass);
        if (!bypass.empty() && bypass.back() == bt && bypass.size() > 1) {
          int merge_idx = -1;
          for (size_t i = 0; i < path.size(); i++) if (path[i] == bt) { merge_idx = (int)i; break; }
          if (merge_idx >= 0) {
            v.pre_stuck_detour_path = path;
            v.stuck_detour_path = bypass; v.stuck_detour_path_none = false;
            out = bypass;
            out.insert(out.end(), path.begin() + merge_idx + 1, path.end());
            e->C.in_stuck_detour++;
            v.is_in_stuck_detour = true;
            v.stuck_detour_duration = 0;
            return;
          }
        }
      }
      out = path;
      return;
    }
  }
  out = path;
}

// _compute_path (vehicle_base.py:143-167)
void compute_path(E* e, int vid, bool use_cache, std::vector<int>& out) {
  Vehicle& v = e->veh[vid];
  v.cooldown = e->P.pathfinding_cooldown;
  uint64_t key = ((uint64_t)(uint32_t)v.pos << 32) | (uint32_t)v.target;
  if (use_cache && e->P.pathfinding_cache) {
    auto it = e->path_cache.find(key);
    if (it != e->path_cache.end()) { out = it->second; return; }
  }
  compute_path_internal(e, vid, out);
  if (use_cache && e->P.pathfinding_cache && !out.empty() && !v.is_overtaking && !v.is_in_stuck_detour)
    e->path_cache[key] = out;
}

inline void set_path(Vehicle& v, std::vector<int>& p) { v.path.swap(p); v.head = 0; }

// _check_sideswipe_collision (vehicle_base.py:567-605)
void check_sideswipe(E* e, int vid) {
  Vehicle& v = e->veh[vid];
  if (!e->P.sideswipe_active || v.direction < 0) return;
  const int W = e->W, H = e->H;
  const int left_dir = (v.direction + 3) & 3, right_dir = (v.direction + 1) & 3;
  const int opposite = (v.direction + 2) & 3;
  const int lat[2] = {left_dir, right_dir};
  int x = v.pos % W, y = v.pos / W;
  for (int k = 0; k < 2; k++) {
    int nx = x + DX[lat[k]], ny = y + DY[lat[k]];
    if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
    for (int a = e->cell_head[ny * W + nx]; a >= 0; a = e->veh[a].next_in_cell) {
      Vehicle& ag = e->veh[a];
      if (ag.current_speed <= 0 || ag.is_stuck || ag.is_parked || ag.is_in_collision || ag.is_in_malfunction)
        continue;
      if (ag.direction != opposite) continue;
      if (e->rng_global.random() >= e->P.sideswipe_chance) return;
      set_collision(e, v, e->P.sideswipe_duration);
      set_collision(e, ag, e->P.sideswipe_duration);
      return;
    }
  }
}

// _recompute_path_on_obstacle (vehicle_base.py:454-504)
void recompute_on_obstacle(E* e, int vid, int& idx_stop, int& idx_vehicle) {
  Vehicle& v = e->veh[vid];
  const TsParams& P = e->P;
  if (v.is_overtaking && (v.overtake_path_none || v.overtake_path.empty() || !contains(v.overtake_path, v.pos))) {
    v.overtake_path.clear(); v.overtake_path_none = true; v.is_overtaking = false;
  }
  if (v.is_in_stuck_detour &&
      (v.stuck_detour_path_none || v.stuck_detour_path.empty() || !contains(v.stuck_detour_path, v.pos))) {
    v.stuck_detour_path.clear(); v.stuck_detour_path_none = true; v.is_in_stuck_detour = false;
  }
  scan_ahead(e, v, idx_stop, idx_vehicle);
  if (v.is_overtaking) {
    v.overtaking_duration += 1;
    if (v.overtaking_duration <= P.contraflow_overtake_duration) return;
  }
  if (v.is_in_stuck_detour) {
    v.stuck_detour_duration += 1;
    if (v.stuck_detour_duration <= P.contraflow_stuck_detour_duration) return;
  }
  if (v.cooldown > 0) {
    if (idx_vehicle == 0) {
      int b = first_vehicle_on_cell(e, v.path[v.head]);
      if (b >= 0 && (e->veh[b].stranded() || e->veh[b].is_parked)) {
        // immediate pathfinding
      } else { v.cooldown -= 1; return; }
    } else { v.cooldown -= 1; return; }
  }
  if (idx_stop >= 0 || idx_vehicle >= 0) {
    std::vector<int> p;
    compute_path(e, vid, false, p);
    if (!p.empty()) {
      set_path(v, p);
      scan_ahead(e, v, idx_stop, idx_vehicle);
    }
  }
}

// step_decide (vehicle_base.py:616-663); returns true if the vehicle removed itself
bool step_decide(E* e, int vid) {
  Vehicle& v = e->veh[vid];
  const TsParams& P = e->P;
  v.early_exit = false;
  if (tick_stranded(e, v)) { v.base_speed = 0; v.current_speed = 0; v.early_exit = true; return false; }
  // _check_malfunction (608-610): `not ACTIVE or random() < CHANCE` (short-circuit: no draw when inactive)
  if (!P.malfunction_active || e->rng_global.random() < P.malfunction_chance)
    set_malfunction(e, v, P.malfunction_duration);
  if (v.stranded()) { v.base_speed = 0; v.current_speed = 0; v.early_exit = true; return false; }
  check_sideswipe(e, vid);
  if (v.stranded()) { v.base_speed = 0; v.current_speed = 0; v.early_exit = true; return false; }
  if (e->stop[v.pos] == 1) { v.base_speed = 0; v.current_speed = 0; v.early_exit = true; return false; }
  // _compute_speed (94-112)
  if (v.base_speed == 0) v.base_speed = e->rng_global.randint(P.vehicle_min_speed, P.vehicle_max_speed);
  int speed = v.base_speed;
  if (P.rain_enabled && e->rain[v.pos] == 1) speed = std::max(1, speed - P.rain_speed_reduction);
  v.current_speed = speed;
  // _recompute_path_on_stuck (506-517)
  {
    int thresh = e->intersection[v.pos] == 1 ? P.stuck_recompute_threshold_intersection : P.stuck_recompute_threshold;
    if (v.stuck_ticks >= thresh) {
      std::vector<int> p;
      compute_path(e, vid, false, p);
      set_path(v, p);
    }
  }
  int idx_stop, idx_vehicle;
  recompute_on_obstacle(e, vid, idx_stop, idx_vehicle);
  // _determine_max_steps (719-731)
  int max_steps = std::min<int>(v.current_speed, (int)v.plen());
  bool blocked = false;
  if (idx_stop >= 0) max_steps = std::min(max_steps, idx_stop);
  if (idx_vehicle >= 0) {
    if (idx_vehicle == 0) blocked = true;
    max_steps = std::min(max_steps, idx_vehicle);
  }
  v.max_steps = max_steps; v.blocked_by_vehicle = blocked;
  if (max_steps <= 0) {
    v.base_speed = 0;
    bool removed = false;
    if (v.pos == v.target) { on_target_reached(e, vid); removed = !e->veh[vid].alive; }
    e->veh[vid].early_exit = true;
    return removed;
  }
  v.early_exit = false;
  return false;
}

// _move_to (521-532) + CityModel.move_vehicle (city_model.py:1945-1963)
void move_to(E* e, int vid, int new_pos) {
  Vehicle& v = e->veh[vid];
  int old_pos = v.pos;
  e->occ[old_pos] = 0;
  cell_remove(e, old_pos, vid);
  cell_append(e, new_pos, vid);
  e->occ[new_pos] = 1;
  e->stuck[old_pos] = 0;
  e->stuck[new_pos] = v.is_stuck ? 1 : 0;
  v.pos = new_pos;
  int dx = new_pos % e->W - old_pos % e->W, dy = new_pos / e->W - old_pos / e->W;
  int dir = -1;  // compute_direction (numba_utilities.py:14-28)
  if (dx == 0 && dy == 1) dir = 0; else if (dx == 1 && dy == 0) dir = 1;
  else if (dx == 0 && dy == -1) dir = 2; else if (dx == -1 && dy == 0) dir = 3;
  if (dir != -1) v.direction = dir;
  if (v.stuck_ticks > 0) {
    if (v.is_stuck) { e->C.stuck--; v.is_stuck = false; }
    v.stuck_ticks = 0;
  }
}

// VehicleAgent.step (vehicle_base.py:666-685): with PATHFINDING_BATCHING=False step_decide runs here (669-670), on the
// maps and the global stream as the agents stepped before this one left them
void finish_service(E* e, int vid);
void vehicle_step(E* e, int vid) {
  Vehicle& v = e->veh[vid];
  e->C.agent_steps++;
  if (v.svc_type && v.svc_phase == 1) {  // ServiceVehicleAgent.step (vehicle_service.py:43-52)
    v.service_ticks -= 1;
    if (v.service_ticks <= 0) finish_service(e, vid);
    return;
  }
  if (!e->P.pathfinding_batching) {
    // a vehicle that despawns inside its own step_decide has pos None for the rest of step(): nothing below applies to it
    if (step_decide(e, vid)) return;
  }
  if (!v.early_exit) {
    // _execute_movement (733-753)
    for (int step_idx = 0; step_idx < v.max_steps; step_idx++) {
      if (v.plen() == 0) break;
      int c = v.path[v.head];
      if (e->occ[c] == 1 && c != v.pos) break;
      if (e->stop[c] == 1 && step_idx != v.max_steps - 1) break;  // tuple `is not` => always true (746)
      move_to(e, vid, c);
      v.steps_traveled++;
      v.head++;
    }
    v.has_prev = true;  // previous_pos = pos (677)
  } else {
    v.early_exit = false;
    // tick_stuck (687-693)
    if (v.has_prev && e->stop[v.pos] != 1) {
      v.stuck_ticks++;
      if (v.stuck_ticks > e->P.stuck_recompute_threshold && !v.is_stuck) { e->C.stuck++; v.is_stuck = true; }
    }
  }
  if (v.pos == v.target) on_target_reached(e, vid);
  // _despawn_check (695-706).  A vehicle on_target_reached has just removed has pos None: not on an intersection, and
  // removing it a second time raises in the reference - it cannot have both arrived by moving and be stuck that long.
  if (e->P.stuck_despawn_enabled && v.alive) {
    const int thr = e->intersection[v.pos] == 1 ? e->P.stuck_despawn_threshold_intersection : e->P.stuck_despawn_threshold;
    if (v.stuck_ticks >= thr) {
      if (v.is_stuck) { e->C.stuck--; v.is_stuck = false; }
      if (v.pop_type == TS_POP_INTERNAL) e->C.errored_internal++; else e->C.errored_through++;
      remove_vehicle(e, vid);
    }
  }
}


// ------------------------------ city blocks + service vehicles -------------------------------
// CityBlock.step (city_block.py:110-150)
void block_step(E* e, int bi) {
  if (bi >= (int)e->gen.blocks.size()) return;   // no block tables: nothing observable
  Block& b = e->gen.blocks[bi];
  const TsTrafficTables& T = e->gen.T;
  if (b.needs_food) {
    if (T.gradual_city_block_resources) {
      b.food_rem += b.food_rate;
      if (b.food_rem >= 1.0) {
        double whole = std::trunc(b.food_rem);   // int(remainder)
        b.food = std::max(b.food - whole, 0.0);
        b.food_rem -= whole;
      }
    } else if (++b.ticks_since_food >= T.food_consumption_ticks) {
      b.food = std::max(b.food - (double)b.cells, 0.0);
      b.ticks_since_food = 0;
    }
  }
  if (b.produces_waste) {
    if (T.gradual_city_block_resources) {
      b.waste_rem += b.waste_rate;
      if (b.waste_rem >= 1.0) {
        double whole = std::trunc(b.waste_rem);
        b.waste = std::min(b.waste + whole, b.max_waste);
        b.waste_rem -= whole;
      }
    } else if (++b.ticks_since_waste >= T.waste_production_ticks) {
      b.waste = std::min(b.waste + (double)b.cells, b.max_waste);
      b.ticks_since_waste = 0;
    }
  }
}

// CityBlock.get_service_road_cell step 4 (city_block.py:192-202): first ranked cell without a parked vehicle
int service_road_cell(E* e, int bi) {
  for (int c : e->gen.blocks[bi].service_cells) {
    bool parked = false;
    for (int a = e->cell_head[c]; a >= 0; a = e->veh[a].next_in_cell) if (e->veh[a].is_parked) { parked = true; break; }
    if (!parked) return c;
  }
  return -1;
}

// ServiceVehicleAgent._start_service (vehicle_service.py:85-104)
void start_service(E* e, int vid) {
  Vehicle& v = e->veh[vid];
  if (!v.is_parked) { v.is_parked = true; e->C.parked++; }
  Block& b = e->gen.blocks[v.current_block];
  if (v.svc_type == TS_TRIP_SERVICE_FOOD) {
    double need = b.max_food - b.food;
    double amt = std::min(v.current_load, need);
    b.food = std::min(b.food + amt, b.max_food);
    v.current_load -= amt;
  } else {
    double surplus = b.waste;
    double cap = v.max_load - v.current_load;
    double amt = std::min(cap, surplus);
    b.waste = std::max(b.waste - amt, 0.0);
    v.current_load += amt;
  }
  v.service_ticks = e->gen.T.service_load_time;
  v.svc_phase = 1;
}

// ServiceVehicleAgent._finish_service (vehicle_service.py:106-141)
void finish_service(E* e, int vid) {
  Vehicle& v = e->veh[vid];
  Generator& G = e->gen;
  if (v.is_parked) { v.is_parked = false; e->C.parked--; }
  bool more = v.svc_type == TS_TRIP_SERVICE_FOOD ? v.current_load > 0 : v.current_load < v.max_load;
  if (more) {
    // get_block_most_in_need_of_food / _waste_pickup (city_model.py:2078-2087): stable sorts, first element
    int next_blk = -1;
    for (size_t b = 0; b < G.blocks.size(); b++) {
      const Block& B = G.blocks[b];
      if (v.svc_type == TS_TRIP_SERVICE_FOOD) {
        if (B.needs_food && (next_blk < 0 || B.food < G.blocks[next_blk].food)) next_blk = (int)b;
      } else {
        if (B.produces_waste && (next_blk < 0 || B.waste > G.blocks[next_blk].waste)) next_blk = (int)b;
      }
    }
    if (next_blk >= 0) {
      v.current_block = next_blk;
      int cell = service_road_cell(e, next_blk);
      if (cell < 0) {  // self.target = None -> _compute_path raises AttributeError
        e->fatal = TS_E_UNSUPPORTED;
        e->err = "service vehicle: every service road cell of the next block holds a parked vehicle (the reference raises)";
        return;
      }
      v.target = cell;
      std::vector<int> p;
      compute_path(e, vid, true, p);
      set_path(v, p);
      v.svc_phase = 0;
      return;
    }
  }
  const int W = e->W;
  int best = -1, best_d = 0;
  for (int c : G.hw_out) {   // min(exits, key=manhattan): first minimum
    int d = std::abs(c % W - v.pos % W) + std::abs(c / W - v.pos / W);
    if (best < 0 || d < best_d) { best = c; best_d = d; }
  }
  v.target = best;
  std::vector<int> p;
  compute_path(e, vid, true, p);
  set_path(v, p);
  v.remove_on_arrival = true;
  v.svc_phase = 2;
}

// ------------------------------ light groups ---------------------------------------------------
void light_set(E* e, int light, int8_t val) {  // CellAgent.set_light_stop/go (cell.py:241-251)
  e->stop[e->light_cell[light]] = val;
  for (int c : e->light_ctrl[light]) e->stop[c] = val;
}
inline int queue_sum(E* e, const std::vector<int>& cells) {  // compute_approach_queue (numba_utilities.py:65-72)
  int q = 0;
  for (int c : cells) q += e->occ[c];
  return q;
}
void apply_phase(Group& g, int phase) {  // intersection_light_group.py:386-393
  if (phase == g.current_phase || phase == g.pending_phase) return;
  g.pending_phase = phase;
}
void group_step(E* e, int gi) {  // IntersectionLightGroup.step (396-423)
  Group& g = e->groups[gi];
  const TsParams& P = e->P;
  if (g.pending_phase < 0) {
    switch (P.light_algorithm) {
      case TS_LIGHTS_FIXED_TIME:  // run_fixed_time (427-441)
        g.fixed_time_timer += 1;
        if (g.fixed_time_timer == 1) apply_phase(g, g.ft_phase);
        if (g.fixed_time_timer >= P.green_duration) { g.ft_phase = 1 - g.ft_phase; g.fixed_time_timer = 0; }
        break;
      case TS_LIGHTS_QUEUE_ACTUATED: {  // run_queue_actuated (463-494)
        g.queue_timer += 1;
        int ns_q = queue_sum(e, g.ns_in), ew_q = queue_sum(e, g.ew_in);
        int current_q, opp_q;
        if (g.current_phase == 0) { current_q = ns_q; opp_q = ew_q; } else { current_q = ew_q; opp_q = ns_q; }
        if (g.queue_timer == 1) { g.last_arrival = current_q; g.gap_timer = 0; }
        if (current_q > g.last_arrival) { g.last_arrival = current_q; g.gap_timer = 0; }
        else g.gap_timer += 1;
        if (g.queue_timer >= P.qa_min_green &&
            (g.gap_timer >= P.qa_gap || g.queue_timer >= P.qa_max_green || (opp_q > current_q && current_q == 0))) {
          apply_phase(g, 1 - g.current_phase);
          g.queue_timer = 0;
        }
        break;
      }
      case TS_LIGHTS_PRESSURE_CONTROL:            // run_pressure_control (448-461): raises in the reference
      case TS_LIGHTS_NEIGHBOR_PRESSURE_CONTROL: { // run_neighbor_pressure_control (496-519)
        int ns_p = queue_sum(e, g.ns_in) - queue_sum(e, g.ns_out);
        int ew_p = queue_sum(e, g.ew_in) - queue_sum(e, g.ew_out);
        if (P.light_algorithm == TS_LIGHTS_NEIGHBOR_PRESSURE_CONTROL) {
          const int* nd = g.links_repopulated ? g.nb_dir : g.nb_dir_ctor;
          const int* ngp = g.links_repopulated ? g.nb_grp : g.nb_grp_ctor;
          for (int k = 0; k < 4; k++) {
            if (nd[k] < 0 || ngp[k] < 0) continue;
            const Group& n = e->groups[ngp[k]];
            // The pressures feed back between neighbours and grow geometrically.  In the capture harness they are np.int32
            // scalars (0 + int32 array element) and wrap; spelled out here as unsigned arithmetic.  Under numba the
            // reference's values are unbounded Python ints from that point on: parity unpinned past the first wrap.
            if (nd[k] == 0 || nd[k] == 2) ns_p = (int)((unsigned)ns_p - (unsigned)n.ns_pressure);
            else ew_p = (int)((unsigned)ew_p - (unsigned)n.ew_pressure);
          }
        }
        g.ns_pressure = ns_p; g.ew_pressure = ew_p;
        apply_phase(g, ns_p > ew_p ? 0 : 1);
        break;
      }
      case TS_LIGHTS_NEIGHBOR_GREEN_WAVE: {  // run_neighbor_green_wave (521-546)
        int ns_q = queue_sum(e, g.ns_in), ew_q = queue_sum(e, g.ew_in);
        bool favor_ns = false, favor_ew = false;
        const int* nd = g.links_repopulated ? g.nb_dir : g.nb_dir_ctor;
        const int* ngp = g.links_repopulated ? g.nb_grp : g.nb_grp_ctor;
        for (int k = 0; k < 4; k++) {
          if (nd[k] < 0 || ngp[k] < 0) continue;
          const Group& n = e->groups[ngp[k]];
          if ((nd[k] == 0 || nd[k] == 2) && n.current_phase == 0) favor_ns = true;
          if ((nd[k] == 1 || nd[k] == 3) && n.current_phase == 1) favor_ew = true;
        }
        if (favor_ns && !favor_ew) apply_phase(g, 0);
        else if (favor_ew && !favor_ns) apply_phase(g, 1);
        else apply_phase(g, ns_q > ew_q ? 0 : 1);
        break;
      }
      default: break;
    }
  }
  // _execute_phase_change (348-384)
  if (g.pending_phase < 0) return;
  if (P.transition_duration_enabled && g.transition_timer > 0) {
    g.transition_timer -= 1;
    for (int l : g.lights) light_set(e, l, 1);
    return;
  }
  if (P.transition_clearance_enabled) {
    bool occupied = false;  // is_intersection_occupied (285-291)
    for (int c : g.icells) if (e->occ[c]) { occupied = true; break; }
    if (occupied) { for (int l : g.lights) light_set(e, l, 1); return; }
  }
  if (P.transition_duration_enabled && g.clearance_timer > 0) g.transition_timer = P.all_red_duration;
  // get_opposite_traffic_lights() (303-307): opposite_pairs is empty after construction, so the first
  // call re-runs populate_links(); from then on ns/ew lights and neighbor_groups are complete
  g.links_repopulated = true;
  if (g.pending_phase == 0) {
    for (int l : g.ns_lights) light_set(e, l, 0);
    for (int l : g.ew_lights) light_set(e, l, 1);
  } else {
    for (int l : g.ew_lights) light_set(e, l, 0);
    for (int l : g.ns_lights) light_set(e, l, 1);
  }
  g.current_phase = g.pending_phase;
  g.pending_phase = -1;
}

// ------------------------------ rain (rain.py) ---------------------------------------------------
// math.hypot of CPython 3.10 (Modules/mathmodule.c vector_norm: Veltkamp splitting + Neumaier sums + one
// Newton correction); libm's hypot may differ in the last bit.  Checked against the interpreter on 3e5 inputs.
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

// RainManager.add_random_rain (rain.py:100-148) + RainAgent.__init__ (24-57)
void add_random_rain(E* e) {
  const double w = e->W, h = e->H, off = e->P.rain_spawn_offset;
  const int edge = (int)e->rng_global.randbelow(4);  // random.choice(['N', 'S', 'E', 'W'])
  double x0, y0, xt, yt;
  int corner;  // 0 NW, 1 NE, 2 SW, 3 SE
  if (edge == 0) { x0 = 0.0 + (w - 0.0) * e->rng_global.random(); y0 = h - off; corner = e->rng_global.randbelow(2) ? 3 : 2; }
  else if (edge == 1) { x0 = 0.0 + (w - 0.0) * e->rng_global.random(); y0 = off; corner = e->rng_global.randbelow(2) ? 1 : 0; }
  else if (edge == 2) { x0 = w - off; y0 = 0.0 + (h - 0.0) * e->rng_global.random(); corner = e->rng_global.randbelow(2) ? 2 : 0; }
  else { x0 = off; y0 = 0.0 + (h - 0.0) * e->rng_global.random(); corner = e->rng_global.randbelow(2) ? 3 : 1; }
  if (corner == 0) { xt = 0; yt = h; } else if (corner == 1) { xt = w; yt = h; } else if (corner == 2) { xt = 0; yt = 0; } else { xt = w; yt = 0; }
  double dx = xt - x0, dy = yt - y0;
  double length = py_hypot(dx, dy);
  if (length == 0.0) length = 1.0;
  dx /= length; dy /= length;
  Rain r;
  r.x = x0; r.y = y0;
  double l2 = py_hypot(dx, dy);
  if (l2 == 0.0) l2 = 1.0;
  r.dx = dx / l2; r.dy = dy / l2;
  r.radius = e->rng_global.randint(e->P.rain_radius_min, e->P.rain_radius_max);
  r.sched_idx = (int)e->sched.size();
  e->sched.push_back(SchedEntry{5, (int)e->rains_all.size(), true});
  e->rains.push_back((int)e->rains_all.size());
  e->rains_all.push_back(r);
  e->rain_counter++;
}

void rain_manager_step(E* e) {  // RainManager.step (rain.py:156-184)
  for (int c : e->prev_raining) e->rain[c] = 0;
  if (e->rain_cooldown_left > 0) e->rain_cooldown_left--;
  if ((int)e->rains.size() < e->P.rain_occurrences_max && e->rain_cooldown_left == 0 &&
      e->rng_global.random() < e->P.rain_spawn_chance)
    add_random_rain(e);
  e->prev_raining.clear();
  for (int ri : e->rains)
    for (int c : e->rains_all[ri].covered) { e->rain[c] = 1; e->prev_raining.push_back(c); }
}

void rain_agent_step(E* e, int ri) {  // RainAgent.step (rain.py:60-84)
  Rain& r = e->rains_all[ri];
  r.x += r.dx; r.y += r.dy;
  r.covered.clear();
  const int cx = (int)r.x, cy = (int)r.y, R = r.radius;  // int(): truncation toward zero
  for (int ox = -R; ox <= R; ox++)
    for (int oy = -R; oy <= R; oy++) {
      if (ox * ox + oy * oy > R * R) continue;
      int xi = cx + ox, yi = cy + oy;
      if (xi < 0 || xi >= e->W || yi < 0 || yi >= e->H) continue;
      r.covered.push_back(yi * e->W + xi);
    }
  if (r.x < -R || r.x > e->W + R || r.y < -R || r.y > e->H + R) {
    // manager.on_rain_exit runs while the cloud is still in city_model.rains, so `not rains` is never true and
    // the cooldown never starts (rain.py:150-154, 79-84)
    for (size_t k = 0; k < e->rains.size(); k++) if (e->rains[k] == ri) { e->rains.erase(e->rains.begin() + k); break; }
    e->sched[r.sched_idx].alive = false;
    r.alive = false;
  }
}

// ------------------------------ traffic generator ----------------------------------------------
extern "C" int tso_add_vehicles(ts_handle e, int32_t n, const int32_t* start_xy, const int32_t* goal_xy,
                                const int32_t* population_type, const int32_t* path_off, const int32_t* path_xy);

// _generate_day (dynamic_traffic_generator.py:307-396) for internal and through trips
void generate_day(E* e, int day_idx) {
  Generator& G = e->gen;
  // compute_quotas (319-331): floors + the largest fractional parts (stable, descending)
  auto quotas = [&](int total) {
    const int nz = G.T.n_zones;
    std::vector<double> fc(nz);
    std::vector<int> fl(nz), order(nz);
    long long sum = 0;
    for (int z = 0; z < nz; z++) { fc[z] = (double)total * G.T.zones[z].through_distribution; fl[z] = (int)std::floor(fc[z]); sum += fl[z]; order[z] = z; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fc[a] - std::floor(fc[a]) > fc[b] - std::floor(fc[b]); });
    long long rem = total - sum;
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
      long long cnt = (long long)std::nearbyint((double)G.T.internal_population_per_day * z.fraction[k]);  // round()
      if (cnt == 0) continue;
      std::vector<int> origins, dests;   // city.get_city_blocks_by_type (city_model.py:2040-2055)
      for (size_t b = 0; b < G.blk_type.size(); b++) {
        if (G.blk_type[b] == z.origin_type[k]) origins.push_back((int)b);
        if (G.blk_type[b] == z.dest_type[k]) dests.push_back((int)b);
      }
      if (origins.empty() || dests.empty()) continue;
      for (long long q = 0; q < cnt; q++) {
        double t = z0 + e->rng_global.random() * span;
        int ob = origins[e->rng_global.randbelow((uint32_t)origins.size())];
        int db = dests[e->rng_global.randbelow((uint32_t)dests.size())];
        int oc = G.blk_entr[ob][e->rng_global.randbelow((uint32_t)G.blk_entr[ob].size())];
        int dc = G.blk_entr[db][e->rng_global.randbelow((uint32_t)G.blk_entr[db].size())];
        G.pending.push_back(Trip{oc, dc, t, TS_POP_INTERNAL, day_idx});
      }
    }
    // service vehicles, uniform per zone (362-376): one entrance draw per trip
    const int Nf = food_q[zi], Nw = waste_q[zi];
    for (int j = 1; j <= Nf; j++) {
      double t = z0 + (double)((long long)j * (long long)span) / (double)(Nf + 1);
      int sc = G.hw_in[e->rng_global.randbelow((uint32_t)G.hw_in.size())];
      G.pending.push_back(Trip{sc, -1, t, TS_TRIP_SERVICE_FOOD, day_idx});
    }
    for (int j = 1; j <= Nw; j++) {
      double t = z0 + (double)((long long)j * (long long)span) / (double)(Nw + 1);
      int sc = G.hw_in[e->rng_global.randbelow((uint32_t)G.hw_in.size())];
      G.pending.push_back(Trip{sc, -1, t, TS_TRIP_SERVICE_WASTE, day_idx});
    }
    long long thr = (long long)std::nearbyint((double)G.T.passing_population_per_day * z.through_distribution);
    thr -= Nf + Nw;   // SERVICE_VEHICLES_COUNT_AS_THROUGH defaults to True (90, 381-382)
    if (thr < 0) thr = 0;
    for (long long q = 0; q < thr; q++) {
      double t = z0 + e->rng_global.random() * span;
      int ent = G.hw_in[e->rng_global.randbelow((uint32_t)G.hw_in.size())];
      int ex = G.hw_out[e->rng_global.randbelow((uint32_t)G.hw_out.size())];
      G.pending.push_back(Trip{ent, ex, t, TS_POP_THROUGH, day_idx});
    }
  }
}


// _spawn for service trips (dynamic_traffic_generator.py:419-430) + ServiceVehicleAgent.__init__ (vehicle_service.py:19-41)
// `id` = index into the fleet's id pool, -1 for a vehicle the UI created with an id of its own
void spawn_service_at(E* e, int origin, int kind, int id) {
  Generator& G = e->gen;
  const bool food = kind == TS_TRIP_SERVICE_FOOD;
  // _find_initial_target (62-83): `attempt` is never advanced, so only valid_blocks[0] is ever tried
  int blk = -1;
  for (size_t b = 0; b < G.blocks.size(); b++)
    if (food ? G.blocks[b].needs_food : G.blocks[b].produces_waste) { blk = (int)b; break; }
  int target, phase;
  if (blk >= 0) {
    target = service_road_cell(e, blk);
    if (target < 0) {
      e->fatal = TS_E_UNSUPPORTED;
      e->err = "service vehicle: no free service road cell at its first block (the reference loops forever)";
      return;
    }
    phase = 0;
  } else {
    if (G.hw_out.empty()) { e->fatal = TS_E_UNSUPPORTED; e->err = "service vehicle without highway exits (IndexError in the reference)"; return; }
    target = G.hw_out[0];
    phase = 2;
  }
  if (id >= 0) {
    char& live = G.sv_live[(food ? 0 : G.T.total_service_vehicles_food) + id];
    if (live) {   // BaseScheduler.add raises on a unique_id that is already scheduled (Mesa <= 2.1)
      e->fatal = TS_E_UNSUPPORTED;
      e->err = "service vehicle id drawn while a vehicle with that id is still live (the scheduler raises in the reference)";
      return;
    }
    live = 1;
  }
  int32_t s[2] = {origin % e->W, origin / e->W}, g[2] = {target % e->W, target / e->W};
  int32_t pop = TS_POP_THROUGH;
  e->svc_pending_type = kind;
  tso_add_vehicles(e, 1, s, g, &pop, nullptr, nullptr);
  e->svc_pending_type = 0;
  Vehicle& v = e->veh.back();
  v.svc_id = id; v.svc_phase = phase; v.current_block = blk;
  v.max_load = food ? G.T.service_max_load_food : G.T.service_max_load_waste;
  v.current_load = food ? v.max_load : 0.0;
  v.remove_on_arrival = false;
  v.service_ticks = 0;
}
void spawn_service(E* e, const Trip& t) {
  Generator& G = e->gen;
  const bool food = t.kind == TS_TRIP_SERVICE_FOOD;
  if (food) e->C.created_service_food++; else e->C.created_service_waste++;
  const int pool = food ? G.T.total_service_vehicles_food : G.T.total_service_vehicles_waste;
  const int id = (int)e->rng_global.randbelow((uint32_t)pool);   // vid = random.choice(pool)
  spawn_service_at(e, t.origin, t.kind, id);
}

// DynamicTrafficAgent._update_cached_stats (dynamic_traffic_generator.py:525-648): the sums over city.schedule.agents as they
// are at this point of the shuffled order, the counters, and the daily figures (pending_trips_today 244-248, next_service_eta 278-288)
void update_cached_stats(E* e) {
  Generator& G = e->gen;
  TsCachedStats& c = G.cs;
  memset(&c, 0, sizeof(c));
  c.valid = 1;
  c.update_step = e->C.step_count;
  for (const Vehicle& v : e->veh) {
    if (!v.alive) continue;
    const int k = v.pop_type == TS_POP_INTERNAL ? 0 : v.pop_type == TS_POP_THROUGH ? 1 : -1;
    if (k >= 0) { c.dur_live[k] += e->C.elapsed - v.depart_time; c.dist_live[k] += v.steps_traveled; c.n_live[k]++; }
    if (v.is_stuck) { c.stuck_ticks_sum += v.stuck_ticks; c.stuck_ticks_max = std::max<int64_t>(c.stuck_ticks_max, v.stuck_ticks); }
  }
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
    for (const Trip& t : G.pending) {
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
}

// DynamicTrafficAgent.step (dynamic_traffic_generator.py:153-194) + _spawn (398-416)
void generator_step(E* e) {
  Generator& G = e->gen;
  const double prev = e->C.elapsed;
  e->C.elapsed += e->P.time_per_step_seconds;
  const double total_secs = G.T.start_offset_seconds + e->C.elapsed;
  const int new_day = (int)std::floor(total_secs / 86400.0);
  if (new_day > G.current_day) {
    const long long done = e->C.count_completed_internal + e->C.count_completed_through;
    G.daily_difference_history.push_back((done - G.completed_at_day_start) - (e->C.created_internal + e->C.created_through));   // finished - spawned (165-167)
    G.completed_at_day_start = done;
    for (int dd = G.current_day + 1; dd <= new_day; dd++) generate_day(e, dd);
    G.current_day = new_day;
    e->C.created_internal = 0; e->C.created_through = 0;
    e->C.created_service_food = 0; e->C.created_service_waste = 0;
  }
  std::vector<Trip> keep, spawn;
  for (const Trip& t : G.pending) (prev < t.depart && t.depart <= e->C.elapsed ? spawn : keep).push_back(t);
  for (const Trip& t : spawn) {
    if (e->fatal) break;
    if (t.kind == TS_TRIP_SERVICE_FOOD || t.kind == TS_TRIP_SERVICE_WASTE) { spawn_service(e, t); continue; }
    if (t.kind == TS_POP_INTERNAL) e->C.created_internal++; else e->C.created_through++;
    (void)e->rng_global.randint(0, 9999);  // the id suffix: vid = f"V_{depart:06d}_{randint(0, 9999):04d}"
    int32_t s[2] = {t.origin % e->W, t.origin / e->W}, g[2] = {t.dest % e->W, t.dest / e->W};
    int32_t pop = t.kind;
    tso_add_vehicles(e, 1, s, g, &pop, nullptr, nullptr);
  }
  G.pending.swap(keep);
  // _update_cached_stats every STATISTICS_UPDATE_INTERVAL ticks, inside this step (188-194)
  const int interval = G.T.statistics_update_interval > 0 ? G.T.statistics_update_interval : 20;
  if (++G.ticks_since_stats >= interval) { update_cached_stats(e); G.ticks_since_stats = 0; }
}

// ------------------------------ one tick (city_model.py:1831-1860) -----------------------------
void tick(E* e) {
  // _update_density_map (1853): the map is a function of the occupancy at this point; it is
  // materialised lazily from a snapshot unless eager_density asks for the reference's cost profile
  e->occ_snap = e->occ;
  e->density_valid = false;
  if (e->P.eager_density) update_density_fast(e);
  // run_parallel_decide with one worker (1811-1829): list order; removing the current element while
  // the generator iterates the live list makes the iterator skip the following element.
  // (PATHFINDING_BATCHING=False, 1855: no decide phase - every vehicle decides at the top of its own step)
  {
    size_t w = 0;
    for (size_t i = 0; i < e->active.size(); i++) if (e->active[i] >= 0) e->active[w++] = e->active[i];
    e->active.resize(w);
    bool skip_next = false;
    for (size_t i = 0; e->P.pathfinding_batching && i < e->active.size(); i++) {
      int vid = e->active[i];
      if (vid < 0) continue;
      if (skip_next) { skip_next = false; continue; }
      if (step_decide(e, vid)) skip_next = true;
    }
  }
  // RandomActivation.step: keys in insertion order, shuffled with model.random
  std::vector<int> keys;
  keys.reserve(e->sched.size());
  {
    size_t w = 0;
    bool any_dead = false;
    for (size_t i = 0; i < e->sched.size(); i++) if (!e->sched[i].alive) { any_dead = true; break; }
    if (any_dead) {
      for (size_t i = 0; i < e->sched.size(); i++) {
        if (e->sched[i].alive) {
          if (e->sched[i].kind == 100) e->veh_sched[e->sched[i].ref] = (int)w;
          if (e->sched[i].kind == 5) e->rains_all[e->sched[i].ref].sched_idx = (int)w;
          e->sched[w++] = e->sched[i];
        }
      }
      e->sched.resize(w);
    }
  }
  for (size_t i = 0; i < e->sched.size(); i++) keys.push_back((int)i);
  for (int i = (int)keys.size() - 1; i >= 1; i--) {  // random.shuffle
    uint32_t j = e->rng_sched.randbelow((uint32_t)(i + 1));
    std::swap(keys[i], keys[j]);
  }
  for (int k : keys) {
    if (e->fatal) return;
    SchedEntry se = e->sched[k];
    if (!se.alive) continue;
    switch (se.kind) {
      case 100: vehicle_step(e, se.ref); break;
      case TS_AGENT_LIGHT_GROUP: group_step(e, se.ref); break;
      case TS_AGENT_RAIN_MANAGER: rain_manager_step(e); break;
      case 5: rain_agent_step(e, se.ref); break;
      case TS_AGENT_CITY_BLOCK: block_step(e, se.ref); break;
      case TS_AGENT_CLOCK:
        if (e->gen.armed) generator_step(e);
        else e->C.elapsed += e->P.time_per_step_seconds;
        break;
      default: break;
    }
  }
  e->C.step_count++;
}

uint32_t path_crc(E* e, const Vehicle& v) {
  if (v.plen() == 0) return 0;
  uint32_t crc = 0xFFFFFFFFU;
  for (size_t i = v.head; i < v.path.size(); i++) {
    int32_t xy[2] = {v.path[i] % e->W, v.path[i] / e->W};
    crc = crc_update(crc, xy, 8);
  }
  return crc ^ 0xFFFFFFFFU;
}

}  // namespace

// =============================================================================================
// C-ABI (same signatures as include/trafficsim.h, prefix tso_)
// =============================================================================================
extern "C" {

void tso_default_params(TsParams* p) {
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
  p->stuck_despawn_enabled = 0; p->stuck_despawn_threshold = 3600; p->stuck_despawn_threshold_intersection = 20;
  p->pathfinding_batching = 1;
}

int tso_create(const TsWorld* w, const TsParams* params, ts_handle* out) {
  if (!w || !params || !out || w->width <= 0 || w->height <= 0 || !w->allowed_dirs_map || !w->is_road_map ||
      !w->road_type_map || !w->intersection_map)
    return TS_E_INVALID;
  if (!crc_init_done) crc_init();
  E* e = new E();
  e->W = w->width; e->H = w->height; e->N = w->width * w->height;
  e->P = *params;
  size_t N = e->N;
  e->allowed.assign(w->allowed_dirs_map, w->allowed_dirs_map + N);
  e->is_road.assign(w->is_road_map, w->is_road_map + N);
  e->road_type.assign(w->road_type_map, w->road_type_map + N);
  e->intersection.assign(w->intersection_map, w->intersection_map + N);
  e->occ.assign(N, 0); e->stop.assign(N, 0); e->stuck.assign(N, 0); e->rain.assign(N, 0);
  e->cell_head.assign(N, -1);
  e->occ_snap.assign(N, 0);  // harness rule: _update_density_map() on the fresh (empty) model
  memset(&e->C, 0, sizeof(e->C));
  *out = e;
  return TS_OK;
}

int tso_destroy(ts_handle h) { delete h; return TS_OK; }
const char* tso_last_error(ts_handle h) { return h ? h->err.c_str() : "null handle"; }

int tso_set_lights(ts_handle e, const TsLightTables* t) {
  if (!e || !t) return TS_E_INVALID;
  const int W = e->W, H = e->H;
  auto cell = [&](const int32_t* xy, int i, int& out) {
    int x = xy[2 * i], y = xy[2 * i + 1];
    if (x < 0 || x >= W || y < 0 || y >= H) return false;
    out = y * W + x;
    return true;
  };
  e->groups.assign(t->n_groups, Group());
  e->light_cell.assign(t->n_lights, 0);
  e->light_ctrl.assign(t->n_lights, {});
  for (int l = 0; l < t->n_lights; l++) {
    if (!cell(t->light_xy, l, e->light_cell[l])) return fail(e, TS_E_INVALID, "light cell out of bounds");
    for (int k = t->light_ctrl_off[l]; k < t->light_ctrl_off[l + 1]; k++) {
      int c;
      if (!cell(t->light_ctrl_xy, k, c)) return fail(e, TS_E_INVALID, "controlled block out of bounds");
      e->light_ctrl[l].push_back(c);
    }
  }
  auto fill = [&](std::vector<int>& dst, const int32_t* off, const int32_t* xy, int g) {
    for (int k = off[g]; k < off[g + 1]; k++) {
      int c;
      if (!cell(xy, k, c)) return false;
      dst.push_back(c);
    }
    return true;
  };
  for (int g = 0; g < t->n_groups; g++) {
    Group& G = e->groups[g];
    for (int l = t->g_light_off[g]; l < t->g_light_off[g + 1]; l++) G.lights.push_back(l);
    for (int k = t->g_ns_off[g]; k < t->g_ns_off[g + 1]; k++) G.ns_lights.push_back(t->g_ns[k]);
    for (int k = t->g_ew_off[g]; k < t->g_ew_off[g + 1]; k++) G.ew_lights.push_back(t->g_ew[k]);
    if (!fill(G.icells, t->g_icell_off, t->g_icell_xy, g) || !fill(G.ns_in, t->g_ns_in_off, t->g_ns_in_xy, g) ||
        !fill(G.ns_out, t->g_ns_out_off, t->g_ns_out_xy, g) || !fill(G.ew_in, t->g_ew_in_off, t->g_ew_in_xy, g) ||
        !fill(G.ew_out, t->g_ew_out_off, t->g_ew_out_xy, g))
      return fail(e, TS_E_INVALID, "group cell out of bounds");
    for (int k = 0; k < 4; k++) {
      G.nb_dir[k] = t->g_neighbors ? t->g_neighbors[(g * 4 + k) * 2] : -1;
      G.nb_grp[k] = t->g_neighbors ? t->g_neighbors[(g * 4 + k) * 2 + 1] : -1;
      const int32_t* nc = t->g_neighbors_ctor ? t->g_neighbors_ctor : t->g_neighbors;
      G.nb_dir_ctor[k] = nc ? nc[(g * 4 + k) * 2] : -1;
      G.nb_grp_ctor[k] = nc ? nc[(g * 4 + k) * 2 + 1] : -1;
      if (G.nb_grp[k] >= t->n_groups || G.nb_grp_ctor[k] >= t->n_groups)
        return fail(e, TS_E_INVALID, "neighbor group index out of range");
    }
    // __init__: apply_phase(self._ft_phase) unless DISABLED (intersection_light_group.py:115-116)
    if (e->P.light_algorithm != TS_LIGHTS_DISABLED) apply_phase(G, G.ft_phase);
  }
  e->groups_scheduled = 0;
  return TS_OK;
}

int tso_schedule_add(ts_handle e, int32_t kind, int32_t count) {
  if (!e || count < 0) return TS_E_INVALID;
  for (int i = 0; i < count; i++) {
    SchedEntry se{kind, 0, true};
    if (kind == TS_AGENT_LIGHT_GROUP) {
      if (e->groups_scheduled >= (int)e->groups.size()) return fail(e, TS_E_INVALID, "more group slots than groups");
      se.ref = e->groups_scheduled++;
    } else if (kind == TS_AGENT_CITY_BLOCK) {
      se.ref = e->gen.blocks_scheduled++;
    } else if (kind != TS_AGENT_NOOP && kind != TS_AGENT_CLOCK && kind != TS_AGENT_RAIN_MANAGER) return fail(e, TS_E_INVALID, "bad agent kind");
    e->sched.push_back(se);
  }
  return TS_OK;
}

int tso_set_traffic_generator(ts_handle e, const TsTrafficTables* t) {
  if (!e || !t || t->n_blocks < 0 || t->n_zones < 0 || t->n_zones > 8) return TS_E_INVALID;
  if (!e->seeded[0]) return fail(e, TS_E_STATE, "seed the global stream before constructing the traffic generator");
  Generator& G = e->gen;
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
    if (G.blk_entr[b].empty()) return fail(e, TS_E_UNSUPPORTED, "a city block without entrances (random.choice([]) raises in the reference)");
  }
  G.hw_in.clear(); G.hw_out.clear();
  for (int k = 0; k < t->n_highway_entrances; k++) { int c; if (!cellxy(t->highway_entrances_xy, k, c)) return TS_E_INVALID; G.hw_in.push_back(c); }
  for (int k = 0; k < t->n_highway_exits; k++) { int c; if (!cellxy(t->highway_exits_xy, k, c)) return TS_E_INVALID; G.hw_out.push_back(c); }
  if ((G.hw_in.empty() || G.hw_out.empty()) && t->passing_population_per_day > 0)
    return fail(e, TS_E_UNSUPPORTED, "through traffic needs highway entrances and exits");
  for (int z = 0; z < t->n_zones; z++) if (t->zones[z].n_internal < 0 || t->zones[z].n_internal > 8) return TS_E_INVALID;
  G.blocks.clear();
  if (t->blk_inner_cells) {
    if (!t->blk_service_off || !t->blk_service_xy) return fail(e, TS_E_INVALID, "blk_service_* tables missing");
    G.blocks.resize(t->n_blocks);
    for (int b = 0; b < t->n_blocks; b++) {
      Block& B = G.blocks[b];
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
  if (n_sv > 0 && (G.blocks.empty() || G.hw_in.empty()))
    return fail(e, TS_E_UNSUPPORTED, "service vehicles need the block tables and highway entrances");
  G.sv_live.assign(n_sv, 0);
  G.pending.clear();
  G.current_day = 0;
  G.armed = true;
  generate_day(e, 0);
  return TS_OK;
}

int tso_seed(ts_handle e, int32_t stream, const uint32_t* mt, uint32_t index) {
  if (!e || !mt || stream < 0 || stream > 1 || index > 624) return TS_E_INVALID;
  MT& r = stream == TS_RNG_GLOBAL ? e->rng_global : e->rng_sched;
  memcpy(r.mt, mt, sizeof(r.mt)); r.idx = index;
  e->seeded[stream] = true;
  return TS_OK;
}
int tso_seed_int(ts_handle e, int32_t stream, uint64_t seed) {
  if (!e || stream < 0 || stream > 1) return TS_E_INVALID;
  (stream == TS_RNG_GLOBAL ? e->rng_global : e->rng_sched).seed_int(seed);
  e->seeded[stream] = true;
  return TS_OK;
}
int tso_rng_state(ts_handle e, int32_t stream, uint32_t* mt_out, uint32_t* index_out) {
  if (!e || stream < 0 || stream > 1 || !mt_out || !index_out) return TS_E_INVALID;
  MT& r = stream == TS_RNG_GLOBAL ? e->rng_global : e->rng_sched;
  memcpy(mt_out, r.mt, sizeof(r.mt)); *index_out = r.idx;
  return TS_OK;
}

int tso_add_vehicles(ts_handle e, int32_t n, const int32_t* start_xy, const int32_t* goal_xy,
                     const int32_t* population_type, const int32_t* path_off, const int32_t* path_xy) {
  if (!e || n < 0 || (n > 0 && (!start_xy || !goal_xy))) return TS_E_INVALID;
  const int W = e->W, H = e->H;
  for (int i = 0; i < n; i++) {
    int sx = start_xy[2 * i], sy = start_xy[2 * i + 1], gx = goal_xy[2 * i], gy = goal_xy[2 * i + 1];
    if (sx < 0 || sx >= W || sy < 0 || sy >= H || gx < 0 || gx >= W || gy < 0 || gy >= H)
      return fail(e, TS_E_INVALID, "vehicle start/goal out of bounds");
    if (path_off) {
      int px = sx, py = sy;
      for (int k = path_off[i]; k < path_off[i + 1]; k++) {
        int x = path_xy[2 * k], y = path_xy[2 * k + 1];
        if (x < 0 || x >= W || y < 0 || y >= H || std::abs(x - px) + std::abs(y - py) != 1)
          return fail(e, TS_E_INVALID, "explicit path is not a 4-adjacent in-bounds chain");
        px = x; py = y;
      }
    }
  }
  for (int i = 0; i < n; i++) {
    Vehicle v;
    int vid = (int)e->veh.size();
    v.spawn_idx = vid;
    v.pos = start_xy[2 * i + 1] * W + start_xy[2 * i];
    v.target = goal_xy[2 * i + 1] * W + goal_xy[2 * i];
    v.pop_type = population_type ? population_type[i] : TS_POP_UNDEFINED;
    v.depart_time = e->P.enable_traffic ? e->C.elapsed : 0.0;
    v.svc_type = e->svc_pending_type;
    e->veh.push_back(v);
    // place_vehicle (city_model.py:1897-1918)
    e->active.push_back(vid);
    cell_append(e, e->veh[vid].pos, vid);
    e->occ[e->veh[vid].pos] = 1;
    e->stuck[e->veh[vid].pos] = 0;
    e->veh_sched.push_back((int)e->sched.size());
    e->sched.push_back(SchedEntry{100, vid, true});
    if (e->veh[vid].pop_type == TS_POP_INTERNAL) e->C.live_internal++;
    else if (e->veh[vid].pop_type == TS_POP_THROUGH) {
      e->C.live_through++;
      if (e->veh[vid].svc_type == TS_TRIP_SERVICE_FOOD) e->C.live_service_food++;
      else if (e->veh[vid].svc_type == TS_TRIP_SERVICE_WASTE) e->C.live_service_waste++;
    }
    // self.path = self._compute_path() (vehicle_base.py:80-81)
    std::vector<int> p;
    if (path_off) {
      e->veh[vid].cooldown = e->P.pathfinding_cooldown;
      for (int k = path_off[i]; k < path_off[i + 1]; k++) p.push_back(path_xy[2 * k + 1] * W + path_xy[2 * k]);
    } else {
      compute_path(e, vid, true, p);
    }
    set_path(e->veh[vid], p);
  }
  return TS_OK;
}

int tso_add_vehicles_dirs(ts_handle e, int32_t n, const int32_t* start_xy, const int32_t* goal_xy,
                          const int32_t* population_type, const int64_t* path_off, const uint8_t* path_dirs) {
  if (!e || n < 0 || !path_off || !path_dirs) return TS_E_INVALID;
  // expand to (x, y) chains vehicle by vehicle and reuse the plain entry (keeps memory bounded)
  std::vector<int32_t> xy;
  for (int i = 0; i < n; i++) {
    long long len = path_off[i + 1] - path_off[i];
    xy.resize((size_t)len * 2);
    int x = start_xy[2 * i], y = start_xy[2 * i + 1];
    for (long long k = 0; k < len; k++) {
      int d = path_dirs[path_off[i] + k];
      if (d > 3) return fail(e, TS_E_INVALID, "direction code out of range");
      x += DX[d]; y += DY[d];
      xy[2 * k] = x; xy[2 * k + 1] = y;
    }
    int32_t off[2] = {0, (int32_t)len};
    int rc = tso_add_vehicles(e, 1, start_xy + 2 * i, goal_xy + 2 * i, population_type ? population_type + i : nullptr,
                              off, xy.data());
    if (rc) return rc;
  }
  return TS_OK;
}

int tso_remove_vehicle(ts_handle e, int32_t spawn_idx, int32_t population_type) {   // city_model.py:1920-1941, called between ticks
  if (!e) return TS_E_INVALID;
  if (spawn_idx < 0 || spawn_idx >= (int)e->veh.size() || !e->veh[spawn_idx].alive) { e->err = "no such live vehicle"; return TS_E_INVALID; }
  if (e->veh[spawn_idx].svc_type) { e->err = "service vehicles cannot be removed by the host"; return TS_E_UNSUPPORTED; }
  remove_vehicle(e, spawn_idx, population_type == TS_POP_INTERNAL || population_type == TS_POP_THROUGH ? population_type : TS_POP_UNDEFINED);
  return TS_OK;
}

int tso_upload_map(ts_handle e, int32_t which, const int8_t* src) {
  if (!e || !src) return TS_E_INVALID;
  std::vector<int8_t>* m = which == TS_MAP_STOP ? &e->stop : which == TS_MAP_RAIN ? &e->rain : nullptr;
  if (!m) return fail(e, TS_E_INVALID, "only stop_map and rain_map are host-writable");
  memcpy(m->data(), src, e->N);
  return TS_OK;
}

int tso_step(ts_handle e, int32_t n_ticks) {
  if (!e || n_ticks < 0) return TS_E_INVALID;
  if (e->fatal) return e->fatal;
  if (!e->seeded[0] || !e->seeded[1]) return fail(e, TS_E_STATE, "both RNG streams must be seeded before step");
  for (int t = 0; t < n_ticks; t++) { tick(e); if (e->fatal) return e->fatal; }
  return TS_OK;
}

int tso_num_vehicles(ts_handle e) {
  int n = 0;
  for (int v : e->active) if (v >= 0) n++;
  return n;
}
int tso_num_groups(ts_handle e) { return (int)e->groups.size(); }
int tso_num_scheduled(ts_handle e) {
  int n = 0;
  for (auto& s : e->sched) if (s.alive) n++;
  return n;
}

int tso_download_map(ts_handle e, int32_t which, int8_t* dst) {
  if (!e || !dst) return TS_E_INVALID;
  const std::vector<int8_t>* m = which == TS_MAP_OCCUPANCY ? &e->occ : which == TS_MAP_STOP ? &e->stop
                               : which == TS_MAP_STUCK ? &e->stuck : which == TS_MAP_RAIN ? &e->rain : nullptr;
  if (!m) return TS_E_INVALID;
  memcpy(dst, m->data(), e->N);
  return TS_OK;
}
int tso_download_density(ts_handle e, float* dst) {
  if (!e || !dst) return TS_E_INVALID;
  std::vector<int8_t> keep = e->occ_snap;
  e->occ_snap = e->occ;
  update_density(e);
  memcpy(dst, e->density.data(), sizeof(float) * e->N);
  e->occ_snap.swap(keep);
  e->density_valid = false;
  return TS_OK;
}

int tso_download_vehicles(ts_handle e, int32_t* rows, int32_t cap_rows) {
  if (!e || !rows) return TS_E_INVALID;
  int n = 0;
  for (int vid : e->active) {
    if (vid < 0) continue;
    if (n >= cap_rows) return TS_E_CAPACITY;
    const Vehicle& v = e->veh[vid];
    int32_t* r = rows + (size_t)n * TS_V_NFIELDS;
    r[TS_V_SPAWN_IDX] = v.spawn_idx; r[TS_V_X] = v.pos % e->W; r[TS_V_Y] = v.pos / e->W;
    r[TS_V_BASE_SPEED] = v.base_speed; r[TS_V_CURRENT_SPEED] = v.current_speed; r[TS_V_MAX_STEPS] = v.max_steps;
    r[TS_V_DIRECTION] = v.direction; r[TS_V_STUCK_TICKS] = v.stuck_ticks; r[TS_V_COOLDOWN] = v.cooldown;
    int f = 0;
    if (v.early_exit) f |= TS_F_EARLY_EXIT;
    if (v.is_stuck) f |= TS_F_STUCK;
    if (v.is_parked) f |= TS_F_PARKED;
    if (v.is_in_collision) f |= TS_F_COLLISION;
    if (v.is_in_malfunction) f |= TS_F_MALFUNCTION;
    if (v.is_overtaking) f |= TS_F_OVERTAKING;
    if (v.is_in_stuck_detour) f |= TS_F_DETOUR;
    if (v.blocked_by_vehicle) f |= TS_F_BLOCKED;
    if (v.has_prev) f |= TS_F_HAS_PREV;
    r[TS_V_FLAGS] = f; r[TS_V_STRANDED_LEFT] = v.stranded_left; r[TS_V_STEPS_TRAVELED] = v.steps_traveled;
    r[TS_V_PATH_LEN] = (int)v.plen(); r[TS_V_PATH_CRC] = (int32_t)path_crc(e, v);
    r[TS_V_OVERTAKE_DUR] = v.overtaking_duration; r[TS_V_DETOUR_DUR] = v.stuck_detour_duration;
    n++;
  }
  return n;
}

int tso_num_spawned(ts_handle e) { return e ? (int)e->veh.size() : TS_E_INVALID; }
int tso_download_vehicle_meta(ts_handle e, int32_t* rows, int32_t cap_rows) {
  if (!e || !rows) return TS_E_INVALID;
  int n = 0;
  for (int vid : e->active) {
    if (vid < 0) continue;
    if (n >= cap_rows) return TS_E_CAPACITY;
    const Vehicle& v = e->veh[vid];
    int32_t* r = rows + (size_t)n * TS_M_NFIELDS;
    r[TS_M_SPAWN_IDX] = v.spawn_idx; r[TS_M_POPULATION] = v.pop_type;
    r[TS_M_TARGET_X] = v.target % e->W; r[TS_M_TARGET_Y] = v.target / e->W;
    r[TS_M_VEHICLE_TYPE] = v.svc_type; r[TS_M_SERVICE_PHASE] = v.svc_type ? v.svc_phase : -1;
    n++;
  }
  return n;
}
int tso_download_service_vehicles(ts_handle e, int32_t* spawn_idx, double* loads, int32_t* block, int32_t cap) {
  if (!e || !spawn_idx || !loads || !block) return TS_E_INVALID;
  int n = 0;
  for (int vid : e->active) {
    if (vid < 0 || !e->veh[vid].svc_type) continue;
    if (n >= cap) return TS_E_CAPACITY;
    const Vehicle& v = e->veh[vid];
    spawn_idx[n] = v.spawn_idx; loads[2 * n] = v.current_load; loads[2 * n + 1] = v.max_load; block[n] = v.current_block;
    n++;
  }
  return n;
}
int tso_download_path(ts_handle e, int32_t active_pos, int32_t* xy, int32_t cap_cells) {
  if (!e) return TS_E_INVALID;
  int n = 0;
  for (int vid : e->active) {
    if (vid < 0) continue;
    if (n == active_pos) {
      const Vehicle& v = e->veh[vid];
      int len = (int)v.plen();
      if (xy) {
        if (len > cap_cells) return TS_E_CAPACITY;
        for (int i = 0; i < len; i++) { xy[2 * i] = v.path[v.head + i] % e->W; xy[2 * i + 1] = v.path[v.head + i] / e->W; }
      }
      return len;
    }
    n++;
  }
  return TS_E_INVALID;
}

int tso_download_groups(ts_handle e, int32_t* rows) {
  if (!e || !rows) return TS_E_INVALID;
  for (size_t g = 0; g < e->groups.size(); g++) {
    const Group& G = e->groups[g];
    int32_t* r = rows + g * TS_G_NFIELDS;
    r[TS_G_CURRENT_PHASE] = G.current_phase; r[TS_G_PENDING_PHASE] = G.pending_phase;
    r[TS_G_QUEUE_TIMER] = G.queue_timer; r[TS_G_GAP_TIMER] = G.gap_timer; r[TS_G_LAST_ARRIVAL] = G.last_arrival;
    r[TS_G_FIXED_TIME_TIMER] = G.fixed_time_timer; r[TS_G_FT_PHASE] = G.ft_phase;
    r[TS_G_NS_PRESSURE] = G.ns_pressure; r[TS_G_EW_PRESSURE] = G.ew_pressure;
  }
  return (int)e->groups.size();
}

int tso_num_blocks(ts_handle e) { return e ? (int)e->gen.blocks.size() : TS_E_INVALID; }
int tso_download_blocks(ts_handle e, double* rows) {
  if (!e || !rows) return TS_E_INVALID;
  for (size_t b = 0; b < e->gen.blocks.size(); b++) { rows[2 * b] = e->gen.blocks[b].food; rows[2 * b + 1] = e->gen.blocks[b].waste; }
  return TS_OK;
}
int tso_group_links(ts_handle e, int32_t group, int32_t repopulate) {
  if (!e || group < 0 || group >= (int)e->groups.size()) return TS_E_INVALID;
  if (repopulate) e->groups[group].links_repopulated = true;
  return e->groups[group].links_repopulated ? 1 : 0;
}
int tso_add_service_vehicle(ts_handle e, int32_t x, int32_t y, int32_t service_type) {
  if (!e || x < 0 || x >= e->W || y < 0 || y >= e->H) return TS_E_INVALID;
  if (service_type != TS_TRIP_SERVICE_FOOD && service_type != TS_TRIP_SERVICE_WASTE) return fail(e, TS_E_INVALID, "service_type");
  if (e->gen.blocks.empty()) return fail(e, TS_E_STATE, "service vehicles need the block tables (ts_set_traffic_generator)");
  if (e->fatal) return e->fatal;
  spawn_service_at(e, y * e->W + x, service_type, -1);
  return e->fatal;
}
int tso_rain_info(ts_handle e, TsRainInfo* out) {
  if (!e || !out) return TS_E_INVALID;
  bool has = false;
  for (const auto& se : e->sched) if (se.alive && se.kind == TS_AGENT_RAIN_MANAGER) has = true;
  out->has_manager = has; out->n_rains = (int32_t)e->rains.size();
  out->cooldown = e->rain_cooldown_left; out->counter = e->rain_counter;
  return TS_OK;
}
int tso_rain_spawn(ts_handle e) {
  if (!e) return TS_E_INVALID;
  bool has = false;
  for (const auto& se : e->sched) if (se.alive && se.kind == TS_AGENT_RAIN_MANAGER) has = true;
  if (!has) return fail(e, TS_E_STATE, "no RainManager is scheduled");
  if (!e->seeded[0]) return fail(e, TS_E_STATE, "seed the global stream first");
  add_random_rain(e);
  return TS_OK;
}
int tso_counters(ts_handle e, TsCounters* out) {
  if (!e || !out) return TS_E_INVALID;
  *out = e->C;
  return TS_OK;
}

int tso_astar(ts_handle e, int32_t sx, int32_t sy, int32_t gx, int32_t gy, int32_t soft, int32_t ignore_flow,
              int32_t maximum_steps, int32_t* out_xy, int32_t cap_cells) {
  if (!e) return TS_E_INVALID;
  if (sx < 0 || sx >= e->W || sy < 0 || sy >= e->H || gx < 0 || gx >= e->W || gy < 0 || gy >= e->H)
    return fail(e, TS_E_INVALID, "astar endpoints out of bounds");
  std::vector<int> p;
  std::vector<int8_t> keep;
  if (soft) { keep = e->occ_snap; e->occ_snap = e->occ; update_density(e); }
  astar(e, sx, sy, gx, gy, soft != 0, ignore_flow != 0, maximum_steps, p);
  if (soft) { e->occ_snap.swap(keep); e->density_valid = false; }
  if ((int)p.size() > cap_cells) return TS_E_CAPACITY;
  for (size_t i = 0; i < p.size(); i++) { out_xy[2 * i] = p[i] % e->W; out_xy[2 * i + 1] = p[i] / e->W; }
  return (int)p.size();
}

int tso_set_device(int32_t) { return TS_OK; }
// (the checker is a single sequential process: there is nothing to shard)
int tso_cached_stats(ts_handle e, TsCachedStats* out) {
  if (!e || !out) return TS_E_INVALID;
  *out = e->gen.cs;
  return TS_OK;
}
int tso_set_replan_sharding(ts_handle, int32_t, int32_t world, ts_exchange_fn, void*) { return world == 1 ? TS_OK : TS_E_UNSUPPORTED; }
int tso_set_replan_sharding_device(ts_handle, int32_t, int32_t world, ts_exchange_fn, void*) { return world == 1 ? TS_OK : TS_E_UNSUPPORTED; }
int tso_profile_enable(ts_handle, int32_t) { return TS_OK; }
int tso_profile_count(void) { return 0; }
const char* tso_profile_name(int32_t) { return ""; }
int tso_profile_get(ts_handle, int32_t, double*, int64_t*, int64_t*) { return TS_E_INVALID; }

/* test hook: write occupancy directly (A* KATs need an arbitrary occupancy map) */
int tso_debug_set_occupancy(ts_handle e, const int8_t* src) {
  if (!e || !src) return TS_E_INVALID;
  memcpy(e->occ.data(), src, e->N);
  return TS_OK;
}

#ifdef TSO_STATS
void tso_stats_read(long long* out) { memcpy(out, &g_astats, sizeof g_astats); }   // 64 + 40 + 40 + 5 + 40 + 2 words
#endif
}  // extern "C"
