// astar_quad.h - the replanning searches, sixteen to a wavefront (k_replan_quad).
//
// k_replan (astar.h) spreads ONE search over the 64 lanes of a wave: 138 vector + 151 scalar instructions per expansion,
// issued for one expansion's worth of work (profiles/r03_sq_replan_4096.json).  Here a search owns a QUAD of four lanes -
// lane j of the quad is direction j of astar_numba.py's neighbour loop (N, E, S, W) - and the sixteen quads of a wave run
// sixteen independent searches through one branch-light loop in lockstep: every instruction issued serves sixteen
// expansions, nothing is wave-uniform (no scalar unit work beyond loop control), and the cross-lane traffic is quad_perm
// DPP, which is free.  The algorithm is astar_core's, slot for slot and compare for compare:
//   * heap: the reference's binary heap as 8-byte (f, cell) entries, the first TS_QUAD_LCAP slots of every search in LDS
//     (16 x 156 x 8 B = 19.5 KB per wave: seven waves per CU beside k_replan's side waves), deeper slots in the search's
//     HBM spill.  Sift-down, two levels per LDS round trip: lanes 0 / 1 of the quad fetch the hole's two children, the four
//     lanes its four grandchildren, one DPP swap shows each its sibling's key, "my entry moves up" is one compare (strict
//     '<', ties to the parent, then to the left child: astar_numba.py:67-85).  Sift-up: the four lanes fetch four
//     ancestors of the new slot at once (52-65);
//   * dir_arr (indexed by heap SLOT, never moved by the sifts: the stale-slot quirk) is only ever read at the heap's
//     last slot and written at the slots behind it - a stack.  Two bits per slot, the 32 slots around the heap's end in
//     a register pair, the rest in HBM; slot 0 in a register of its own;
//   * dist / came_from: one 4-byte record per CELL (dist 22 bits | direction 2 | epoch stamp 8) in a table that is a
//     1152 x 1152-cell window centred on the search's start (8 x 8-tiled, addressed by the cell's offset in it): the record's address
//     does not depend on the map entry, so both loads of an expansion leave together (k_replan's node-numbered table costs
//     a dependent second round trip), and a searcher costs 5.3 MB instead of 8 bytes per road cell of the whole map.  A
//     search that reaches a cell 576 or more columns / rows away from its start, whose g reaches 2^22, whose heap outgrows
//     the spill or whose path outgrows the buffers is abandoned, and its vehicle goes to k_replan's queue (nothing of it
//     has been committed: step_decide is a pure function of the tick-start state until its last lines);
//   * half-unit integer costs only (astar_half_units: the reference's defaults), no step limit, no contraflow, no
//     field-of-view mask: the searches that need those (bypasses: a few hundred expansions each, rare) take the vehicle to
//     k_replan as well.
// The policy around the searches (decide_vehicle<DM_QUAD>: step_decide with _compute_path_internal's phases) is run by the
// quad like k_replan runs it by the wave - every lane the same code on the same values - but it cannot keep a search on its
// call stack while fifteen other quads advance theirs.  It is re-run from the top after every search instead: finished
// searches are taken from a log (length + counters; their paths already sit in the buffers the policy handed out), the
// first unfinished one suspends the pass (DV_SUSPEND) and becomes the quad's search.
// Measured: 86 instructions per expansion against k_replan's 289, but only ~50 heaps of this workload
// fit the LDS of a CU whatever the kernel, so the quads run at one to two waves per SIMD with half of each heap in HBM: worth
// 25-30 % on a replanning wave, and slower than k_replan on a queue that is bounded by its longest search - run_replans
// (engine.hip) sends them queues of TS_QUAD_MIN = 262 144 entries and more.
#pragma once
#include "astar.h"

namespace {

#ifndef TS_QUAD_LCAP
#define TS_QUAD_LCAP 156
#endif
#ifndef TS_QUAD_WAVES_PER_CU
#define TS_QUAD_WAVES_PER_CU 7
#endif
constexpr int QL = TS_QUAD_LCAP;            // heap slots per search in LDS (and the stride between two searches' heaps)
// (bank spread: the 8-byte slots of the eight quads of a half-wave at the same heap index must fall into different banks)
static_assert(QL % 8 == 4, "TS_QUAD_LCAP must be 4 mod 8 (LDS bank spread between the quads of a wave)");
static_assert(((size_t)16 * QL + 64) * 8 * TS_QUAD_WAVES_PER_CU <= 160 * 1024 - 512, "the LDS heaps of TS_QUAD_WAVES_PER_CU waves must fit a CU");
#ifndef TS_QUAD_WINDOW
#define TS_QUAD_WINDOW 1152
#endif
constexpr int QT_MAX = TS_QUAD_WINDOW;       // table window: at most 1152 x 1152 cells around the search's start (a multiple of 8)
static_assert(QT_MAX % 8 == 0, "the table window is 8 x 8-tiled");
constexpr int Q_CELLS = 4096;               // path buffer capacity per search (cells)
constexpr int Q_SPILL = 7936;               // heap slots per search beyond LDS (HBM)
#ifndef TS_QUAD_MAX_EXP
#define TS_QUAD_MAX_EXP (1 << 17)
#endif
constexpr int Q_MAX_EXP = TS_QUAD_MAX_EXP;  // expansions after which a quad hands its search (and vehicle) to k_replan
constexpr uint32_t Q_DIST_MASK = (1u << 22) - 1, Q_STAMP_SHIFT = 24, Q_DIR_SHIFT = 22;
enum { QS_NEEDJOB = 0, QS_POLICY = 1, QS_SEARCH = 2, QS_FOUND = 3, QS_EMPTY = 4, QS_ABANDON = 5, QS_IDLE = 6 };

struct QSlots {
  int n_slots;            // searches = quads: sixteen per wave
  int tw, th;             // table window in cells, multiples of 8 (the whole map when it is smaller)
  int chk_x, chk_y;       // the map is wider / taller than the window: it is centred on the search's start and relaxations are bounded to it
  size_t tab_entries;     // per slot
  uint32_t* tab;
  unsigned long long* gq; // per slot Q_SPILL entries
  uint32_t* gdw;          // per slot (QL + Q_SPILL) / 16 + 4 words of 2-bit directions
  int32_t* cells;         // per slot 5 * Q_CELLS + 3 * MAXB
  int32_t* log;           // per slot 3 * QLOG
  uint32_t* slot_epoch;
};
constexpr int Q_DWORDS = (QL + Q_SPILL) / 16 + 4;
constexpr int Q_HEAP_MAX = (QL + Q_SPILL) < (32 * QL - 2) ? (QL + Q_SPILL) : (32 * QL - 2);   // largest heap a quad carries

__shared__ unsigned long long q_lds[16 * QL + 64];     // (the last 64 entries: one scratch slot per lane, see q_lput_if)

// quad_perm DPP: lane j of every quad reads lane P[j] of its quad
template <int CTRL> __device__ __forceinline__ int qperm(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
constexpr int QP_SWAP1 = 0xB1;   // [1,0,3,2]
constexpr int QP_SWAP2 = 0x4E;   // [2,3,0,1]
constexpr int QP_B0 = 0x00, QP_B1 = 0x55, QP_B2 = 0xAA, QP_B3 = 0xFF;   // broadcasts of lane 0 .. 3
__device__ __forceinline__ int quad_or(int v) {   // OR over the four lanes of the quad, in every lane
  v |= qperm<QP_SWAP1>(v);
  v |= qperm<QP_SWAP2>(v);
  return v;
}

// what a quad keeps about its search while the lockstep loop runs (every lane of the quad holds the same values)
struct QState {
  int hs;                   // heap_size
  int dir0;                 // dir_arr[0]
  int wb;                   // first slot of the direction window (a multiple of 16)
  unsigned long long dwin;  // dir_arr[wb .. wb + 32), two bits per slot
  int gx, gy, sx, sy;
  int ox, oy;               // origin of the table window
  uint32_t goal_xy;         // x | y << 16
  uint32_t stamp;           // epoch << 24
  int soft;
  int n_exp, n_relax;
  unsigned long long xpre;  // heap entry hs - 1 when it lies beyond LDS (requested at the end of the previous turn)
  int why;                  // why the search was abandoned (1 window / g, 2 heap, 3 expansion budget, 4 path buffer): ts_debug statistics
#ifdef TS_QUAD_PROF
  long long pf[8], pt;
#endif
};
#ifdef TS_QUAD_PROF
#define QP(k) do { const long long _t = clock64(); s.pf[k] += _t - s.pt; s.pt = _t; } while (0)
#else
#define QP(k) do { } while (0)
#endif
struct QConst {   // per quad, fixed for the kernel's lifetime
  TS_GLOBAL uint32_t* tab;
  gu64p gq;
  TS_GLOBAL uint32_t* gdw;
  int lbase;                // first slot of this quad's heap in q_lds
  int j;                    // lane of the quad = neighbour direction
  int dx, dy;
  // wave-uniform
  const TS_GLOBAL u64* amap;
  int W, H, W8, tw, th, tw8, chk_x, chk_y;
  int turn2, stop2, rt2_1, rt2_2, rt2_3;
  bool turn_on, rt_on;
};

// the table record of cell (x, y): the window's origin (ox, oy) is the search's start minus half a window when the map is
// larger than the window, (0, 0) otherwise
__device__ __forceinline__ uint32_t q_tix(const QConst& K, int ox, int oy, int x, int y) {
  const uint32_t xm = (uint32_t)(x - ox), ym = (uint32_t)(y - oy);
  return ((__umul24(ym >> 3, (uint32_t)K.tw8) + (xm >> 3)) << 6) | ((ym & 7u) << 3) | (xm & 7u);
}
__device__ __forceinline__ uint32_t q_aix(const QConst& K, int x, int y) {
  return (((__umul24((uint32_t)(y >> 3), (uint32_t)K.W8) + (uint32_t)(x >> 3)) << 6) | (uint32_t)((y & 7) << 3) | (uint32_t)(x & 7));
}
// heap slot k of the quad's search: LDS below QL, the search's HBM spill above.  The turn's hot path uses the LDS forms
// only (no branch around a memory operation, no wait for the expansion's loads in flight); slots that may lie beyond QL
// are touched in blocks of their own, which only run for quads whose heap has outgrown LDS.
__device__ __forceinline__ u64 q_lget(const QConst& K, int k) { return q_lds[K.lbase + k]; }
__device__ __forceinline__ void q_lput(const QConst& K, int k, u64 v) { q_lds[K.lbase + k] = v; }
// A store some lanes make and others do not, without a branch: the others write their own scratch slot behind the heaps.  (A
// predicated store costs the wave an exec-mask bracket and a branch - three scalar instructions and a pipeline bubble; the
// turn has a dozen of them, and at one or two waves per SIMD nothing hides them.)
__device__ __forceinline__ void q_lput_if(const QConst& K, bool on, int k, u64 v) { q_lds[on ? K.lbase + k : 16 * QL + (int)threadIdx.x] = v; }
// heap slot k for the lanes that want it (on) and whose slot lies in LDS; the store of a slot beyond LDS stays behind a branch
// (one, and only quads whose heap has outgrown LDS take it)
__device__ __forceinline__ void q_hput_if(const QConst& K, bool on, int k, u64 v) {
  q_lput_if(K, on & (k < QL), k, v);
  if (on & (k >= QL)) K.gq[k - QL] = v;
}
// A load from the HBM spill in one of the turn's rare side paths, waited for on the spot and kept out of the compiler's
// bookkeeping of outstanding memory operations: a load it can see behind a branch costs an unconditional
// `s_waitcnt vmcnt(0)` where the branch rejoins - for every quad of the wave, on every turn, and with the turn's stores
// (table records, heap entries) in flight that is a wait of microseconds.
__device__ __forceinline__ u64 q_gload_now(const TS_GLOBAL u64* p) {
  u64 v;
  asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ u64 q_hget(const QConst& K, int k) {
  u64 v;
  if (k < QL) v = q_lds[K.lbase + k]; else v = q_gload_now(K.gq + (k - QL));
  return v;
}
__device__ __forceinline__ void q_hput(const QConst& K, int k, u64 v) {
  if (k < QL) q_lds[K.lbase + k] = v; else K.gq[k - QL] = v;
}

// Two levels of heap_sift_down (astar_numba.py:67-85) below the hole at slot p: lanes 0 / 1 of the quad fetch the hole's
// children (2 / 3 mirror them), the four lanes its four grandchildren (quad_sift2_load); every lane decides with its
// sibling's key (one DPP swap) whether its entry would move up if its parent were the hole - smallest of (x, left, right),
// ties to x, then to the left - and one OR over the quad tells every lane the path (quad_sift2_apply).  apply returns
// true when x has found its place (p).  `wr0` / `wr1`: the slots it wrote (-1: none), for whoever holds copies of them.
struct QPair { u64 ce, ge; bool cv, gv; };
template <bool DEEP>
__device__ __forceinline__ QPair quad_sift2_load(const QConst& K, int p, int size) {
  const int j = K.j, side = j & 1;
  const int c = 2 * p + 1 + side, gc = 4 * p + 3 + j;
  QPair r;
  r.cv = c < size; r.gv = gc < size;
  r.ce = 0; r.ge = 0;
  if (DEEP) {
    if (r.cv) r.ce = q_hget(K, c);
    if (r.gv) r.ge = q_hget(K, gc);
  } else {       // (no branch around the reads: a slot that does not exist is read at the root and not used)
    r.ce = q_lget(K, r.cv ? c : 0);
    r.ge = q_lget(K, r.gv ? gc : 0);
  }
  return r;
}
// The same fetch with nothing conditional about it: an LDS read and an HBM load per entry, both always issued (clamped to
// a valid address when the slot lies on the other side or does not exist), the right one picked afterwards.  A load the
// compiler can count is a load it can wait for precisely: behind a branch it would make every later wait for an older
// load (the expansion's map entries and table records) a wait for all of them.
struct QBoth { u64 cl, cg, gl, gg; int c, gc; bool cv, gv; };
__device__ __forceinline__ QBoth quad_sift2_load_both(const QConst& K, int p, int size) {
  const int j = K.j, side = j & 1;
  QBoth r;
  r.c = 2 * p + 1 + side; r.gc = 4 * p + 3 + j;
  r.cv = r.c < size; r.gv = r.gc < size;
  r.cl = q_lget(K, min(r.c, QL - 1)); r.gl = q_lget(K, min(r.gc, QL - 1));
  r.cg = K.gq[max(min(r.c, QL + Q_SPILL - 1) - QL, 0)]; r.gg = K.gq[max(min(r.gc, QL + Q_SPILL - 1) - QL, 0)];
  return r;
}
template <bool DEEP>
__device__ __forceinline__ bool quad_sift2_apply(const QConst& K, int& p, int xf, const QPair& r, int& wr0, int& wr1) {
  const int j = K.j, side = j & 1;
  const int cf = r.cv ? hq_f(r.ce) : 0x7FFFFFFF, gf = r.gv ? hq_f(r.ge) : 0x7FFFFFFF;
  const int csf = qperm<QP_SWAP1>(cf), gsf = qperm<QP_SWAP1>(gf);
  const unsigned adj = (unsigned)(side ^ 1);
  const bool cw = (unsigned)cf < min((unsigned)xf, (unsigned)csf + adj);
  const bool gw = (unsigned)gf < min((unsigned)xf, (unsigned)gsf + adj);
  const int bits = quad_or((cw ? (1 << j) : 0) | (gw ? (16 << j) : 0));
  const int w1 = bits & 3;
  wr0 = -1; wr1 = -1;
  if (w1 == 0) return true;
  const int sdn = w1 >> 1;                              // the hole moves to the left (0) / right (1) child
  const int pc = 2 * p + 1 + sdn;
  const int w2 = (bits >> (4 + 2 * sdn)) & 3;
  const int t = 2 * sdn + (w2 >> 1);                    // ... and on to grandchild t (if w2)
  if (DEEP) {
    q_hput_if(K, cw & (j < 2), p, r.ce);
    q_hput_if(K, (w2 != 0) & (j == t), pc, r.ge);
  } else {
    q_lput_if(K, cw & (j < 2), p, r.ce);
    q_lput_if(K, (w2 != 0) & (j == t), pc, r.ge);
  }
  wr0 = p;
  if (w2 == 0) { p = pc; return true; }
  wr1 = pc;
  p = 4 * p + 3 + t;
  return false;
}

// One turn of astar_core's main loop (astar_numba.py:136-237) for every quad whose search is on.  Returns the quad's
// new state: QS_SEARCH, QS_FOUND (the goal was popped), QS_EMPTY (heap empty: `return []`) or QS_ABANDON.
// Order of the turn (what the reference does one after the other is only reordered where the results cannot tell):
//   pop -> the expansion's loads leave (map entries, table records; the parents of the next two heap slots when they lie
//   beyond LDS) -> sift-down through the levels in LDS -> the loads of its first two levels beyond LDS leave -> the popped
//   cell's neighbours are evaluated (registers only) while those travel -> the sift-down finishes -> table records, pushes.
__device__ __forceinline__ int quad_turn(const QConst& K, QState& s) {
  if (s.hs <= 0) return QS_EMPTY;
  const int j = K.j;
  QP(7);
  // ---- the direction window follows the heap's end: slots hs - 1 (read by this pop) .. hs + 2 (written by its pushes)
  if (s.hs + 2 > s.wb + 31) {
    K.gdw[s.wb >> 4] = (uint32_t)s.dwin;
    s.dwin = (s.dwin >> 32) | ((u64)K.gdw[(s.wb >> 4) + 2] << 32);
    s.wb += 16;
  } else if (s.hs - 1 < s.wb) {
    K.gdw[(s.wb >> 4) + 1] = (uint32_t)(s.dwin >> 32);
    s.dwin = (s.dwin << 32) | (u64)K.gdw[(s.wb >> 4) - 1];
    s.wb -= 16;
  }
  // ---- pop (138-148).  The last entry (it takes the root's place) was requested at the end of the previous turn when it
  // lies beyond LDS
  const u64 root = q_lget(K, 0);
  const int last_i = s.hs - 1;
  const int f_top = hq_f(root);
  const uint32_t cxy = (uint32_t)hq_i(root);
  const int cx = (int)(cxy & 0xFFFFu), cy = (int)(cxy >> 16);
  const int prev_dir = s.dir0;
  const int xd = (int)((s.dwin >> (2 * ((last_i - s.wb) & 31))) & 3ull);
  s.hs = last_i;
  // everything the expansion reads from HBM leaves now and travels while the sift-down works: lane j the map entry and
  // the table record of neighbour j, every lane (one request) the popped cell's own
  const int nx = cx + K.dx, ny = cy + K.dy;
  const bool inb = (unsigned)nx < (unsigned)K.W && (unsigned)ny < (unsigned)K.H;
  const uint32_t a_ixc = q_aix(K, cx, cy), t_ixc = q_tix(K, s.ox, s.oy, cx, cy);
  const uint32_t a_ix = inb ? q_aix(K, nx, ny) : a_ixc;
  const bool inw = inb & ((unsigned)(nx - s.ox) < (unsigned)K.tw) & ((unsigned)(ny - s.oy) < (unsigned)K.th);
  const uint32_t t_ix = inw ? q_tix(K, s.ox, s.oy, nx, ny) : t_ixc;
  const u64 am_l = ld8(K.amap, a_ix);
  const uint32_t t_l = K.tab[t_ix];
  const uint32_t am_c = *(const TS_GLOBAL uint32_t*)((const TS_GLOBAL char*)K.amap + (uint32_t)(a_ixc << 3));
  const uint32_t t_c = K.tab[t_ixc];
  // the pushes of this turn go to slots last_i, last_i + 1, ...: lane j will want ancestor 1 + j of each, ((i + 1) >> (1 + j)) - 1.
  // Those that lie beyond LDS are requested now, each lane its own two (pa0 / pa1: the slots, -1 = not held or in LDS)
  int pa0 = (int)(((unsigned)(last_i + 1) >> (1 + j)) - 1u), pa1 = (int)(((unsigned)(last_i + 2) >> (1 + j)) - 1u);
  pa0 = pa0 >= QL ? pa0 : -1;
  pa1 = pa1 >= QL ? pa1 : -1;
  // (both loads always leave - of the spill's first entry when there is nothing to fetch: see quad_sift2_load_both)
  const u64 pv0 = K.gq[max(pa0 - QL, 0)], pv1 = K.gq[max(pa1 - QL, 0)];
  QP(0);
  int p = 0;
  bool sinking = false;
  u64 x = 0;
  int xf = 0;
  if (last_i > 0) {
    s.dir0 = xd;
    const u64 xl = q_lget(K, min(last_i, QL - 1));
    x = last_i < QL ? xl : s.xpre;
    xf = hq_f(x);
    bool done = false;
    int w0, w1;
    // (a step reads slots up to 4 p + 6: LDS-only code while those that exist all lie below QL)
    while (!done && (last_i <= QL || 4 * p + 6 < QL)) {
      const QPair r = quad_sift2_load<false>(K, p, last_i);
      done = quad_sift2_apply<false>(K, p, xf, r, w0, w1);
    }
    sinking = !done;
  }
  const QBoth db = quad_sift2_load_both(K, sinking ? p : 0, sinking ? last_i : 0);     // (on their way while the neighbours are evaluated)
  QP(1);
  // ---- lane j evaluates neighbour j (171-225) in half units, exactly as astar_loop<HALF> does: registers only
  const bool is_goal = cxy == s.goal_xy;                        // (151: before the staleness test)
  const int g = f_top - (abs(cx - s.gx) + abs(cy - s.gy));
  const int dist_c = ((t_c ^ s.stamp) >> Q_STAMP_SHIFT) == 0u ? (int)(t_c & Q_DIST_MASK) : A_INF;
  const bool stale = g > dist_c;                                // (165)
  const uint32_t a_l = (uint32_t)am_l;
  const bool node_l = (uint32_t)(am_l >> 32) != 0xFFFFFFFFu;
  const int dist_l = ((t_l ^ s.stamp) >> Q_STAMP_SHIFT) == 0u ? (int)(t_l & Q_DIST_MASK) : A_INF;
  const bool n_occ = ((a_l >> 8) & 1u) != 0u, n_stop = ((a_l >> 9) & 1u) != 0u, n_road = ((a_l >> 4) & 1u) != 0u;
  const uint32_t rt = (a_l >> 6) & 3u;
  const bool flow = ((am_c >> j) & 1u) != 0u;
  const bool turn = K.turn_on & (prev_dir != -1) & (j != prev_dir);
  int n2 = 2 * (g + 1);
  n2 += turn ? K.turn2 : 0;
  n2 += n_occ ? (int)(a_l >> AMAP_PEN_SHIFT) : 0;
  n2 += n_stop ? K.stop2 : 0;
  const int rtp = rt == 1u ? K.rt2_1 : rt == 2u ? K.rt2_2 : rt == 3u ? K.rt2_3 : 0;
  n2 += (K.rt_on & n_road) ? rtp : 0;
  const bool cand = !is_goal & !stale & inb & node_l & flow & ((s.soft != 0) | !(n_occ | n_stop));
  const bool ok = cand & inw & (n2 < 2 * dist_l);
  const int ngi = n2 >> 1;
  // what this searcher cannot carry: a g beyond the record's 22 bits, a neighbour outside the table window (its record
  // cannot be consulted: whether it would be relaxed is unknowable here)
  const bool bad = (ok & (ngi > (int)Q_DIST_MASK)) | (cand & !inw);
  int relax = quad_or((ok ? (1 << j) : 0) | (bad ? 16 : 0));
  const int nf_l = ngi + abs(nx - s.gx) + abs(ny - s.gy);
  const int nxy_l = (int)((uint32_t)nx | ((uint32_t)ny << 16));
  QP(3);
  // ---- the sift-down finishes beyond LDS.  (Every path takes delivery of the two entries here, before the turn's stores
  // leave: a load still pending on some path would be waited for behind them - stores included - where its register is
  // next written.)
  asm volatile("" : : "v"(relax), "v"(nf_l), "v"(nxy_l), "v"(ngi));   // (the evaluation first ...)
  asm volatile("" : : "v"(db.cg), "v"(db.gg), "v"(pv0), "v"(pv1));
  QPair dp;
  dp.cv = db.cv; dp.gv = db.gv;
  dp.ce = db.c < QL ? db.cl : db.cg;
  dp.ge = db.gc < QL ? db.gl : db.gg;
  if (last_i > 0) {
    if (sinking) {
      bool done;
      int w0, w1;
      done = quad_sift2_apply<true>(K, p, xf, dp, w0, w1);
      if ((w0 >= 0) & ((w0 == pa0) | (w0 == pa1))) { if (w0 == pa0) pa0 = -1; if (w0 == pa1) pa1 = -1; }
      if ((w1 >= 0) & ((w1 == pa0) | (w1 == pa1))) { if (w1 == pa0) pa0 = -1; if (w1 == pa1) pa1 = -1; }
      while (!done) {
        const QPair r = quad_sift2_load<true>(K, p, last_i);
        done = quad_sift2_apply<true>(K, p, xf, r, w0, w1);
        if ((w0 >= 0) & ((w0 == pa0) | (w0 == pa1))) { if (w0 == pa0) pa0 = -1; if (w0 == pa1) pa1 = -1; }
        if ((w1 >= 0) & ((w1 == pa0) | (w1 == pa1))) { if (w1 == pa0) pa0 = -1; if (w1 == pa1) pa1 = -1; }
      }
    }
    q_hput_if(K, j == 0, p, x);
    if (p == pa0) pa0 = -1;
    if (p == pa1) pa1 = -1;
  }
  wave_mem_sync();
  QP(2);
  int st = QS_SEARCH;
  do {
    if (is_goal) { st = QS_FOUND; break; }
    if (stale) break;
    s.n_exp++;
    if (s.n_exp > Q_MAX_EXP) { st = QS_ABANDON; s.why = 3; break; }   // a long search: k_replan's single search is the faster one
    if (relax & 16) { st = QS_ABANDON; s.why = 1; break; }
    if (relax == 0) break;
    const int n_new = __builtin_popcount((unsigned)relax);
    // (a heap the spill holds and whose slots' ancestors 5 and up all lie in LDS: (i + 1) / 32 - 1 < QL)
    if (s.hs + n_new > Q_HEAP_MAX) { st = QS_ABANDON; s.why = 2; break; }
    s.n_relax += n_new;
    if (ok) K.tab[t_ix] = (uint32_t)ngi | ((uint32_t)j << Q_DIR_SHIFT) | s.stamp;   // dist / came_from (226-227)
    const int nf0 = qperm<QP_B0>(nf_l), nf1 = qperm<QP_B1>(nf_l), nf2 = qperm<QP_B2>(nf_l), nf3 = qperm<QP_B3>(nf_l);
    const int nc0 = qperm<QP_B0>(nxy_l), nc1 = qperm<QP_B1>(nxy_l), nc2 = qperm<QP_B2>(nxy_l), nc3 = qperm<QP_B3>(nxy_l);
    QP(4);
    // ---- heap pushes in the reference's order N, E, S, W (229-237)
    int npush = 0;
    while (relax) {
      const int dd = __builtin_ctz((unsigned)relax);
      relax &= relax - 1;
      const int nf = dd == 0 ? nf0 : dd == 1 ? nf1 : dd == 2 ? nf2 : nf3;
      const int nxy = dd == 0 ? nc0 : dd == 1 ? nc1 : dd == 2 ? nc2 : nc3;
      const int i = s.hs;
      if (i == 0) s.dir0 = dd;
      else {
        const int sh = 2 * ((i - s.wb) & 31);
        s.dwin = (s.dwin & ~(3ull << sh)) | ((u64)(unsigned)dd << sh);
      }
      // heap_sift_up (52-65): the four lanes fetch four ancestors of slot i at a time; ancestor k = ((i + 1) >> k) - 1.
      // (ancestors 1-4 of a slot beyond 2 QL can lie beyond LDS: those of the first two pushes of a turn were requested at
      // the pop, unless something has written to them since; the rest are fetched like any slot)
      const int depth = 31 - __builtin_clz((unsigned)(i + 1));
      const int pa = npush == 0 ? pa0 : npush == 1 ? pa1 : -1;
      const u64 pv = npush == 0 ? pv0 : pv1;
      int rise = 0;
      for (int base = 0; base < depth; base += 4) {
        const int k = base + 1 + j;
        const bool has = k <= depth;
        const int a = (int)(((unsigned)(i + 1) >> (k & 31)) - 1u);
        // (the read is unconditional - of the root for a lane without an ancestor, or whose ancestor lies beyond LDS: that
        // entry was requested at the pop or is fetched behind one branch)
        u64 anc = q_lget(K, (has & (a < QL)) ? a : 0);
        if (base == 0 && i > 2 * QL) {
          if (has & (a >= QL)) { if (a == pa) anc = pv; else anc = q_gload_now(K.gq + (a - QL)); }
        }
        const bool up = has & (nf < hq_f(anc));
        const int um = quad_or(up ? (1 << j) : 0);
        const int cnt = __builtin_ctz(~(unsigned)um);      // the entry passes a PREFIX of its ancestors (heap order)
        const int dst = (int)(((unsigned)(i + 1) >> ((k - 1) & 31)) - 1u);
        q_hput_if(K, up, dst, anc);                        // ancestor k moves to where k - 1 was
        rise += cnt;
        if (cnt < 4) break;
      }
      const int fin = (int)(((unsigned)(i + 1) >> (rise & 31)) - 1u);
      q_hput_if(K, j == 0, fin, hq_pack(nf, nxy));
      // (this push wrote ancestors 0 .. rise of slot i: a lane's copy for the next push is stale if its slot is one of them -
      // on the copy's level that can only be ancestor 1 + j, or j where slot i + 1 opens a new level)
      {
        const int w_a = (int)(((unsigned)(i + 1) >> (1 + j)) - 1u), w_b = (int)(((unsigned)(i + 1) >> j) - 1u);
        if (((pa1 == w_a) & (1 + j <= rise)) | ((pa1 == w_b) & (j <= rise))) pa1 = -1;
      }
      s.hs = i + 1;
      npush++;
      wave_mem_sync();
    }
  } while (0);
  QP(5);
  // the entry the next pop moves to the root, if it lies beyond LDS: on its way while this turn ends and the next begins
  s.xpre = K.gq[max(s.hs - 1 - QL, 0)];
  QP(6);
  return st;
}

struct QReq { int start, goal, soft, cap; int32_t* out; };
struct QQueue { int32_t* l[4]; int n[4]; int32_t *retry_list, *fallback_list, *owned_list; int rank, world; };

__device__ __forceinline__ void quad_scratch_bind(const QSlots& qs, int slot, AScratch& S) {
  S.tab = nullptr; S.gq = nullptr; S.gd = nullptr; S.heap_cap = 0; S.epoch = 0;
  int32_t* c = qs.cells + (size_t)slot * ((size_t)5 * Q_CELLS + 3 * MAXB);
  S.A = c; S.P = c + Q_CELLS; S.T = c + 2 * Q_CELLS; S.PO = c + 3 * Q_CELLS; S.PD = c + 4 * Q_CELLS;
  S.BYP = c + 5 * Q_CELLS; S.OV = S.BYP + MAXB; S.DV = S.OV + MAXB;
  S.cap = Q_CELLS;
  S.use_reach = 0;
  S.calls = 0; S.expansions = 0; S.relaxations = 0;
  S.q_log = qs.log + (size_t)slot * (3 * QLOG);
  S.q_status = DV_BAIL; S.q_replay = 0; S.q_done = 0;
  S.q_start = 0; S.q_goal = 0; S.q_soft = 0; S.q_cap = 0; S.q_out = nullptr;
}

// One pass of vehicle i's step_decide by the quad (see the header: re-run from the top after every search).  Kept out of
// line like replan_turn.
__device__ __attribute__((noinline)) int quad_policy(const Dev& d, const TsParams& P, const QSlots& qs, const QQueue& q, int slot, int i,
                                                     int n_done, QReq& req) {
  AScratch S;
  quad_scratch_bind(qs, slot, S);
  S.q_done = n_done;
  const int r = decide_vehicle<DM_QUAD>(d, P, i, &S);
  const bool one = (threadIdx.x & 3) == 0;
  if (r == DV_SUSPEND) {
    req.start = S.q_start; req.goal = S.q_goal; req.soft = S.q_soft; req.cap = S.q_cap; req.out = S.q_out;
  } else if (one) {
    if (r == DV_DONE) {
      const int vid = d.active[i];
      if (S.calls > 0) d.tier_hint[vid] = (uint8_t)cost_bits(S.expansions);
      atomicAdd((unsigned long long*)&d.cnt->astar_calls, (unsigned long long)S.calls);
      atomicAdd((unsigned long long*)&d.cnt->astar_exp, (unsigned long long)S.expansions);
      atomicAdd((unsigned long long*)&d.cnt->astar_relax, (unsigned long long)S.relaxations);
      if (q.owned_list) q.owned_list[atomicAdd(&d.cnt->replan_n[6], 1)] = i;
    } else if (r == DV_POOL_FULL) q.retry_list[atomicAdd(&d.cnt->replan_n[4], 1)] = i;
    else { atomicAdd(&d.cnt->dbg[r == DV_BAIL ? 5 : 6], 1); __hip_atomic_store(&q.fallback_list[atomicAdd(&d.cnt->quad_n[0], 1)], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }     // DV_BAIL, DV_OVERFLOW: k_replan takes the vehicle
  }
  return r;
}

// Replanning vehicles, sixteen per wave: a work queue like k_replan's (class lists `class_mask` selects, most expensive
// first; cursor quad_n[1]).  Every quad takes entries until the queue is empty; vehicles this searcher cannot carry go
// to `fallback_list` (counter quad_n[0]) for k_replan.
__global__ void __launch_bounds__(64) k_replan_quad(Dev d, TsParams P, QSlots qs, RLists lists, int class_mask, int32_t* retry_list,
                                                    int32_t* fallback_list, int rank, int world, int32_t* owned_list) {
  const int lane = (int)threadIdx.x, j = lane & 3;
  const int slot = (int)blockIdx.x * 16 + (lane >> 2);
  const bool one = j == 0;
  QQueue q;
  for (int c = 0; c < 4; c++) { q.l[c] = lists.l[c]; q.n[c] = ((class_mask >> c) & 1) ? d.cnt->replan_n[c] : 0; }
  q.retry_list = retry_list; q.fallback_list = fallback_list; q.owned_list = owned_list; q.rank = rank; q.world = world;
  const int n3 = q.n[3], n2 = q.n[2], n1 = q.n[1], n0 = q.n[0];
  QConst K;
  K.tab = (TS_GLOBAL uint32_t*)(uintptr_t)(qs.tab + (size_t)slot * qs.tab_entries);
  K.gq = (gu64p)(uintptr_t)(qs.gq + (size_t)slot * Q_SPILL);
  K.gdw = (TS_GLOBAL uint32_t*)(uintptr_t)(qs.gdw + (size_t)slot * Q_DWORDS);
  K.lbase = (lane >> 2) * QL;
  K.j = j;
  K.dx = (j == 1) - (j == 3); K.dy = (j == 0) - (j == 2);
  K.amap = (const TS_GLOBAL u64*)(uintptr_t)uni64((u64)(uintptr_t)d.amap);
  K.W = uni(d.W); K.H = uni(d.H); K.W8 = uni(d.W8);
  K.tw = qs.tw; K.th = qs.th; K.tw8 = qs.tw >> 3;
  K.chk_x = qs.chk_x; K.chk_y = qs.chk_y;
  K.turn_on = P.turn_penalty_enabled != 0; K.rt_on = P.road_type_penalties_enabled != 0;
  K.turn2 = (int)((double)P.turn_penalty * 2.0); K.stop2 = (int)((double)P.obstacle_penalty_stop * 2.0);
  K.rt2_1 = (int)((double)P.road_type_penalty_r1 * 2.0); K.rt2_2 = (int)((double)P.road_type_penalty_r2 * 2.0);
  K.rt2_3 = (int)((double)P.road_type_penalty_r3 * 2.0);
  uint32_t epoch = qs.slot_epoch[slot];
  QState s;
  s.hs = 0; s.dir0 = -1; s.wb = 0; s.dwin = 0; s.xpre = 0; s.why = 0; s.gx = s.gy = s.sx = s.sy = 0; s.ox = s.oy = 0; s.goal_xy = 0; s.stamp = 0; s.soft = 0; s.n_exp = 0; s.n_relax = 0;
  QReq req;
  req.start = req.goal = req.soft = req.cap = 0; req.out = nullptr;
  int st = QS_NEEDJOB, job = -1, n_done = 0;
#ifdef TS_QUAD_PROF
  long long pf_t0 = clock64(), pf_search = 0, pf_turns = 0, pf_quadturns = 0;
  for (int k = 0; k < 8; k++) s.pf[k] = 0;
  s.pt = clock64();
#endif
  for (;;) {
    // ---------------- per quad: everything that is not a search turn ----------------
    while (st != QS_SEARCH && st != QS_IDLE) {
      if (st == QS_FOUND || st == QS_EMPTY) {
        // the search the policy waited for is over: its path (start excluded, goal included: 151-162) goes where the
        // policy asked for it, its length and counters into the log
        int len = 0;
        if (st == QS_FOUND) {
          const TS_GLOBAL uint32_t* tab = K.tab;
          gi32p outg = (gi32p)(uintptr_t)req.out;
          int px = (int)(s.goal_xy & 0xFFFFu), py = (int)(s.goal_xy >> 16);
          while (px != s.sx || py != s.sy) {
            if (len >= req.cap) { len = -1; break; }
            outg[req.cap - 1 - len] = py * K.W + px;
            len++;
            const int dd = (int)((tab[q_tix(K, s.ox, s.oy, px, py)] >> Q_DIR_SHIFT) & 3u);
            px -= (dd == 1) - (dd == 3); py -= (dd == 0) - (dd == 2);
          }
          wave_mem_sync();
          const int shift = req.cap - len;
          if (len > 0 && shift > 0)
            for (int k0 = 0; k0 < len; k0 += 4) {
              const int k = k0 + j;
              const int v = k < len ? outg[shift + k] : 0;
              if (k < len) outg[k] = v;
            }
          wave_mem_sync();
        }
        if (len < 0) { st = QS_ABANDON; s.why = 4; }
        else {
          int32_t* lg = qs.log + (size_t)slot * (3 * QLOG) + 3 * n_done;
          lg[0] = len; lg[1] = s.n_exp; lg[2] = s.n_relax;
          n_done++;
          st = QS_POLICY;
        }
      }
      if (st == QS_ABANDON) {
        if (one) atomicAdd(&d.cnt->dbg[s.why & 7], 1);
        if (one) __hip_atomic_store(&fallback_list[atomicAdd(&d.cnt->quad_n[0], 1)], job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st = QS_NEEDJOB;
      }
      if (st == QS_NEEDJOB) {
        int t = 0;
        if (one) t = atomicAdd(&d.cnt->quad_n[1], 1);
        t = quad_first(t);
        if (t >= n3 + n2 + n1 + n0) { st = QS_IDLE; break; }
        int i;
        if (t < n3) i = q.l[3][t];
        else if (t < n3 + n2) i = q.l[2][t - n3];
        else if (t < n3 + n2 + n1) i = q.l[1][t - n3 - n2];
        else i = q.l[0][t - n3 - n2 - n1];
        if (world > 1 && (t % world) != rank) continue;      // (position in the totally ordered queue, as in replan_turn)
        job = i; n_done = 0;
        st = QS_POLICY;
      }
      if (st == QS_POLICY) {
        const int r = quad_policy(d, P, qs, q, slot, job, n_done, req);
        if (r != DV_SUSPEND) { st = QS_NEEDJOB; continue; }
        // ---- a fresh search (113-128): new epoch for the table, start record, one-entry heap
        epoch++;
        if (epoch > 255u) {
          const size_t n = qs.tab_entries;
          for (size_t t = (size_t)j; t < n; t += 4) K.tab[t] = 0u;
          epoch = 1;
          wave_mem_sync();
        }
        int gxx, gyy, sxx, syy;
        cell_xy(d, req.goal, gxx, gyy); cell_xy(d, req.start, sxx, syy);
        s.gx = gxx; s.gy = gyy; s.sx = sxx; s.sy = syy;
        s.ox = K.chk_x ? sxx - (K.tw >> 1) : 0; s.oy = K.chk_y ? syy - (K.th >> 1) : 0;
        s.goal_xy = (uint32_t)gxx | ((uint32_t)gyy << 16);
        s.stamp = epoch << Q_STAMP_SHIFT;
        s.soft = req.soft;
        s.n_exp = 0; s.n_relax = 0;
        s.dir0 = -1; s.wb = 0; s.dwin = 0;
        if (one) {
          K.tab[q_tix(K, s.ox, s.oy, sxx, syy)] = s.stamp;    // dist 0
          q_lds[K.lbase] = hq_pack(abs(sxx - gxx) + abs(syy - gyy), (int)((uint32_t)sxx | ((uint32_t)syy << 16)));
        }
        s.hs = 1;
        wave_mem_sync();
        st = QS_SEARCH;
      }
    }
    const unsigned long long act = ballot(st == QS_SEARCH);
    if (act == 0ull) break;
    // ---------------- the searches of the wave advance in lockstep until one of them ends ----------------
#ifdef TS_QUAD_PROF
    const long long pf_a = clock64();
#endif
    do {
#ifdef TS_QUAD_PROF
      pf_turns++; pf_quadturns += __builtin_popcountll(act) >> 2;
#endif
      if (st == QS_SEARCH) st = quad_turn(K, s);
    } while (ballot(st == QS_SEARCH) == act);
#ifdef TS_QUAD_PROF
    pf_search += clock64() - pf_a;
#endif
  }
#ifdef TS_QUAD_PROF
  if (lane == 0) {   // wave cycles in all / in the lockstep loop, turns of the wave, quad-turns
    atomicAdd((unsigned long long*)&d.cnt->prof[0], (unsigned long long)(clock64() - pf_t0));
    atomicAdd((unsigned long long*)&d.cnt->prof[1], (unsigned long long)pf_search);
    atomicAdd((unsigned long long*)&d.cnt->prof[2], (unsigned long long)pf_turns);
    atomicAdd((unsigned long long*)&d.cnt->prof[3], (unsigned long long)pf_quadturns);
    for (int k = 0; k < 8; k++) atomicAdd((unsigned long long*)&d.cnt->qprof[k], (unsigned long long)s.pf[k]);
  }
#endif
  if (one) qs.slot_epoch[slot] = epoch;
  // every hand-back of this wave is published before the wave counts itself out (k_replan's replan_turn waits on both)
  __threadfence();
  if (lane == 0) atomicAdd(&d.cnt->quad_n[3], 1);
}

}  // namespace
