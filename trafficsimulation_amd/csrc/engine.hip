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
#include <unistd.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>   // host_shuffle.h: the AVX-512 draw extraction (picked at run time)
#endif

#include <hipcub/hipcub.hpp>   // the radix sort that orders a tick's replanning queue in space (run_replans)

#include "../../include/trafficsim.h"
#include "mt19937.h"

#include "dev.h"
#include "astar.h"
#include "astar_quad.h"
#include "kernels.h"
#include "host_state.h"
#include "host_shuffle.h"
#include "host_agents.h"

namespace {

// The replanning work queue (k_replan): replan_n[0..3] = class list lengths as k_decide_main left them (e->hint[8..]),
// lists 0..3 = the classes, list 4 = entries that found the path pool full.
inline int replan_pending(const int* n8) { return n8[0] + n8[1] + n8[2] + n8[3]; }

constexpr int SEG_VEHICLES = 1 << 20;   // vehicles per decide pass (bounds one pass' look-ahead into the MT19937 word ring)

int run_replans(E* e) {   // e->hint[8..15] = replan_n as k_decide_main left it
  Dev& d = e->d;
  const TsParams& P = e->P;
  hipStream_t st = e->stream;
  RLists rl;
  for (int q = 0; q < 6; q++) rl.l[q] = e->replan_list[q];
  int rc = ensure_slots(e);
  if (rc) return rc;
  if (!e->density_valid) { rc = ensure_density(e, d.occ_snap); if (rc) return rc; e->density_valid = true; }
  rc = ensure_amap(e);
  if (rc) return rc;
  // room in the path pool for what these replans will write (a planner that finds the pool full throws its searches
  // away and is run again): 128 words = 2048 path cells per entry, garbage-collecting / growing the pool if need be
  // (TS_DEBUG_POOL_PER_ENTRY shrinks the reservation so that tests can walk the pool-full retry path)
  const char* dbg_per = getenv("TS_DEBUG_POOL_PER_ENTRY");
  const size_t per_entry = dbg_per ? (size_t)atoi(dbg_per) : 128;
  rc = pool_make_room(e, (size_t)replan_pending(e->hint + 8) * per_entry + (dbg_per ? 64u : (1u << 20)));
  if (rc) return rc;
  if (dbg_per) d.pool_cap_words = std::min(e->pool_cap, e->pool_used + (size_t)replan_pending(e->hint + 8) * per_entry + 64u);
  if (dbg_per && getenv("TS_DEBUG_REPLAN")) fprintf(stderr, "[replan] pool used %zu cap %zu -> logical cap %llu\n", e->pool_used, e->pool_cap, d.pool_cap_words);
  // Order every class list by expected cost (largest first: the longest search of a tick bounds it) and, among equals, in
  // space (Morton order of 32 x 32-cell blocks of the vehicles' positions): the searches that run at the same time then
  // read the same few megabytes of the map snapshot, which stay in the XCDs' L2s instead of competing with thousands of
  // private tables for the MALL.  The order of the queue does not touch any result.
  static const bool spatial = !getenv("TS_NO_SPATIAL_QUEUE");
  const bool sharded = e->dist_world > 1;      // (then the order must be total and the same on every rank: 64-bit keys, every list)
  for (int h = 0; h < 4 && (spatial || sharded); h++) {
    const int n = e->hint[8 + h];
    if (n < (sharded ? 2 : 256)) continue;
    if ((size_t)n > e->cap_sortbuf) {
      const size_t nc = (size_t)n * 2;
      rc = regrow(e, &e->sort_keys, 0, nc * 2); if (rc) return rc;          // (room for 64-bit keys)
      rc = regrow(e, &e->sort_keys_alt, 0, nc * 2); if (rc) return rc;
      rc = regrow(e, &e->sort_vals_alt, 0, nc); if (rc) return rc;
      e->cap_sortbuf = nc;
    }
    if (sharded) {
      unsigned long long* k0 = (unsigned long long*)e->sort_keys;
      unsigned long long* k1 = (unsigned long long*)e->sort_keys_alt;
      hipLaunchKernelGGL(k_replan_keys64, dim3(nblk(n)), dim3(BLK), 0, st, d, e->replan_list[h], n, k0);
      size_t tmp_bytes = 0;
      HIPOK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, k0, k1, n, 0, 32 + REPLAN_KEY_BITS, st));
      if (tmp_bytes > e->cap_sorttmp) {
        rc = regrow(e, &e->sort_tmp, 0, tmp_bytes * 2); if (rc) return rc;
        e->cap_sorttmp = tmp_bytes * 2;
      }
      HIPOK(hipcub::DeviceRadixSort::SortKeys(e->sort_tmp, tmp_bytes, k0, k1, n, 0, 32 + REPLAN_KEY_BITS, st));
      hipLaunchKernelGGL(k_replan_unkey64, dim3(nblk(n)), dim3(BLK), 0, st, k1, n, e->replan_list[h]);
      continue;
    }
    hipLaunchKernelGGL(k_replan_keys, dim3(nblk(n)), dim3(BLK), 0, st, d, e->replan_list[h], n, e->sort_keys);
    size_t tmp_bytes = 0;
    HIPOK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, e->sort_keys, e->sort_keys_alt, e->replan_list[h], e->sort_vals_alt, n, 0, REPLAN_KEY_BITS, st));
    if (tmp_bytes > e->cap_sorttmp) {
      rc = regrow(e, &e->sort_tmp, 0, tmp_bytes * 2); if (rc) return rc;
      e->cap_sorttmp = tmp_bytes * 2;
    }
    HIPOK(hipcub::DeviceRadixSort::SortPairs(e->sort_tmp, tmp_bytes, e->sort_keys, e->sort_keys_alt, e->replan_list[h], e->sort_vals_alt, n, 0, REPLAN_KEY_BITS, st));
    HIPOK(hipMemcpyAsync(e->replan_list[h], e->sort_vals_alt, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  }
  // The quad searcher (astar_quad.h; TS_QUAD=0 switches it off): a big
  // queue (a replanning wave of TS_QUAD_MIN entries or more) sends the classes in TS_QUAD_CLASSES to k_replan_quad (sixteen
  // searches per wave, on its own stream) while k_replan runs beside it on the most expensive class and on every vehicle
  // the quads hand back as they work (searches that outgrow their window, heap or expansion budget, step-limited ones).
  // Smaller queues are bounded by their longest search, and that one is faster alone on a wave.
  // (a queue is the quads' when it is long against what k_replan can have in flight: 262 144 entries on the 5 751 slots a 4096^2
  // map leaves it, in proportion fewer where its node tables are bigger and its slots fewer - 1 534 at 8192^2)
  const int quad_min = getenv("TS_QUAD_MIN") ? atoi(getenv("TS_QUAD_MIN"))
                                             : (int)std::min<long long>(262144, std::max<long long>(16384, 46ll * e->slots.n_slots));
  bool split_done = false;      // (the queue is split between the ranks once; what is queued again - pool-full entries, hand-backs - is this rank's own)
  // (in the sharded multi-GPU mode the threshold applies to this rank's share: every world-th entry of the queue)
  const bool quad_queue = replan_pending(e->hint + 8) / std::max(e->dist_world, 1) >= std::max(quad_min, 1);
  // (set up with the first replans of a population that will fill such a queue - its first replanning wave - rather than inside that wave)
  const bool quad_soon = e->n_active / std::max(e->dist_world, 1) >= std::max(quad_min, 1);
  if (e->quad_on && (quad_queue || quad_soon)) { rc = ensure_qslots(e); if (rc) return rc; }
  if (e->quad_on && e->qslots_ready && quad_queue) {
    const int quad_mask = getenv("TS_QUAD_CLASSES") ? atoi(getenv("TS_QUAD_CLASSES")) & 15 : 7;
    int nq = 0, nw = 0;
    for (int c = 0; c < 4; c++) { if ((quad_mask >> c) & 1) nq += e->hint[8 + c]; else nw += e->hint[8 + c]; }
    const double tl = now_ms();
    split_done = true;
    rc = arena_to_quads(e);
    if (rc) return rc;
    HIPOK(hipMemsetAsync(d.cnt->quad_n, 0, sizeof(int) * 4, st));
    int tok = prof_begin(e, PK_REPLAN, nq + nw);
    int qgrid = 0;
    if (nq > 0) {
      HIPOK(hipMemsetAsync(e->replan_list[5], 0xFF, (size_t)nq * 4, st));     // (hand-back entries: -1 = not written yet)
      HIPOK(hipEventRecord(e->quad_ev0, st));
      HIPOK(hipStreamWaitEvent(e->quad_stream, e->quad_ev0, 0));
      qgrid = std::min((nq + 15) / 16, e->qslots.n_slots / 16);
      if (g_trace_launches) { fprintf(stderr, "[launch] k_replan_quad items=%d grid=%d\n", nq, qgrid); fflush(stderr); }
      hipLaunchKernelGGL(k_replan_quad, dim3(qgrid), dim3(64), 0, e->quad_stream, d, P, e->qslots, rl, quad_mask, e->replan_list[4],
                         e->replan_list[5], e->dist_rank, e->dist_world, e->dist_world > 1 ? e->owned_list : nullptr);
      HIPOK(hipEventRecord(e->quad_ev1, e->quad_stream));
    }
    // (k_replan never holds anything the quads wait for: were the two launches ever serialised, it would simply find the
    // hand-back list complete)
    // k_replan's waves beside the quads serve the most expensive class and the quads' hand-backs: 512 of them for a wave that hands back
    // thousands, 384 once the previous wave handed back few (they take issue slots from the quads: measured on the bench workload's four
    // waves, 5 678 / 2 895 / 1 379 / 2 514 hand-backs: 4.18 / 3.65 / 3.22 / 2.69 s with 512, 4.56 / 3.33 / 2.88 / 2.44 s with 384)
    const int side_waves = getenv("TS_QUAD_SIDE_WAVES") ? atoi(getenv("TS_QUAD_SIDE_WAVES"))
                                                        : (e->quad_last_fb < 0 || e->quad_last_fb > 4096 ? 512 : 384);
    const int wgrid = std::min(e->side_slots, std::max(std::min(nw, e->side_slots), nq > 0 ? side_waves : 1));
    if (nw > 0 || nq > 0)
      hipLaunchKernelGGL(k_replan, dim3(wgrid), dim3(64), 0, st, d, P, e->slots, rl, e->replan_list[4], e->dist_rank,
                         e->dist_world, e->dist_world > 1 ? e->owned_list : nullptr, 15 & ~quad_mask, e->replan_list[5], qgrid, nq);
    if (nq > 0) HIPOK(hipStreamWaitEvent(st, e->quad_ev1, 0));
    prof_end(e, tok);
    int qn[4] = {0, 0, 0, 0};
    HIPOK(hipMemcpyAsync(e->hint + 8, d.cnt->replan_n, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
    HIPOK(hipMemcpyAsync(e->hint + 3, &d.cnt->error, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPOK(hipMemcpyAsync(qn, d.cnt->quad_n, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    if (g_trace_launches) { fprintf(stderr, "[done] replanning pass with the quads\n"); fflush(stderr); }
    if (e->hint[3] == TS_E_CAPACITY) return fail(e, TS_E_CAPACITY, "an A* search exceeded its heap or path buffers");
    const int fb = qn[0], retry = e->hint[8 + 4];
    e->quad_jobs += nq; e->quad_fallbacks += fb;
    e->quad_last_fb = fb;
#ifdef TS_QUAD_PROF
    {
      long long pf[8];
      HIPOK(hipMemcpy(pf, d.cnt->prof, sizeof(pf), hipMemcpyDeviceToHost));
      fprintf(stderr, "[quadprof] wave cycles %lld, in the lockstep loop %lld, wave turns %lld, quad turns %lld: %.0f cycles per turn, %.1f quads per turn\n",
              pf[0], pf[1], pf[2], pf[3], pf[2] ? (double)pf[1] / (double)pf[2] : 0.0, pf[2] ? (double)pf[3] / (double)pf[2] : 0.0);
      HIPOK(hipMemset(d.cnt->prof, 0, sizeof(pf)));
      long long qp[8];
      HIPOK(hipMemcpy(qp, d.cnt->qprof, sizeof(qp), hipMemcpyDeviceToHost));
      fprintf(stderr, "[quadprof] per wave turn: pop+loads %.0f, sift LDS %.0f, sift deep %.0f, goal/stale %.0f, eval %.0f, pushes(+skipped) %.0f, tail %.0f, between turns %.0f\n",
              (double)qp[0] / pf[2], (double)qp[1] / pf[2], (double)qp[2] / pf[2], (double)qp[3] / pf[2], (double)qp[4] / pf[2], (double)qp[5] / pf[2], (double)qp[6] / pf[2], (double)qp[7] / pf[2]);
      HIPOK(hipMemset(d.cnt->qprof, 0, sizeof(qp)));
    }
#endif
    if (getenv("TS_DEBUG_REPLAN")) {
      int dbg[8];
      HIPOK(hipMemcpy(dbg, d.cnt->dbg, sizeof(dbg), hipMemcpyDeviceToHost));
      fprintf(stderr, "[replan] hand-backs so far by reason: window / g %d, heap %d, expansion budget %d, path buffer %d, policy (step-limited / contraflow search) %d, other %d\n", dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6]);
    }
    if (getenv("TS_DEBUG_REPLAN"))
      fprintf(stderr, "[replan] tick %lld: %d entries to the quads (%d waves of %d), %d to k_replan (%d waves); handed over %d, pool-full %d, %.2f ms\n",
              (long long)e->C.step_count, nq, qgrid, e->qslots.n_slots / 16, nw, wgrid, fb, retry, now_ms() - tl);
    // what k_replan did not get to serve of the hand-backs (it only gives up on them when the two kernels were not run side
    // by side) and what found the path pool full is queued again below
    const int served = std::min(qn[2], fb), left = fb - served;
    if (retry > 0) {
      d.pool_cap_words = e->pool_cap;
      rc = pool_make_room(e, (size_t)retry * 1024 + (1u << 20));
      if (rc) return rc;
    }
    if (left > 0) HIPOK(hipMemcpyAsync(e->replan_list[0], e->replan_list[5] + served, (size_t)left * 4, hipMemcpyDeviceToDevice, st));
    if (retry > 0) HIPOK(hipMemcpyAsync(e->replan_list[0] + left, e->replan_list[4], (size_t)retry * 4, hipMemcpyDeviceToDevice, st));
    const int keep_owned = e->hint[8 + 6];
    for (int q = 0; q < 8; q++) e->hint[8 + q] = 0;
    e->hint[8] = left + retry; e->hint[8 + 6] = keep_owned;
    HIPOK(hipMemcpyAsync(d.cnt->replan_n, e->hint + 8, sizeof(int) * 8, hipMemcpyHostToDevice, st));
  }
  while (replan_pending(e->hint + 8) > 0) {
    const int n = replan_pending(e->hint + 8);
    const int grid = std::min(n, e->slots.n_slots);
    if (grid > e->side_slots && e->quad_on) { rc = arena_to_waves(e); if (rc) return rc; }
    const int w_rank = split_done ? 0 : e->dist_rank, w_world = split_done ? 1 : e->dist_world;
    split_done = true;
#ifdef TS_TRACE_REPLAN
    static int4* tr_buf = nullptr;
    const int tr_cap = 1 << 22;
    if (!tr_buf) {
      HIPOK(hipMalloc(&tr_buf, (size_t)tr_cap * sizeof(int4)));
      HIPOK(hipMemcpyToSymbol(HIP_SYMBOL(g_rtrace), &tr_buf, sizeof(tr_buf)));
      HIPOK(hipMemcpyToSymbol(HIP_SYMBOL(g_rtrace_cap), &tr_cap, sizeof(tr_cap)));
    }
    HIPOK(hipMemsetAsync(tr_buf, 0, (size_t)std::min(n, tr_cap) * sizeof(int4), st));
#endif
    LAUNCH(e, PK_REPLAN, n, k_replan, dim3(grid), dim3(64), d, P, e->slots, rl, e->replan_list[4], w_rank, w_world,
           e->dist_world > 1 ? e->owned_list : nullptr, 15, (int32_t*)nullptr, 0, 0);
    const double tl = now_ms();
    HIPOK(hipMemcpyAsync(e->hint + 8, d.cnt->replan_n, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
    HIPOK(hipMemcpyAsync(e->hint + 3, &d.cnt->error, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
#ifdef TS_TRACE_REPLAN
    if (n >= 1000) {
      std::vector<int4> h((size_t)std::min(n, tr_cap));
      HIPOK(hipMemcpy(h.data(), tr_buf, h.size() * sizeof(int4), hipMemcpyDeviceToHost));
      char name[128];
      snprintf(name, sizeof(name), "gpurun_out/rtrace_tick%lld.bin", (long long)e->C.step_count);
      if (FILE* f = fopen(name, "wb")) { fwrite(h.data(), sizeof(int4), h.size(), f); fclose(f); }
      fprintf(stderr, "[rtrace] tick %lld: %d entries, %d searchers, %.2f ms\n", (long long)e->C.step_count, n, grid, now_ms() - tl);
    }
#endif
    if (getenv("TS_DEBUG_REPLAN")) {
      int dbg[8];
      HIPOK(hipMemcpy(dbg, d.cnt->dbg, sizeof(dbg), hipMemcpyDeviceToHost));
      fprintf(stderr, "[replan] deepest heap so far %d, longest search so far %d expansions, %lld expansions so far with part of the heap in HBM\n",
              dbg[4], dbg[5], (long long)(((unsigned long long)(unsigned)dbg[7] << 32) | (unsigned)dbg[6]));
    }
    if (e->hint[3] == TS_E_CAPACITY) return fail(e, TS_E_CAPACITY, "an A* search exceeded its heap or path buffers");
    const int retry = e->hint[8 + 4];
    if (getenv("TS_DEBUG_REPLAN"))
      fprintf(stderr, "[replan] tick %lld: %d entries (classes %d/%d/%d/%d) on %d searchers, retry=%d, %.2f ms\n", (long long)e->C.step_count,
              n, e->hint[8], e->hint[9], e->hint[10], e->hint[11], grid, retry, now_ms() - tl);
    if (retry == 0) break;
    // the path pool filled up: make room (GC, then growth) and run the entries that could not commit again
    d.pool_cap_words = e->pool_cap;
    rc = pool_make_room(e, (size_t)retry * 1024 + (1u << 20));
    if (rc) return rc;
    HIPOK(hipMemcpyAsync(e->replan_list[0], e->replan_list[4], (size_t)retry * 4, hipMemcpyDeviceToDevice, st));
    const int keep_owned = e->hint[8 + 6];
    for (int q = 0; q < 8; q++) e->hint[8 + q] = 0;
    e->hint[8] = retry; e->hint[8 + 6] = keep_owned;
    HIPOK(hipMemcpyAsync(d.cnt->replan_n, e->hint + 8, sizeof(int) * 8, hipMemcpyHostToDevice, st));
  }
  d.pool_cap_words = e->pool_cap;
  return TS_OK;
}

// Replicated-state multi-GPU mode: after this rank's share of the replans, trade results with the other ranks.
// `before` = the device counters as they were when the replanning phase began.
struct XHeader { int64_t n_recs, n_words, n_arr, error; int64_t delta[16]; double ddelta[2]; };
// `local_rc` != 0: this rank failed on the host side before the exchange (capacity, a device error): it still joins the
// collective, with a header that says so, and every rank fails the tick together instead of waiting in the all-gather
// until the process group times out.
int exchange_replans(E* e, const DevCnt& before, int local_rc) {
  Dev& d = e->d;
  hipStream_t st = e->stream;
  const double t0 = now_ms();
  if (local_rc) {
    XHeader hd;
    memset(&hd, 0, sizeof(hd));
    hd.delta[14] = local_rc;
    e->send_buf.resize(sizeof(hd));
    memcpy(e->send_buf.data(), &hd, sizeof(hd));
    void* recv = nullptr;
    int64_t* sizes = nullptr;
    int64_t stride = 0;
    const void* sendp = e->send_buf.data();
    if (e->dist_dev) {
      if (e->cap_send < sizeof(hd) && regrow(e, &e->d_send, 0, (size_t)4096) == TS_OK) e->cap_send = 4096;
      if (e->cap_send >= sizeof(hd) && hipMemcpy(e->d_send, &hd, sizeof(hd), hipMemcpyHostToDevice) == hipSuccess) sendp = e->d_send;
      else return local_rc;     // (no device memory left even for a header: the other ranks run into their collective's time-out)
    }
    (void)e->dist_fn(e->dist_user, sendp, (int64_t)sizeof(hd), &recv, &sizes, &stride);
    return local_rc;
  }
  const int n_owned = e->hint[8 + 6];
  if ((size_t)std::max(n_owned, 1) > e->cap_recs) {
    const size_t nc = (size_t)n_owned * 2 + 1024;
    int rc = regrow(e, &e->d_recs, 0, nc); if (rc) return rc;
    e->cap_recs = nc;
  }
  if (!e->d_xwords_n) HIPOK(dalloc(e, &e->d_xwords_n, 1));
  // the words this rank's replans rewrote, counted first (the pool's growth over the phase is no bound: pool_make_room may
  // have garbage-collected inside it), then exported into a buffer that holds them
  HIPOK(hipMemcpyAsync(e->hcnt, d.cnt, sizeof(DevCnt), hipMemcpyDeviceToHost, st));
  HIPOK(hipMemsetAsync(e->d_xwords_n, 0, sizeof(unsigned long long), st));
  if (n_owned > 0)
    hipLaunchKernelGGL(k_replan_export, dim3(nblk(n_owned)), dim3(BLK), 0, st, d, e->owned_list, n_owned, e->d_recs, e->d_xwords, e->d_xwords_n, 1);
  unsigned long long need_words = 0;
  HIPOK(hipMemcpyAsync(&need_words, e->d_xwords_n, sizeof(need_words), hipMemcpyDeviceToHost, st));
  HIPOK(hipStreamSynchronize(st));
  const DevCnt after = *e->hcnt;
  if ((size_t)need_words + 16 > e->cap_xwords) {
    const size_t nc = (size_t)need_words * 2 + 4096;
    int rc = regrow(e, &e->d_xwords, 0, nc); if (rc) return rc;
    e->cap_xwords = nc;
  }
  HIPOK(hipMemsetAsync(e->d_xwords_n, 0, sizeof(unsigned long long), st));
  if (n_owned > 0)
    hipLaunchKernelGGL(k_replan_export, dim3(nblk(n_owned)), dim3(BLK), 0, st, d, e->owned_list, n_owned, e->d_recs, e->d_xwords, e->d_xwords_n, 0);
  unsigned long long n_words = 0;
  HIPOK(hipMemcpyAsync(&n_words, e->d_xwords_n, sizeof(n_words), hipMemcpyDeviceToHost, st));
  HIPOK(hipStreamSynchronize(st));
  if (n_words != need_words) return fail(e, TS_E_DEVICE, "replan export wrote a different number of words than it counted (internal error)");
  const int n_arr = after.arr_n - before.arr_n;
  XHeader hd;
  memset(&hd, 0, sizeof(hd));
  hd.n_recs = n_owned; hd.n_words = (int64_t)n_words; hd.n_arr = n_arr; hd.error = after.error;
  const long long dl[] = {after.stuck - before.stuck, after.collisions - before.collisions, after.malfunctions - before.malfunctions,
                          after.overtaking - before.overtaking, after.in_stuck_detour - before.in_stuck_detour, after.parked - before.parked,
                          after.completed_internal - before.completed_internal, after.completed_through - before.completed_through,
                          after.dist_internal - before.dist_internal, after.dist_through - before.dist_through,
                          after.astar_calls - before.astar_calls, after.astar_exp - before.astar_exp, after.astar_relax - before.astar_relax};
  for (size_t q = 0; q < sizeof(dl) / sizeof(dl[0]); q++) hd.delta[q] = dl[q];
  hd.delta[13] = after.dec_arrived;
  hd.ddelta[0] = after.dur_internal - before.dur_internal; hd.ddelta[1] = after.dur_through - before.dur_through;
  const size_t bytes = sizeof(XHeader) + (size_t)n_owned * sizeof(ReplanRec) + (size_t)n_words * 4 + (size_t)std::max(n_arr, 0) * 12;
  if (e->dist_dev) {
    // ---- device-direct exchange: header | records | words | service records packed in device memory, gathered by the
    // callback between device buffers, imported straight out of the gathered slots
    if (bytes > e->cap_send) {
      const size_t nc = bytes * 2 + 4096;
      int rc = regrow(e, &e->d_send, 0, nc); if (rc) return rc;
      e->cap_send = nc;
    }
    uint8_t* p = e->d_send;
    HIPOK(hipMemcpyAsync(p, &hd, sizeof(hd), hipMemcpyHostToDevice, st)); p += sizeof(hd);
    if (n_owned > 0) HIPOK(hipMemcpyAsync(p, e->d_recs, (size_t)n_owned * sizeof(ReplanRec), hipMemcpyDeviceToDevice, st));
    p += (size_t)n_owned * sizeof(ReplanRec);
    if (n_words > 0) HIPOK(hipMemcpyAsync(p, e->d_xwords, (size_t)n_words * 4, hipMemcpyDeviceToDevice, st));
    p += (size_t)n_words * 4;
    if (n_arr > 0) HIPOK(hipMemcpyAsync(p, d.arr + 3 * (size_t)before.arr_n, (size_t)n_arr * 12, hipMemcpyDeviceToDevice, st));
    HIPOK(hipStreamSynchronize(st));
    void* recv = nullptr;
    int64_t* sizes = nullptr;
    int64_t stride = 0;
    const int xrc = e->dist_fn(e->dist_user, e->d_send, (int64_t)bytes, &recv, &sizes, &stride);
    if (xrc != 0 || !recv || !sizes) return fail(e, TS_E_DEVICE, "the replan exchange callback failed");
    std::vector<XHeader> hs((size_t)e->dist_world);
    for (int r = 0; r < e->dist_world; r++) {
      if (r == e->dist_rank) continue;
      if ((size_t)sizes[r] < sizeof(XHeader)) return fail(e, TS_E_DEVICE, "short replan exchange buffer");
      HIPOK(hipMemcpyAsync(&hs[r], (const uint8_t*)recv + (size_t)r * (size_t)stride, sizeof(XHeader), hipMemcpyDeviceToHost, st));
    }
    HIPOK(hipStreamSynchronize(st));
    DevCnt merged = after;
    long long in_words = 0, in_arr = 0;
    for (int r = 0; r < e->dist_world; r++) {
      if (r == e->dist_rank) continue;
      const XHeader& h2 = hs[r];
      if (h2.delta[14]) return fail(e, (int)h2.delta[14], "rank " + std::to_string(r) + " failed in its share of the replans (error " + std::to_string((long long)h2.delta[14]) + ")");
      in_words += h2.n_words; in_arr += h2.n_arr;
      if ((size_t)sizes[r] != sizeof(XHeader) + (size_t)h2.n_recs * sizeof(ReplanRec) + (size_t)h2.n_words * 4 + (size_t)h2.n_arr * 12)
        return fail(e, TS_E_DEVICE, "replan exchange buffer size mismatch");
    }
    if (in_words > 0) { int rc = pool_make_room(e, (size_t)in_words + 64); if (rc) return rc; }
    if (in_arr > 0 && (long long)after.arr_n + in_arr > d.arr_cap) return fail(e, TS_E_CAPACITY, "more service records in one tick than the record buffer holds");
    e->exchange_bytes += (long long)bytes;
    int arr_at = after.arr_n;
    for (int r = 0; r < e->dist_world; r++) {
      if (r == e->dist_rank) continue;
      const XHeader& h2 = hs[r];
      const uint8_t* q = (const uint8_t*)recv + (size_t)r * (size_t)stride + sizeof(XHeader);
      merged.stuck += h2.delta[0]; merged.collisions += h2.delta[1]; merged.malfunctions += h2.delta[2]; merged.overtaking += h2.delta[3];
      merged.in_stuck_detour += h2.delta[4]; merged.parked += h2.delta[5]; merged.completed_internal += h2.delta[6];
      merged.completed_through += h2.delta[7]; merged.dist_internal += h2.delta[8]; merged.dist_through += h2.delta[9];
      merged.astar_calls += h2.delta[10]; merged.astar_exp += h2.delta[11]; merged.astar_relax += h2.delta[12];
      merged.dur_internal += h2.ddelta[0]; merged.dur_through += h2.ddelta[1];
      if (h2.error && !merged.error) merged.error = (int)h2.error;
      if ((int)h2.delta[13] > merged.dec_arrived) merged.dec_arrived = (int)h2.delta[13];
      if (h2.n_recs > 0)
        hipLaunchKernelGGL(k_replan_import, dim3(nblk((long long)h2.n_recs)), dim3(BLK), 0, st, d, (const ReplanRec*)q, (int)h2.n_recs,
                           (const uint32_t*)(q + (size_t)h2.n_recs * sizeof(ReplanRec)));
      if (h2.n_arr > 0) {
        HIPOK(hipMemcpyAsync(d.arr + 3 * (size_t)arr_at, q + (size_t)h2.n_recs * sizeof(ReplanRec) + (size_t)h2.n_words * 4, (size_t)h2.n_arr * 12, hipMemcpyDeviceToDevice, st));
        arr_at += (int)h2.n_arr;
      }
    }
    unsigned long long pool_now = 0;
    HIPOK(hipMemcpyAsync(&pool_now, &d.cnt->pool_used, sizeof(pool_now), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));     // (the gathered slots are the callee's again after this tick's imports)
    merged.pool_used = pool_now; merged.arr_n = arr_at;
    *e->hcnt = merged;
    HIPOK(hipMemcpyAsync(d.cnt, e->hcnt, sizeof(DevCnt), hipMemcpyHostToDevice, st));
    HIPOK(hipStreamSynchronize(st));
    e->exchange_ms += now_ms() - t0;
    return TS_OK;
  }
  e->send_buf.resize(bytes);
  uint8_t* p = e->send_buf.data();
  memcpy(p, &hd, sizeof(hd)); p += sizeof(hd);
  if (n_owned > 0) HIPOK(hipMemcpyAsync(p, e->d_recs, (size_t)n_owned * sizeof(ReplanRec), hipMemcpyDeviceToHost, st));
  p += (size_t)n_owned * sizeof(ReplanRec);
  if (n_words > 0) HIPOK(hipMemcpyAsync(p, e->d_xwords, (size_t)n_words * 4, hipMemcpyDeviceToHost, st));
  p += (size_t)n_words * 4;
  if (n_arr > 0) HIPOK(hipMemcpyAsync(p, d.arr + 3 * (size_t)before.arr_n, (size_t)n_arr * 12, hipMemcpyDeviceToHost, st));
  HIPOK(hipStreamSynchronize(st));
  void* recv = nullptr;
  int64_t* sizes = nullptr;
  int64_t stride = 0;
  const int xrc = e->dist_fn(e->dist_user, e->send_buf.data(), (int64_t)bytes, &recv, &sizes, &stride);
  if (xrc != 0 || !recv || !sizes) return fail(e, TS_E_DEVICE, "the replan exchange callback failed");
  // apply the other ranks' results; counters through the host copy
  DevCnt merged = after;
  long long in_words = 0, in_arr = 0;
  for (int r = 0; r < e->dist_world; r++) {
    if (r == e->dist_rank) continue;
    const uint8_t* q = (const uint8_t*)recv + (size_t)r * (size_t)stride;
    if ((size_t)sizes[r] < sizeof(XHeader)) return fail(e, TS_E_DEVICE, "short replan exchange buffer");
    XHeader h2;
    memcpy(&h2, q, sizeof(h2));
    if (h2.delta[14]) return fail(e, (int)h2.delta[14], "rank " + std::to_string(r) + " failed in its share of the replans (error " + std::to_string((long long)h2.delta[14]) + ")");
    in_words += h2.n_words; in_arr += h2.n_arr;
    if ((size_t)sizes[r] != sizeof(XHeader) + (size_t)h2.n_recs * sizeof(ReplanRec) + (size_t)h2.n_words * 4 + (size_t)h2.n_arr * 12)
      return fail(e, TS_E_DEVICE, "replan exchange buffer size mismatch");
  }
  if (in_words > 0) { int rc = pool_make_room(e, (size_t)in_words + 64); if (rc) return rc; }
  if (in_arr > 0 && (long long)after.arr_n + in_arr > d.arr_cap) return fail(e, TS_E_CAPACITY, "more service records in one tick than the record buffer holds");
  e->exchange_bytes += (long long)bytes;
  int arr_at = after.arr_n;
  for (int r = 0; r < e->dist_world; r++) {
    if (r == e->dist_rank) continue;
    const uint8_t* q = (const uint8_t*)recv + (size_t)r * (size_t)stride;
    XHeader h2;
    memcpy(&h2, q, sizeof(h2)); q += sizeof(h2);
    merged.stuck += h2.delta[0]; merged.collisions += h2.delta[1]; merged.malfunctions += h2.delta[2]; merged.overtaking += h2.delta[3];
    merged.in_stuck_detour += h2.delta[4]; merged.parked += h2.delta[5]; merged.completed_internal += h2.delta[6];
    merged.completed_through += h2.delta[7]; merged.dist_internal += h2.delta[8]; merged.dist_through += h2.delta[9];
    merged.astar_calls += h2.delta[10]; merged.astar_exp += h2.delta[11]; merged.astar_relax += h2.delta[12];
    merged.dur_internal += h2.ddelta[0]; merged.dur_through += h2.ddelta[1];
    if (h2.error && !merged.error) merged.error = (int)h2.error;
    if ((int)h2.delta[13] > merged.dec_arrived) merged.dec_arrived = (int)h2.delta[13];
    if (h2.n_recs > 0) {
      if ((size_t)h2.n_recs > e->cap_recs) { const size_t nc = (size_t)h2.n_recs * 2; int rc = regrow(e, &e->d_recs, 0, nc); if (rc) return rc; e->cap_recs = nc; }
      if ((size_t)h2.n_words + 16 > e->cap_xwords) { const size_t nc = (size_t)h2.n_words * 2 + 4096; int rc = regrow(e, &e->d_xwords, 0, nc); if (rc) return rc; e->cap_xwords = nc; }
      HIPOK(hipMemcpyAsync(e->d_recs, q, (size_t)h2.n_recs * sizeof(ReplanRec), hipMemcpyHostToDevice, st));
      if (h2.n_words > 0) HIPOK(hipMemcpyAsync(e->d_xwords, q + (size_t)h2.n_recs * sizeof(ReplanRec), (size_t)h2.n_words * 4, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_replan_import, dim3(nblk((long long)h2.n_recs)), dim3(BLK), 0, st, d, e->d_recs, (int)h2.n_recs, e->d_xwords);
      HIPOK(hipStreamSynchronize(st));   // (the staging buffers are reused for the next rank)
    }
    if (h2.n_arr > 0) {
      HIPOK(hipMemcpyAsync(d.arr + 3 * (size_t)arr_at, q + (size_t)h2.n_recs * sizeof(ReplanRec) + (size_t)h2.n_words * 4, (size_t)h2.n_arr * 12, hipMemcpyHostToDevice, st));
      arr_at += (int)h2.n_arr;
    }
  }
  // counters: the device copy moved on only in pool_used (imports); everything else is after + the others' deltas
  unsigned long long pool_now = 0;
  HIPOK(hipMemcpyAsync(&pool_now, &d.cnt->pool_used, sizeof(pool_now), hipMemcpyDeviceToHost, st));
  HIPOK(hipStreamSynchronize(st));
  merged.pool_used = pool_now; merged.arr_n = arr_at;
  *e->hcnt = merged;
  HIPOK(hipMemcpyAsync(d.cnt, e->hcnt, sizeof(DevCnt), hipMemcpyHostToDevice, st));
  HIPOK(hipStreamSynchronize(st));
  e->exchange_ms += now_ms() - t0;
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


int tick(E* e) {
  Dev& d = e->d;
  const TsParams& P = e->P;
  hipStream_t st = e->stream;
  int nA = e->n_active, nS = e->n_sched;
  // PATHFINDING_BATCHING=False (vehicle_base.py:669-670): no decide phase - every vehicle runs step_decide at the top of its own
  // step(), in the shuffled order.  Here: the move phase below stops in front of every vehicle (a "point", like the traffic
  // generator's), runs the decide machinery for that one vehicle on the maps and the stream as they are, and goes on.  One
  // vehicle at a time is the reference's own speed limit for this switch; nothing about it is parallel.
  const bool seq = !P.pathfinding_batching;
  d.seq = seq ? 1 : 0;
  if (seq && e->dist_world > 1) return fail(e, TS_E_UNSUPPORTED, "PATHFINDING_BATCHING=False has no sharded form (its decisions are sequential)");
  // Vehicles that stand on their target (a trip that ends where it starts) despawn inside the decide phase
  // (vehicle_base.py:657-661) - and leave the schedule before it is shuffled.  `standing` = their decide indices.
  std::vector<int32_t> standing;
  if (e->standing_possible && nA > 0 && !seq) {
    if (!e->d_standing) HIPOK(dalloc(e, &e->d_standing, (size_t)E::STANDING_CAP + 1));
    HIPOK(hipMemsetAsync(e->d_standing, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_find_standing, dim3(nblk(nA)), dim3(BLK), 0, st, d, nA, e->d_standing, (int)E::STANDING_CAP);
    int32_t n_st = 0;
    HIPOK(hipMemcpyAsync(&n_st, e->d_standing, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    if (n_st > E::STANDING_CAP) return fail(e, TS_E_CAPACITY, "more than 4096 vehicles stand on their own target");
    standing.resize((size_t)n_st);
    if (n_st > 0) HIPOK(hipMemcpy(standing.data(), e->d_standing + 1, (size_t)n_st * 4, hipMemcpyDeviceToHost));
    std::sort(standing.begin(), standing.end());
    if (n_st == 0) e->standing_possible = false;
  }
  const bool careful = !standing.empty();
  // the scheduler stream is independent of everything the decide phase does: shuffle on a host thread (unless the decide
  // phase may still change the schedule)
  struct Joiner { E* e; bool done = false; ~Joiner() { if (!done) shuffle_wait(e); } } joiner{e};
  if (!careful) shuffle_start(e, nS); else joiner.done = true;
  const double t_tick0 = now_ms();

  // density_map is a function of the occupancy at this point (city_model.py:1853)
  HIPOK(hipMemcpyAsync(d.occ_snap, d.occ, (size_t)e->N, hipMemcpyDeviceToDevice, st));
  e->density_valid = false;
  e->amap_valid = false;   // the A* snapshot is rebuilt from the cell records when the first search of the tick needs it
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
    // One stretch [lo, hi) of the decide order, start to finish: draws, step_decide, the searches it asks for.  A tick
    // is one stretch unless a vehicle may despawn inside the decide phase (see below).
    auto decide_range = [&](const int lo, const int hi) -> int {
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
    static const int SEG = getenv("TS_DEBUG_SEG") ? std::max(64, atoi(getenv("TS_DEBUG_SEG"))) : SEG_VEHICLES;
    int start = lo;
    bool main_done = false;
    LAUNCH(e, PK_DECIDE_PRE, hi - lo, k_decide_pre, dim3(nblk(hi - lo)), dim3(BLK), d, P, lo, hi);
    while (start < hi) {
      const int seg_end = std::min(hi, start + SEG), n = seg_end - start;
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
          if (e->take_ev_recorded) HIPOK(hipEventSynchronize(e->take_ev));   // the table built ahead still owns d_take / h_take
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
        if (seg_end == hi) {  // k_decide_main returns at once if a draw fired (the fix-up below re-runs it)
          LAUNCH(e, PK_DECIDE_MAIN, hi - lo, k_decide_main, dim3(nblk(hi - lo)), dim3(BLK), d, P, lo, hi, rlists);
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
          main_done = seg_end == hi;
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
        if (start < hi) LAUNCH(e, PK_DECIDE_PRE, hi - start, k_decide_pre, dim3(nblk(hi - start)), dim3(BLK), d, P, start, hi);
      }
    }
    if (!main_done) {  // the last pass ended with an event at the very last vehicle (or there was no pass left)
      HIPOK(hipMemsetAsync(&d.cnt->rng_event, 0xFF, sizeof(unsigned int), st));
      LAUNCH(e, PK_DECIDE_MAIN, hi - lo, k_decide_main, dim3(nblk(hi - lo)), dim3(BLK), d, P, lo, hi, rlists);
      HIPOK(hipMemcpyAsync(e->hint + 8, d.cnt->replan_n, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
      HIPOK(hipStreamSynchronize(st));
    }
    if (hi == nA) {
    // prefetch the part of the stream the next tick will most likely read
    // (at most one pass' worth - the ring holds MAX_AHEAD_BLOCKS blocks; populations above SEG vehicles decide in passes)
    { const double t_wu = now_ms();
      const uint64_t ahead = std::min<uint64_t>((uint64_t)std::min(nA, SEG_VEHICLES) * 4 + (1u << 16), MTPipe::MAX_AHEAD_BLOCKS * 600ull - (1u << 16));
      int rc = words_upload(e, r.pos() + ahead); if (rc) return rc;
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
        e->take_ev_recorded = true;
        e->take_base = nb; e->take_n = nt;
      }
    }
    }
    if (e->dist_world > 1) {
      // every rank sees the same work lists (as sets): plan this rank's share, then trade results - also when this
      // rank has nothing to plan, the exchange is collective
      HIPOK(hipMemcpyAsync(e->hcnt, d.cnt, sizeof(DevCnt), hipMemcpyDeviceToHost, st));
      HIPOK(hipStreamSynchronize(st));
      const DevCnt before = *e->hcnt;
      const int n_all = replan_pending(e->hint + 8);
      int rc_local = TS_OK;
      if (n_all > e->cap_owned) { const int nc = n_all * 2 + 1024; rc_local = regrow(e, &e->owned_list, 0, (size_t)nc); if (!rc_local) e->cap_owned = nc; }
      if (!rc_local && n_all > 0) rc_local = run_replans(e);
      int rc = exchange_replans(e, before, rc_local);
      if (rc) return rc;
    } else if (replan_pending(e->hint + 8) > 0) { int rc = run_replans(e); if (rc) return rc; }
    return TS_OK;
    };
  if (nA > 0 && seq) HIPOK(hipMemsetAsync(d.ev, 0, (size_t)e->n_vehicles_total, st));
  if (nA > 0 && !seq) {
    HIPOK(hipMemsetAsync(d.ev, 0, (size_t)e->n_vehicles_total, st));
    if (!careful) {
      int rc = decide_range(0, nA);
      if (rc) return rc;
    } else {
      // Stretch by stretch, each ending at a vehicle that may despawn: everybody up to and including it decides (draws,
      // searches and all), then it leaves the maps and the vehicle behind it loses its turn (k_decide_despawn) - the
      // later stretches see exactly what the reference's sequential loop would show them.
      int lo = 0, removed = 0;
      size_t ci = 0;
      while (lo < nA) {
        while (ci < standing.size() && standing[ci] < lo) ci++;   // (it was the one that lost its turn)
        const bool at_candidate = ci < standing.size();
        const int hi = at_candidate ? standing[ci] + 1 : nA;
        d.dec_expect = at_candidate ? hi : 0;
        if (at_candidate) HIPOK(hipMemsetAsync(&d.cnt->dec_arrived, 0, sizeof(int), st));
        int rc = decide_range(lo, hi);
        d.dec_expect = 0;
        if (rc) return rc;
        lo = hi;
        if (at_candidate) {
          int arrived = 0;
          HIPOK(hipMemcpyAsync(&arrived, &d.cnt->dec_arrived, sizeof(int), hipMemcpyDeviceToHost, st));
          HIPOK(hipStreamSynchronize(st));
          if (arrived == hi) {
            hipLaunchKernelGGL(k_decide_despawn, dim3(1), dim3(64), 0, st, d, P, hi - 1, hi < nA ? hi : -1);
            e->amap_valid = false;
            removed++;
            lo = hi + 1;
          }
          ci++;
        }
      }
      if (removed > 0) {   // the schedule is shuffled without them (RandomActivation.step takes the live keys)
        int na = 0, ns = 0;
        int rc = compact(e, 0, e->n_active, &na); if (rc) return rc;
        rc = compact(e, 1, e->n_sched, &ns); if (rc) return rc;
        e->n_active = na; e->n_sched = ns; e->n_sched_vehicles -= removed;
        nA = na; nS = ns;
        if (e->clock_slot >= 0 && e->mixed_order) {
          std::vector<int8_t> kinds(ns);
          HIPOK(hipMemcpy(kinds.data(), d.sched_kind, ns, hipMemcpyDeviceToHost));
          e->clock_slot = -1;
          for (int q = 0; q < ns; q++) if (kinds[q] == TS_AGENT_CLOCK) { e->clock_slot = q; break; }
        }
      }
    }
    if (svc_on) {
      // on_target_reached inside step_decide for vehicles that stay on the grid (vehicle_base.py:657-661): apply the
      // flag changes now that no decider can see them half-way, then the host part in decide order
      HIPOK(hipMemcpyAsync(e->hint + 2, &d.cnt->arr_n, sizeof(int), hipMemcpyDeviceToHost, st));
      HIPOK(hipStreamSynchronize(st));
      const int n_rec = e->hint[2];
      if (n_rec > 0) {
        int rc = fetch_records(n_rec);
        if (rc) return rc;
        hipLaunchKernelGGL(k_decide_arrive, dim3(nblk(n_rec)), dim3(BLK), 0, st, d, 0, n_rec);
        std::vector<std::pair<int, int>> order;   // (decide index, vehicle)
        for (int k = 0; k < n_rec; k++) if (recs[3 * k + 2] == AR_DECIDE) order.push_back({recs[3 * k], recs[3 * k + 1]});
        std::sort(order.begin(), order.end());
        for (auto& o : order) { int k = svc_find(e, o.second); if (k >= 0) svc_start(e, e->svc[k]); }
      }
    }
  }

  if (careful) { shuffle_start(e, nS); joiner.done = false; }
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
    if (seq && nA > 0) {
      // kind 2 = a vehicle about to step: its step_decide runs first (ServiceVehicleAgent.step returns before it while servicing,
      // vehicle_service.py:43-49).  ref = its index in the decide order.
      std::vector<uint32_t> rk((size_t)nS);
      std::vector<int8_t> kinds((size_t)nS);
      std::vector<int32_t> refs((size_t)nS), aidx((size_t)e->n_vehicles_total);
      std::vector<uint16_t> fl((size_t)e->n_vehicles_total);
      HIPOK(hipMemcpyAsync(rk.data(), d.rank, (size_t)nS * 4, hipMemcpyDeviceToHost, st));
      HIPOK(hipMemcpyAsync(kinds.data(), d.sched_kind, (size_t)nS, hipMemcpyDeviceToHost, st));
      HIPOK(hipMemcpyAsync(refs.data(), d.sched_ref, (size_t)nS * 4, hipMemcpyDeviceToHost, st));
      HIPOK(hipMemcpyAsync(aidx.data(), d.active_idx, (size_t)e->n_vehicles_total * 4, hipMemcpyDeviceToHost, st));
      HIPOK(hipMemcpyAsync(fl.data(), d.flags, (size_t)e->n_vehicles_total * 2, hipMemcpyDeviceToHost, st));
      HIPOK(hipStreamSynchronize(st));
      for (int q = 0; q < nS; q++) {
        if (kinds[q] != K_VEHICLE) continue;
        const int vid = refs[q];
        if (fl[vid] & VF_SERVICING) continue;
        points.push_back(Point{rk[q], 2, aidx[vid], q});
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
          if (h.hid == 0) {
            int rc = rain_manager_step(e, discs); if (rc) return rc;
            if (seq && discs.n >= 0) {   // the vehicles that decide after the manager in this tick read the new rain_map (vehicle_base.py:104)
              hipLaunchKernelGGL(k_rain_map, dim3(nblk((long long)e->N)), dim3(BLK), 0, st, d.rain, e->W, e->H, e->prev_discs, discs);
              e->prev_discs = discs;
              discs.n = -1;
            }
          }
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
        e->amap_valid = false;   // the rounds below move vehicles and switch lights
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
      if (!last && points[pi].kind == 2) {
        // step_decide of the vehicle whose turn it is (vehicle_base.py:669-670), alone: draws from the stream where it stands,
        // searches on the maps as they are, flags / parking / despawn applied before anybody else looks
        const int i = points[pi].ref;
        d.elapsed = e->C.elapsed;
        e->amap_valid = false;
        d.dec_expect = i + 1;
        HIPOK(hipMemsetAsync(&d.cnt->dec_arrived, 0, sizeof(int), st));
        int rc = decide_range(i, i + 1);
        d.dec_expect = 0;
        if (rc) return rc;
        int after[2] = {0, 0};
        HIPOK(hipMemcpyAsync(&after[0], &d.cnt->dec_arrived, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&after[1], &d.cnt->arr_n, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        if (after[0] == i + 1) hipLaunchKernelGGL(k_decide_despawn, dim3(1), dim3(64), 0, st, d, P, i, -1);
        if (svc_on && after[1] > arr_read) {
          const int first = arr_read;
          rc = fetch_records(after[1]);
          if (rc) return rc;
          hipLaunchKernelGGL(k_decide_arrive, dim3(nblk(after[1] - first)), dim3(BLK), 0, st, d, first, after[1]);
          for (size_t k = 0; k + 2 < recs.size(); k += 3)
            if (recs[k + 2] == AR_DECIDE) { int q = svc_find(e, recs[k + 1]); if (q >= 0) svc_start(e, e->svc[q]); }
        }
        e->amap_valid = false;
        e->hint[3] = dev_error; e->hint[1] = dev_deaths;
        continue;      // (its movement belongs to the rounds in front of the next point)
      }
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
    if (e->hint[3]) return fail(e, e->hint[3], "device-side error: a vehicle despawned inside the decide phase without the host expecting it (internal error)");
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
  p->stuck_despawn_enabled = 0; p->stuck_despawn_threshold = 3600; p->stuck_despawn_threshold_intersection = 20;
  p->pathfinding_batching = 1;
}

int ts_create(const TsWorld* w, const TsParams* params, ts_handle* out) {
  if (!w || !params || !out || w->width <= 0 || w->height <= 0 || !w->allowed_dirs_map || !w->is_road_map ||
      !w->road_type_map || !w->intersection_map)
    return TS_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return TS_E_DEVICE;
  if (hipSetDevice(g_device) != hipSuccess) return TS_E_DEVICE;
  {
    // ranges the kernels rely on (int8 speed fields, the MAXB-cell bypass buffers, clz of the speed span)
    const TsParams& q = *params;
    if (q.vehicle_min_speed < 0 || q.vehicle_max_speed < q.vehicle_min_speed || q.vehicle_max_speed > 127) return TS_E_INVALID;
    if (q.max_contraflow_overtake_steps < 0 || q.max_contraflow_stuck_detour_steps < 0 || q.vehicle_awareness_range < 0 ||
        q.rain_speed_reduction < 0 || q.pathfinding_cooldown < 0)
      return TS_E_INVALID;
    if (q.max_contraflow_overtake_steps > MAXB || q.max_contraflow_stuck_detour_steps > MAXB) return TS_E_UNSUPPORTED;
  }
  E* e = new E();
  memset(&e->d, 0, sizeof(e->d));
  memset(&e->C, 0, sizeof(e->C));
  e->P = *params;
  e->W = w->width; e->H = w->height; e->N = w->width * w->height;
  Dev& d = e->d;
  d.W = e->W; d.H = e->H; d.N = e->N;
  d.W8 = (e->W + 7) / 8; d.H8 = (e->H + 7) / 8;
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
  // search nodes: every cell a search can stand on - roads, cells with flow bits, cells a neighbour's flow bit points at
  // - numbered in the 8 x 8-tiled order of the A* snapshot
  int32_t* t_node = nullptr;
  {
    const int W = e->W, H = e->H;
    std::vector<uint8_t> is_node(N, 0);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const size_t c = (size_t)y * W + x;
        const int a = w->allowed_dirs_map[c] & 15;
        if (w->is_road_map[c] == 1 || a) is_node[c] = 1;
        if ((a & 1) && y + 1 < H) is_node[c + W] = 1;
        if ((a & 2) && x + 1 < W) is_node[c + 1] = 1;
        if ((a & 4) && y > 0) is_node[c - W] = 1;
        if ((a & 8) && x > 0) is_node[c - 1] = 1;
      }
    std::vector<int32_t> node(N, -1);
    int n_nodes = 0;
    for (int ty = 0; ty < d.H8; ty++)
      for (int tx = 0; tx < d.W8; tx++)
        for (int yy = 0; yy < 8; yy++)
          for (int xx = 0; xx < 8; xx++) {
            const int x = tx * 8 + xx, y = ty * 8 + yy;
            if (x < W && y < H && is_node[(size_t)y * W + x]) node[(size_t)y * W + x] = n_nodes++;
          }
    d.n_nodes = n_nodes;
    if (hipMalloc((void**)&t_node, N * 4) != hipSuccess) { (void)hipFree(t_allowed); (void)hipFree(t_road_type); (void)hipFree(t_inter); return bail(TS_E_DEVICE); }
    ok &= hipMemcpy(t_node, node.data(), N * 4, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (params->respect_awareness) {
    // straight runs of road cells ending in each cell, for the field-of-view test of the searches (Dev::fovrun)
    const int W = e->W, H = e->H;
    std::vector<uint16_t> run((size_t)N * 4, 0);
    auto road = [&](int x, int y) { return w->is_road_map[(size_t)y * W + x] == 1; };
    for (int x = 0; x < W; x++) {
      for (int y = 0; y < H; y++) run[((size_t)y * W + x) * 4 + 0] = road(x, y) ? (uint16_t)std::min(65535, (y > 0 ? run[((size_t)(y - 1) * W + x) * 4 + 0] : 0) + 1) : 0;
      for (int y = H - 1; y >= 0; y--) run[((size_t)y * W + x) * 4 + 1] = road(x, y) ? (uint16_t)std::min(65535, (y + 1 < H ? run[((size_t)(y + 1) * W + x) * 4 + 1] : 0) + 1) : 0;
    }
    for (int y = 0; y < H; y++) {
      for (int x = 0; x < W; x++) run[((size_t)y * W + x) * 4 + 2] = road(x, y) ? (uint16_t)std::min(65535, (x > 0 ? run[((size_t)y * W + x - 1) * 4 + 2] : 0) + 1) : 0;
      for (int x = W - 1; x >= 0; x--) run[((size_t)y * W + x) * 4 + 3] = road(x, y) ? (uint16_t)std::min(65535, (x + 1 < W ? run[((size_t)y * W + x + 1) * 4 + 3] : 0) + 1) : 0;
    }
    const size_t nt = (size_t)d.W8 * d.H8 * 64;
    std::vector<unsigned long long> tiled(nt, 0ull);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const size_t t = ((((size_t)(y >> 3) * d.W8 + (size_t)(x >> 3)) << 6) | (size_t)((y & 7) << 3) | (size_t)(x & 7));
        const uint16_t* r = &run[((size_t)y * W + x) * 4];
        tiled[t] = (unsigned long long)r[0] | ((unsigned long long)r[1] << 16) | ((unsigned long long)r[2] << 32) | ((unsigned long long)r[3] << 48);
      }
    unsigned long long* fr = nullptr;
    ok &= dalloc(e, &fr, nt) == hipSuccess;
    if (fr) ok &= hipMemcpy(fr, tiled.data(), nt * 8, hipMemcpyHostToDevice) == hipSuccess;
    d.fovrun = fr;
  }
  hipLaunchKernelGGL(k_cells_init, dim3(nblk((long long)N)), dim3(BLK), 0, st, d.cell, (int)N, t_allowed, d.is_road, t_road_type, t_inter, t_node);
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
  (void)hipFree(t_allowed); (void)hipFree(t_road_type); (void)hipFree(t_inter); (void)hipFree(t_node);
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
  if (e->quad_stream) { (void)hipStreamSynchronize(e->quad_stream); (void)hipStreamDestroy(e->quad_stream); }
  if (e->quad_ev0) (void)hipEventDestroy(e->quad_ev0);
  if (e->quad_ev1) (void)hipEventDestroy(e->quad_ev1);
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
  if (t->blk_inner_cells && (t->food_consumption_ticks <= 0 || t->waste_production_ticks <= 0))
    return fail(e, TS_E_INVALID, "food_consumption_ticks / waste_production_ticks must be positive");
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
  struct Temps { void* p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; ~Temps() { for (void* q : p) if (q) (void)hipFree(q); } } temps;
  HIPOK(hipMalloc((void**)&ds, (size_t)n * 4)); temps.p[0] = ds; HIPOK(hipMalloc((void**)&dg, (size_t)n * 4)); temps.p[1] = dg;
  HIPOK(hipMalloc((void**)&dp, (size_t)n * 4)); temps.p[2] = dp; HIPOK(hipMalloc((void**)&dl, (size_t)n * 4)); temps.p[3] = dl;
  HIPOK(hipMalloc((void**)&dof, (size_t)n * 4)); temps.p[4] = dof; HIPOK(hipMalloc((void**)&dser, (size_t)n)); temps.p[5] = dser;
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
                     e->n_sched, e->C.elapsed, e->d_overflow, e->d_total, (e->d.amap && e->amap_valid) ? 1 : 0);
  HIPOK(hipMemcpyAsync(e->hint, e->d_total, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPOK(hipStreamSynchronize(st));
  if (e->hint[0] > 0) hipLaunchKernelGGL(k_spawn_serial, dim3(1), dim3(64), 0, st, e->d, e->d_overflow, e->hint[0]);
  HIPOK(hipStreamSynchronize(st));
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
  if (start == goal) e->standing_possible = true;   // it despawns inside the next decide phase (tick())
  std::vector<int32_t> s1{start}, g1{goal}, p1{pop_type}, l1{0};
  std::vector<uint32_t> o1{0}, enc;
  int rc = add_vehicles_core(e, 1, s1, g1, p1, l1, o1, enc);
  if (rc) return rc;
  return plan_vehicle(e, e->n_vehicles_total - 1, start, goal);
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
  rc = ensure_slots(e);
  if (rc) return rc;
  rc = ensure_amap(e);
  if (rc) return rc;
  for (int attempt = 0; attempt < 3; attempt++) {
    hipLaunchKernelGGL(k_spawn_plan, dim3(1), dim3(64), 0, e->stream, d, e->P, e->slots, vid, e->d_status);
    HIPOK(hipMemcpyAsync(e->hint, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPOK(hipStreamSynchronize(e->stream));
    if (e->hint[0] != -2) break;
    rc = pool_make_room(e, (size_t)e->slots.cap + (1u << 16));  // pool full: make room and plan again
    if (rc) return rc;
  }
  if (e->hint[0] == -2) return fail(e, TS_E_CAPACITY, "path pool exhausted while planning a spawn");
  if (e->hint[0] < 0) return fail(e, TS_E_CAPACITY, "an A* search exceeded its heap or path buffers");
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
      if (sx == gx && sy == gy) e->standing_possible = true;   // it despawns inside the next decide phase (tick())
      if ((long long)e->n_sched + 1 >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^24 agents");
      int rc = add_vehicle_planned(e, sy * W + sx, gy * W + gx, population_type ? population_type[i] : TS_POP_UNDEFINED);
      if (rc) return rc;
    }
    return TS_OK;
  }
  if ((long long)e->n_sched + n >= (long long)RANK_MASK) return fail(e, TS_E_CAPACITY, "schedule exceeds 2^24 agents");
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
    if (sx == gx && sy == gy) e->standing_possible = true;
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

int ts_remove_vehicle(ts_handle e, int32_t spawn_idx, int32_t population_type) {
  if (!e) return TS_E_INVALID;
  if (population_type != TS_POP_INTERNAL && population_type != TS_POP_THROUGH) population_type = TS_POP_UNDEFINED;
  if (spawn_idx < 0 || spawn_idx >= e->n_vehicles_total) return fail(e, TS_E_INVALID, "no such live vehicle");
  Dev& d = e->d;
  uint16_t fl = 0;
  HIPOK(hipMemcpy(&fl, d.flags + spawn_idx, 2, hipMemcpyDeviceToHost));
  if (!(fl & VF_ALIVE)) return fail(e, TS_E_INVALID, "no such live vehicle");
  if (fl & VF_SVC) return fail(e, TS_E_UNSUPPORTED, "service vehicles cannot be removed by the host");
  hipLaunchKernelGGL(k_remove_one, dim3(1), dim3(64), 0, e->stream, d, spawn_idx, (int)population_type);
  // the lists close up at once (the reference's list.remove / schedule.remove): the next tick shuffles the live keys
  int na = 0, ns = 0;
  int rc = compact(e, 0, e->n_active, &na); if (rc) return rc;
  rc = compact(e, 1, e->n_sched, &ns); if (rc) return rc;
  e->n_active = na; e->n_sched = ns; e->n_sched_vehicles -= 1;
  if (e->clock_slot >= 0 && e->mixed_order) {
    std::vector<int8_t> kinds(ns);
    HIPOK(hipMemcpy(kinds.data(), d.sched_kind, ns, hipMemcpyDeviceToHost));
    e->clock_slot = -1;
    for (int q = 0; q < ns; q++) if (kinds[q] == TS_AGENT_CLOCK) { e->clock_slot = q; break; }
  }
  HIPOK(hipMemsetAsync(&d.cnt->deaths, 0, sizeof(int), e->stream));
  e->amap_valid = false;
  return sync_counters(e);
}

int ts_upload_map(ts_handle e, int32_t which, const int8_t* src) {
  if (!e || !src) return TS_E_INVALID;
  int8_t* m = which == TS_MAP_STOP ? e->d.stop : which == TS_MAP_RAIN ? e->d.rain : nullptr;
  if (!m) return fail(e, TS_E_INVALID, "only stop_map and rain_map are host-writable");
  HIPOK(hipMemcpy(m, src, e->N, hipMemcpyHostToDevice));
  e->amap_valid = false;
  if (which == TS_MAP_STOP) {
    hipLaunchKernelGGL(k_plane_to_cells, dim3(nblk((long long)e->N)), dim3(BLK), 0, e->stream, e->d.cell, e->N, e->d.stop, 1);
    HIPOK(hipStreamSynchronize(e->stream));
  }
  return TS_OK;
}
int ts_debug_set_occupancy(ts_handle e, const int8_t* src) {
  if (!e || !src) return TS_E_INVALID;
  HIPOK(hipMemcpy(e->d.occ, src, e->N, hipMemcpyHostToDevice));
  e->amap_valid = false;
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

int ts_cached_stats(ts_handle e, TsCachedStats* out) {
  if (!e || !out) return TS_E_INVALID;
  *out = e->gen.cs;
  return TS_OK;
}

int ts_set_replan_sharding(ts_handle e, int32_t rank, int32_t world, ts_exchange_fn fn, void* user) {
  if (!e || world < 1 || rank < 0 || rank >= world) return TS_E_INVALID;
  if (world > 1 && !fn) return fail(e, TS_E_INVALID, "sharded replans need an exchange callback");
  e->dist_rank = rank; e->dist_world = world; e->dist_fn = world > 1 ? fn : nullptr; e->dist_user = user; e->dist_dev = false;
  return TS_OK;
}
int ts_set_replan_sharding_device(ts_handle e, int32_t rank, int32_t world, ts_exchange_fn fn, void* user) {
  int rc = ts_set_replan_sharding(e, rank, world, fn, user);
  if (rc == TS_OK) e->dist_dev = world > 1;
  return rc;
}

// profiling hook (not part of include/trafficsim.h): the eight debug words the last replanning / search kernel left
int ts_debug_read(ts_handle e, int32_t* out8) {
  if (!e || !out8) return TS_E_INVALID;
  HIPOK(hipStreamSynchronize(e->stream));
  HIPOK(hipMemcpy(out8, e->d.cnt->dbg, sizeof(int) * 8, hipMemcpyDeviceToHost));
  if (getenv("TS_KPROF")) {
    long long pr[8];
    HIPOK(hipMemcpy(pr, e->d.cnt->prof, sizeof(pr), hipMemcpyDeviceToHost));
    fprintf(stderr, "[kprof] top %lld | issue loads %lld | sift-down %lld | goal+stale %lld | evaluate %lld | commit stores %lld | pushes %lld | loop %lld\n", pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7]);
  }
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
  if (maximum_steps < e->N && maximum_steps > A_STEPS_MAX)
    return fail(e, TS_E_UNSUPPORTED, "a binding maximum_steps above 4094 is not carried (use >= width * height for 'unlimited')");
  int rc = ensure_density(e, e->d.occ);  // "evaluated on the engine's current maps"
  if (rc) return rc;
  e->density_valid = false;
  rc = ensure_slots(e);
  if (rc) return rc;
  e->amap_valid = false;
  rc = ensure_amap(e);
  if (rc) return rc;
  e->amap_valid = false;
  hipLaunchKernelGGL(k_astar_single, dim3(1), dim3(64), 0, e->stream, e->d, e->P, e->slots, sy * e->W + sx, gy * e->W + gx, soft,
                     ignore_flow, maximum_steps, e->d_status);
  HIPOK(hipMemcpyAsync(e->hint, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  const int len = e->hint[0];
  if (len < 0) return fail(e, TS_E_CAPACITY, "an A* search exceeded its heap or path buffers");
  if (len > cap_cells) return TS_E_CAPACITY;
  if (len > 0) {
    std::vector<int32_t> cells(len);
    HIPOK(hipMemcpy(cells.data(), e->slots.cells, (size_t)len * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < len; k++) { out_xy[2 * k] = cells[k] % e->W; out_xy[2 * k + 1] = cells[k] / e->W; }
  }
  return len;
}

}  // extern "C"
