// host_shuffle.h - random.shuffle of the schedule with model.random as a two-thread pipeline that streams the permutation to the device.
// Part of the single translation unit engine.hip (included from there, in order).
#pragma once

namespace {

// random.shuffle(keys) with model.random: Fisher-Yates from the top with _randbelow's rejection
// sampling, reading pre-generated words.  Two persistent threads form a pipeline: the first extracts the draws
// (they depend only on the word stream and on n), the second applies the swaps a chunk behind it, copies every
// finished stretch of the permutation (position i is final once element i has been swapped) into the pinned
// buffer e->hrank and sends it to the device on its own stream.  Result: d_perm[q] = schedule slot stepping at
// time q (inverted to rank[slot] by k_rank_invert), the clock agent's rank in rank_clock_host.
constexpr int SH_CH = 1 << 12;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define TS_HAVE_AVX512_DRAWS 1
// The same extraction sixteen words at a time (EPYC 9575F: 0.42 ms instead of 1.3 ms per million elements, measured
// with profiles/shuffle_probe.cpp).  Element k of a vector is tested against i + 1 minus the number of accepted words
// before it, which is not known in parallel - but it lies in (nn - 16, nn], so a word below nn - 16 is certainly a draw
// and one at or above nn certainly a rejected try; a vector holding a word in between goes to the scalar loop.
__attribute__((target("avx512f,popcnt"))) void shuffle_draws_avx512(E* e, int n) {
  if ((int)e->shuffle_j.size() < n + 64) e->shuffle_j.resize((size_t)n + 64);
  uint32_t* jb = e->shuffle_j.data();
  MTPipe& r = e->rng_sched;
  const uint32_t* ring = r.ring();
  uint64_t w = r.pos();
  uint32_t cnt = 0;
  for (int hi = n - 1; hi >= 1; hi -= SH_CH) {
    const int lo = std::max(1, hi - SH_CH + 1);
    r.need((uint64_t)(hi - lo + 1) * 4 + 512 + (w - r.pos()));
    uint64_t limit = w + (uint64_t)(hi - lo + 1) * 4 + 256;
    uint32_t nn = (uint32_t)hi + 1;
    const uint32_t nn_end = (uint32_t)lo;
    while (nn > nn_end) {
      const int shift = __builtin_clz(nn);
      const uint32_t band_end = std::max(nn_end, (1u << (31 - shift)) - 1u);
      const __m128i vshift = _mm_cvtsi32_si128(shift);
      while (nn > band_end) {
        if (w + 256 >= limit) { r.advance_to(w); r.need(8192); limit = w + 8192 - 256; }
        int burst = 4;   // vector steps (at most 64 words per trip, like the scalar burst)
        while (burst > 0 && nn > band_end + 16) {
          const uint64_t off = w & (MTPipe::TW_CAP - 1);
          if (off + 16 > MTPipe::TW_CAP) break;
          const __m512i c = _mm512_srl_epi32(_mm512_loadu_si512((const void*)(ring + off)), vshift);
          const __mmask16 sure = _mm512_cmplt_epu32_mask(c, _mm512_set1_epi32((int)(nn - 16)));
          const __mmask16 maybe = _mm512_cmplt_epu32_mask(c, _mm512_set1_epi32((int)nn));
          if (sure != maybe) break;
          // (compress in a register and store all sixteen lanes: the buffer has 64 spare entries, the tail is overwritten)
          _mm512_storeu_si512((void*)(jb + cnt), _mm512_maskz_compress_epi32(sure, c));
          const uint32_t k = (uint32_t)__builtin_popcount((unsigned)sure);
          cnt += k; nn -= k; w += 16;
          burst--;
        }
        if (burst == 0) continue;
        burst = 16;
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
#endif
void shuffle_draws(E* e, int n) {
#ifdef TS_HAVE_AVX512_DRAWS
  static const bool avx512 = __builtin_cpu_supports("avx512f") && !getenv("TS_NO_AVX512");
  if (avx512) { shuffle_draws_avx512(e, n); return; }
#endif
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
    if (cs != 0xFFFFFFFFu) { uint32_t f = cpos; for (int k = lo; k < sent_hi; k++) f = p[k] == cs ? (uint32_t)k : f; cpos = f; }   // where the clock agent ended up
    if (hipMemcpyAsync(e->d_perm + lo, e->hrank + lo, (size_t)(sent_hi - lo) * 4, hipMemcpyHostToDevice, e->perm_stream) != hipSuccess)
      e->sh_err = 1;
    sent_hi = lo;
  };
  while (done < total) {
    const int avail = e->sh_progress.load(std::memory_order_acquire);
    if (avail <= done) { std::this_thread::yield(); continue; }
    const int m = std::min(avail - done, SH_CH);
    const uint32_t* jb = e->shuffle_j.data() + done;
    // the swap partners are random positions of a 4 MB array: keep a fixed number of their cache lines in flight
    // (0.98 ms instead of 1.4 ms per million swaps on the box's EPYC, profiles/shuffle_probe.cpp)
    constexpr int PF = 32;
    for (int q = 0; q < PF && q < m; q++) __builtin_prefetch(&p[jb[q]], 1, 3);
    const int hi = n - 1 - done;
    for (int q = 0; q < m; q++) {
      if (q + PF < m) __builtin_prefetch(&p[jb[q + PF]], 1, 3);
      const int i = hi - q;
      const uint32_t j = jb[q];
      const uint32_t a = p[i], b = p[j];
      p[i] = b; p[j] = a;
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

}  // namespace
