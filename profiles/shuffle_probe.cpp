// Micro-benchmark of the scheduler shuffle stages (draw extraction, swap loop variants) on the host CPU; not part of the product.
// g++ -O2 -std=c++17 -pthread profiles/shuffle_probe.cpp -o /tmp/shuffle_probe
#include "../trafficsimulation_amd/csrc/mt19937.h"
#include <cstdio>
#include <immintrin.h>
#include <vector>
#include <chrono>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int SH_CH = 1 << 12;
static std::vector<uint32_t> JB;
static int draws(MTPipe& r, int n) {
  if ((int)JB.size() < n + 64) JB.resize((size_t)n + 64);
  uint32_t* jb = JB.data();
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
      while (nn > band_end) {
        if (w + 256 >= limit) { r.advance_to(w); r.need(8192); limit = w + 8192 - 256; }
        int burst = 64;
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
  }
  return (int)cnt;
}
template <int D>
static uint32_t swaps2(int n, std::vector<uint32_t>& perm, uint32_t cs) {
  perm.resize(n);
  uint32_t* p = perm.data();
  for (int i = 0; i < n; i++) p[i] = (uint32_t)i;
  const int total = n - 1;
  const uint32_t* jb = JB.data();
  for (int q = 0; q < D && q < total; q++) __builtin_prefetch(&p[jb[q]], 1, 3);
  int i = n - 1;
  for (int q = 0; q < total; q++, i--) {
    __builtin_prefetch(&p[jb[q + D]], 1, 3);     // JB has 64 spare entries; they index inside p (zero / stale values < n)
    const uint32_t j = jb[q];
    const uint32_t a = p[i], b = p[j];
    p[i] = b; p[j] = a;
  }
  uint32_t cpos = cs;
  for (int k = 0; k < n; k++) if (p[k] == cs) cpos = (uint32_t)k;
  return cpos;
}
static std::vector<uint32_t> JB2;
__attribute__((target("avx512f,popcnt"))) static int draws_v(MTPipe& r, int n) {
  if ((int)JB2.size() < n + 64) JB2.resize((size_t)n + 64);
  uint32_t* jb = JB2.data();
  uint64_t w = r.pos();
  uint32_t cnt = 0;
  const uint32_t* ring = r.ring();
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
        int burst = 4;   // vector steps of 16 words
        while (burst > 0 && nn > band_end + 16) {
          const uint64_t off = w & (MTPipe::TW_CAP - 1);
          if (off + 16 > MTPipe::TW_CAP) break;
          const __m512i c = _mm512_srl_epi32(_mm512_loadu_si512((const void*)(ring + off)), vshift);
          const __mmask16 sure = _mm512_cmplt_epu32_mask(c, _mm512_set1_epi32((int)(nn - 16)));
          const __mmask16 maybe = _mm512_cmplt_epu32_mask(c, _mm512_set1_epi32((int)nn));
          if (sure != maybe) break;      // a value within 16 of the bound: the scalar loop decides
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
  }
  return (int)cnt;
}
static uint32_t swaps(int n, std::vector<uint32_t>& perm, uint32_t cs) {
  perm.resize(n);
  uint32_t* p = perm.data();
  for (int i = 0; i < n; i++) p[i] = (uint32_t)i;
  uint32_t cpos = cs;
  const int total = n - 1;
  int done = 0;
  while (done < total) {
    const int m = std::min(total - done, SH_CH);
    const uint32_t* jb = JB.data() + done;
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
  }
  return cpos;
}
int main() {
  MTPipe r; r.seed_u64(12345);
  const int n = 1083000;
  std::vector<uint32_t> perm;
  std::this_thread::sleep_for(std::chrono::milliseconds(300));   // let the generator run ahead
  {   // the vector extraction must give the same draws and end at the same word
    MTPipe a, b; a.seed_u64(777); b.seed_u64(777);
    for (int nn : {5, 17, 64, 1000, 4097, 65536, 65537, 300000, 1083000}) {
      int ca = draws(a, nn), cb = draws_v(b, nn);
      bool same = ca == cb && a.pos() == b.pos();
      for (int k = 0; k < ca && same; k++) same = JB[k] == JB2[k];
      printf("n=%d: %s (%d draws, pos %llu / %llu)\n", nn, same ? "same" : "DIFFERENT", ca, (unsigned long long)a.pos(), (unsigned long long)b.pos());
    }
  }
  for (int rep = 0; rep < 3; rep++) { double t0 = now_ms(); int c = draws_v(r, n); printf("draws_v %.3f ms (%d)\n", now_ms() - t0, c); std::this_thread::sleep_for(std::chrono::milliseconds(100)); }
  for (int rep = 0; rep < 6; rep++) {
    double t0 = now_ms(); int c = draws(r, n); double t1 = now_ms(); uint32_t cp = swaps(n, perm, 1000); double t2 = now_ms();
    double t3 = now_ms(); uint32_t c16 = swaps2<16>(n, perm, 1000); double t4 = now_ms(); uint32_t c32 = swaps2<32>(n, perm, 1000); double t5 = now_ms(); uint32_t c64 = swaps2<64>(n, perm, 1000); double t6 = now_ms();
    printf("draws %.3f ms (%d)  swaps %.3f ms (cpos %u)  D16 %.3f D32 %.3f D64 %.3f  (%u %u %u)\n", t1 - t0, c, t2 - t1, cp, t4 - t3, t5 - t4, t6 - t5, c16, c32, c64);
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
}
