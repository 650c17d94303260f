// CPU-side harness for trafficsimulation_amd/csrc/mt19937.h (pure host C++): the word pipe that feeds both MT19937
// streams of the engine.  Built by tests/test_host_mtpipe.py with g++ (and once more with -fsanitize=thread).
#include <cstdint>
#include <cstring>
#include <vector>
#include <thread>
#include <chrono>
#include <atomic>
#include <cstdio>
#include "../../trafficsimulation_amd/csrc/mt19937.h"

extern "C" {
// consume `n` words after random.setstate(mt, idx): xor-fold of all words, the last 16 words, and the CPython-visible
// state at the end.  `stride` > 1 consumes in jumps (need + advance_to), as the engine's scans do.
int mtpipe_run(const uint32_t* mt, uint32_t idx, uint64_t n, uint32_t stride, uint32_t* fold_out, uint32_t* last16,
               uint32_t* mt_out, uint32_t* idx_out) {
  MTPipe p;
  p.seed(mt, idx);
  uint32_t fold = 0;
  uint64_t done = 0;
  while (done < n) {
    const uint64_t k = std::min<uint64_t>(stride, n - done);
    p.need(k);
    const uint64_t base = p.pos();
    for (uint64_t j = 0; j < k; j++) {
      const uint32_t w = p.at(base + j);
      fold = (fold << 1 | fold >> 31) ^ w;
      if (n - (done + j) <= 16) last16[16 - (n - (done + j))] = w;
    }
    p.advance_to(base + k);
    done += k;
  }
  *fold_out = fold;
  p.state(mt_out, idx_out);
  return 0;
}

// _randbelow(span) draws: values and, for each draw, the words take() says it consumes
int mtpipe_rolls(const uint32_t* mt, uint32_t idx, uint32_t span, int n, uint32_t* values, uint32_t* takes) {
  MTPipe p;
  p.set_roll(span);
  p.seed(mt, idx);
  for (int i = 0; i < n; i++) {
    p.need(256);
    takes[i] = p.take(p.pos());
    values[i] = p.randbelow(span);
  }
  return 0;
}
}

#ifdef MTPIPE_MAIN
// race check: a long consumption with small and large jumps while the producer runs ahead
int main() {
  HostMT m; m.seed_u64(12345);
  std::vector<uint32_t> mt(624), out(624), last(16);
  uint32_t fold, idx;
  memcpy(mt.data(), m.mt, 624 * 4);
  mtpipe_run(mt.data(), m.idx, 3000000ull, 1000, &fold, last.data(), out.data(), &idx);
  std::vector<uint32_t> v(2000), t(2000);
  mtpipe_rolls(mt.data(), m.idx, 1000003u, 2000, v.data(), t.data());
  std::printf("ok %08x %u\n", fold, idx);
  return 0;
}
#endif
