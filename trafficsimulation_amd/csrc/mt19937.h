// Host-side MT19937 with CPython `random` semantics (Lib/random.py 3.10, Modules/_randommodule.c).
//
// The engine's two streams (module-level `random` and `model.random`) are consumed on the host, because
// their consumption is a data-dependent sequential chain (SURVEY.md §7 "Hard parts").  What is NOT
// sequential is producing the words: MTPipe runs the generator on a producer thread, block by block
// (624 words per twist), into a ring of tempered words, so the consumers (the decide-phase scan and the
// scheduler shuffle) only read.  The CPython-visible state (624 key words + index) at the consumption
// point can be reconstructed at any time from the ring of raw blocks.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

struct HostMT {
  uint32_t mt[624];
  uint32_t idx = 625;

  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  // init_by_array(key): random.seed(int) feeds the 32-bit little-endian limbs of abs(seed)
  void seed_u64(uint64_t s) {
    uint32_t key[2] = {(uint32_t)s, (uint32_t)(s >> 32)};
    const int len = key[1] ? 2 : 1;
    init_genrand(19650218U);
    int i = 1, j = 0;
    for (int k = 624; k; k--) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
      if (++j >= len) j = 0;
    }
    for (int k = 623; k; k--) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000U;
  }
};

// one twist of the 624-word state, in place
static inline void mt_twist(uint32_t* mt) {
  const uint32_t M = 0x9908b0dfU;
  int k = 0;
  for (; k < 227; k++) {
    uint32_t y = (mt[k] & 0x80000000U) | (mt[k + 1] & 0x7fffffffU);
    mt[k] = mt[k + 397] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & M);
  }
  for (; k < 623; k++) {
    uint32_t y = (mt[k] & 0x80000000U) | (mt[k + 1] & 0x7fffffffU);
    mt[k] = mt[k - 227] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & M);
  }
  uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
  mt[623] = mt[396] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & M);
}
static inline uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= (y >> 18);
  return y;
}

class MTPipe {
 public:
  static constexpr uint64_t TW_CAP = 1ull << 23;   // tempered-word ring (32 MB)
  static constexpr uint64_t RB_CAP = 1ull << 14;   // raw-block ring (16384 x 624 words)
  static constexpr uint64_t MAX_AHEAD_BLOCKS = 12000;  // < min(TW_CAP / 624, RB_CAP) with slack

  MTPipe() : tw_(TW_CAP), raw_(RB_CAP * 624) {}
  // Optional "roll table" for _randbelow(span) (getrandbits(k) retried while >= span): take(p) = number of
  // words such a draw consumes when it starts at word p (0 = not known / longer than 255: use the slow loop).
  // Filled by the producer, so the consumer's serial chain is one table load per draw.  Call before seed().
  void set_roll(uint32_t span) {
    roll_span_ = span;
    roll_shift_ = span ? __builtin_clz(span) : 0;
    if (span && take_.empty()) take_.assign(TW_CAP, 0);
  }
  ~MTPipe() { stop(); }
  MTPipe(const MTPipe&) = delete;

  // random.setstate(): `mt` = the 624 key words, `idx` in [0, 624]
  void seed(const uint32_t* mt, uint32_t idx) {
    stop();
    memcpy(cur_, mt, sizeof(cur_));
    memcpy(&raw_[0], mt, sizeof(cur_));
    for (int j = 0; j < 624; j++) tw_[j] = mt_temper(mt[j]);
    produced_blocks_.store(1, std::memory_order_release);
    take_blocks_.store(0, std::memory_order_release);
    if (roll_span_) roll_scan_block(0);
    consumed_ = idx;
    consumed_blocks_.store(0, std::memory_order_release);
    seeded_ = true;
    quit_.store(false);
    producer_ = std::thread([this]() { produce(); });
  }
  void seed_u64(uint64_t s) { HostMT m; m.seed_u64(s); seed(m.mt, m.idx); }
  bool seeded() const { return seeded_; }

  // CPython-visible state at the consumption point
  void state(uint32_t* mt_out, uint32_t* idx_out) const {
    uint64_t b = consumed_ / 624, off = consumed_ % 624;
    if (off == 0 && consumed_ > 0) { b -= 1; off = 624; }
    memcpy(mt_out, &raw_[(b & (RB_CAP - 1)) * 624], 624 * 4);
    *idx_out = (uint32_t)off;
  }

  // ---- consumer side (one thread at a time) ----
  uint64_t pos() const { return consumed_; }
  // make words [pos, pos + n) readable
  inline void need(uint64_t n) {
    const uint64_t want_blocks = (consumed_ + n + 623) / 624;
    while (produced_blocks_.load(std::memory_order_acquire) < want_blocks) std::this_thread::yield();
  }
  inline uint32_t at(uint64_t abs_word) const { return tw_[abs_word & (TW_CAP - 1)]; }
  // make take(p) final for p in [pos, pos + n)
  inline void need_take(uint64_t n) {
    const uint64_t want_blocks = (consumed_ + n + 623) / 624;
    while (take_blocks_.load(std::memory_order_acquire) < want_blocks) std::this_thread::yield();
  }
  inline uint32_t take(uint64_t abs_word) const { return take_[abs_word & (TW_CAP - 1)]; }
  inline void advance_to(uint64_t abs_word) {
    consumed_ = abs_word;
    consumed_blocks_.store(abs_word / 624, std::memory_order_release);
  }
  // convenience (slow path / setup code): one word, random(), _randbelow, randint
  inline uint32_t next() { need(1); uint32_t w = at(consumed_); advance_to(consumed_ + 1); return w; }
  inline double random() {
    uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
  inline uint32_t randbelow(uint32_t n) {
    const int k = 32 - __builtin_clz(n);
    uint32_t r = next() >> (32 - k);
    while (r >= n) r = next() >> (32 - k);
    return r;
  }
  inline int randint(int a, int b) { return a + (int)randbelow((uint32_t)(b - a + 1)); }

 private:
  void stop() {
    if (producer_.joinable()) { quit_.store(true); producer_.join(); }
  }
  void produce() {
    uint64_t b = produced_blocks_.load();
    while (!quit_.load(std::memory_order_relaxed)) {
      // keep at most MAX_AHEAD_BLOCKS unconsumed blocks (the block before the consumption point stays too)
      uint64_t cb = consumed_blocks_.load(std::memory_order_acquire);
      if (b > cb + MAX_AHEAD_BLOCKS) { std::this_thread::sleep_for(std::chrono::microseconds(50)); continue; }
      mt_twist(cur_);
      memcpy(&raw_[(b & (RB_CAP - 1)) * 624], cur_, sizeof(cur_));
      const uint64_t base = b * 624;
      if (((base & (TW_CAP - 1)) + 624) <= TW_CAP) {
        uint32_t* dst = &tw_[base & (TW_CAP - 1)];
        for (int j = 0; j < 624; j++) dst[j] = mt_temper(cur_[j]);
      } else {
        for (int j = 0; j < 624; j++) tw_[(base + j) & (TW_CAP - 1)] = mt_temper(cur_[j]);
      }
      if (roll_span_) {
        roll_scan_block(b);
        // the trailing run of rejected words of block b-1 can now be closed with block b's first entry
        uint32_t carry = take_[(b * 624) & (TW_CAP - 1)];
        for (int j = 623; j >= 0; j--) {
          const uint64_t q = ((b - 1) * 624 + (uint64_t)j) & (TW_CAP - 1);
          if ((tw_[q] >> roll_shift_) < roll_span_) break;
          carry = (carry == 0 || carry >= 255) ? 0 : carry + 1;
          take_[q] = (uint8_t)carry;
        }
        take_blocks_.store(b, std::memory_order_release);  // blocks [0, b) are final
      }
      b++;
      produced_blocks_.store(b, std::memory_order_release);
    }
  }
  // provisional backward scan of one block: a run of rejects that reaches the block end stays 0 (unknown)
  void roll_scan_block(uint64_t b) {
    uint32_t t = 0;
    for (int j = 623; j >= 0; j--) {
      const uint64_t q = (b * 624 + (uint64_t)j) & (TW_CAP - 1);
      if ((tw_[q] >> roll_shift_) < roll_span_) t = 1;
      else t = (t == 0 || t >= 255) ? 0 : t + 1;
      take_[q] = (uint8_t)t;
    }
  }

  std::vector<uint32_t> tw_, raw_;
  std::vector<uint8_t> take_;
  uint32_t roll_span_ = 0;
  int roll_shift_ = 0;
  uint32_t cur_[624];
  std::atomic<uint64_t> produced_blocks_{0}, consumed_blocks_{0}, take_blocks_{0};
  uint64_t consumed_ = 0;
  std::atomic<bool> quit_{false};
  std::thread producer_;
  bool seeded_ = false;
};
