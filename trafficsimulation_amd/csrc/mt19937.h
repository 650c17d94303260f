// Host-side MT19937 with CPython `random` semantics (Lib/random.py 3.10, Modules/_randommodule.c).
//
// The engine's two streams (module-level `random` and `model.random`) are consumed on the host, because
// their consumption is a data-dependent sequential chain (SURVEY.md §7 "Hard parts").  What is NOT
// sequential is producing the words: MTPipe runs the generator on a producer thread, block by block
// (624 words per twist), into a ring of tempered words, so the consumers (the decide-phase scan and the
// scheduler shuffle) only read.  The CPython-visible state (624 key words + index) at the consumption
// point can be reconstructed at any time from the ring of raw blocks.
#pragma once
#include <algorithm>
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

// one twist of the 624-word state, in place (auto-vectorises; an AVX2 clone is picked at run time when the CPU
// has it - the library is built on one machine and runs on another, so no -march flags)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define TS_SIMD_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define TS_SIMD_CLONES
#endif
TS_SIMD_CLONES static void mt_twist(uint32_t* mt) {
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
TS_SIMD_CLONES static void mt_temper_block(const uint32_t* __restrict src, uint32_t* __restrict dst) {
  for (int j = 0; j < 624; j++) {
    uint32_t y = src[j];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    dst[j] = y;
  }
}
// accept flags (top bits < span) of n words as bytes; the bit packing below is cheap once this is vectorised
TS_SIMD_CLONES static void mt_accept_bytes(const uint32_t* __restrict src, uint8_t* __restrict dst, int n, int shift,
                                           uint32_t span) {
  for (int j = 0; j < n; j++) dst[j] = (uint8_t)((src[j] >> shift) < span);
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
  static constexpr uint64_t TW_CAP = 1ull << 23;        // tempered-word ring (32 MB)
  static constexpr uint64_t MAX_AHEAD_BLOCKS = 12000;   // < TW_CAP / 624 with slack
  static constexpr uint64_t CKPT_EVERY = 64;            // raw state checkpoints (for random.getstate())
  static constexpr uint64_t CKPT_CAP = 512;             // > MAX_AHEAD_BLOCKS / CKPT_EVERY + 2

  MTPipe() : ckpt_(CKPT_CAP * 624) {}
  // Storage for the tempered-word ring (TW_CAP words).  The engine passes pinned host memory so that slices
  // of the stream can be copied to the device asynchronously; without a call the pipe allocates its own.
  void use_storage(uint32_t* tw) { tw_ = tw; }
  // Optional accept bitmask for _randbelow(span) (getrandbits(k) retried while >= span): bit p is set when word
  // p would be accepted.  take(p) = number of words such a draw consumes when it starts at word p is then a
  // shift and a count-trailing-zeros on the consumer side.  Call before seed().
  void set_roll(uint32_t span) {
    roll_span_ = span;
    roll_shift_ = span ? __builtin_clz(span) : 0;
    if (span && acc_.empty()) acc_.assign(TW_CAP / 64, 0);
  }
  ~MTPipe() { stop(); }
  MTPipe(const MTPipe&) = delete;

  // random.setstate(): `mt` = the 624 key words, `idx` in [0, 624]
  void seed(const uint32_t* mt, uint32_t idx) {
    stop();
    if (!tw_) { own_tw_.assign(TW_CAP, 0); tw_ = own_tw_.data(); }
    memcpy(cur_, mt, sizeof(cur_));
    memcpy(&ckpt_[0], mt, sizeof(cur_));   // block 0 is a checkpoint
    for (int j = 0; j < 624; j++) tw_[j] = mt_temper(mt[j]);
    if (roll_span_) { std::fill(acc_.begin(), acc_.end(), 0); accept_block(0); }
    produced_s_.v.store(1, std::memory_order_release);
    consumed_ = idx;
    consumed_s_.v.store(0, std::memory_order_release);
    seeded_ = true;
    quit_.store(false);
    producer_ = std::thread([this]() { produce(); });
  }
  void seed_u64(uint64_t s) { HostMT m; m.seed_u64(s); seed(m.mt, m.idx); }
  bool seeded() const { return seeded_; }

  // CPython-visible state at the consumption point: re-twist from the nearest checkpoint
  void state(uint32_t* mt_out, uint32_t* idx_out) const {
    uint64_t b = consumed_ / 624, off = consumed_ % 624;
    if (off == 0 && consumed_ > 0) { b -= 1; off = 624; }
    const uint64_t c = b - (b % CKPT_EVERY);
    memcpy(mt_out, &ckpt_[((c / CKPT_EVERY) % CKPT_CAP) * 624], 624 * 4);
    for (uint64_t k = c; k < b; k++) mt_twist(mt_out);
    *idx_out = (uint32_t)off;
  }

  // ---- consumer side (one thread at a time) ----
  uint64_t pos() const { return consumed_; }
  // make words (and accept bits) [pos, pos + n) readable
  inline void need(uint64_t n) {
    const uint64_t want_blocks = (consumed_ + n + 623) / 624;
    while (produced_s_.v.load(std::memory_order_acquire) < want_blocks) std::this_thread::yield();
  }
  inline uint32_t at(uint64_t abs_word) const { return tw_[abs_word & (TW_CAP - 1)]; }
  const uint32_t* ring() const { return tw_; }
  // words a _randbelow(span) draw starting at abs_word consumes; 0 = more than 64 (count them by hand).
  // Needs need() to cover [abs_word, abs_word + 128).
  inline uint32_t take(uint64_t abs_word) const {
    const uint64_t q = abs_word >> 6;
    const unsigned sh = (unsigned)(abs_word & 63);
    const uint64_t a = acc_[q & (TW_CAP / 64 - 1)], b = acc_[(q + 1) & (TW_CAP / 64 - 1)];
    const uint64_t win = (a >> sh) | (sh ? (b << (64 - sh)) : 0);
    return win ? (uint32_t)__builtin_ctzll(win) + 1u : 0u;
  }
  inline void advance_to(uint64_t abs_word) {
    consumed_ = abs_word;
    consumed_s_.v.store(abs_word / 624, std::memory_order_release);
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
    quit_.store(true);
    if (producer_.joinable()) producer_.join();
  }
  void produce() {
    uint64_t b = produced_s_.v.load();
    while (!quit_.load(std::memory_order_relaxed)) {
      // keep at most MAX_AHEAD_BLOCKS unconsumed blocks in the ring
      const uint64_t cb = consumed_s_.v.load(std::memory_order_acquire);
      if (b > cb + MAX_AHEAD_BLOCKS) { std::this_thread::sleep_for(std::chrono::microseconds(20)); continue; }
      mt_twist(cur_);
      if (b % CKPT_EVERY == 0) memcpy(&ckpt_[((b / CKPT_EVERY) % CKPT_CAP) * 624], cur_, sizeof(cur_));
      const uint64_t base = b * 624;
      if (((base & (TW_CAP - 1)) + 624) <= TW_CAP) {
        mt_temper_block(cur_, &tw_[base & (TW_CAP - 1)]);
      } else {
        for (int j = 0; j < 624; j++) tw_[(base + j) & (TW_CAP - 1)] = mt_temper(cur_[j]);
      }
      if (roll_span_) accept_block(b);
      b++;
      produced_s_.v.store(b, std::memory_order_release);
    }
  }
  // accept bits of block b into the bit ring (bit index = absolute word index).  Blocks are 624 = 9.75 x 64
  // words, so a block starts and ends inside 64-bit words: the head word keeps the previous block's low bits,
  // the tail word is written with zero high bits for the next block to OR into.
  void accept_block(uint64_t b) {
    const uint64_t base = b * 624;
    uint32_t tmp[624];
    const uint64_t off = base & (TW_CAP - 1);
    const uint32_t* src;
    if (off + 624 <= TW_CAP) src = &tw_[off];
    else { for (int j = 0; j < 624; j++) tmp[j] = tw_[(base + (uint64_t)j) & (TW_CAP - 1)]; src = tmp; }
    uint8_t a8[640];
    mt_accept_bytes(src, a8, 624, roll_shift_, roll_span_);
    for (int k = 624; k < 640; k++) a8[k] = 0;
    uint64_t bits[11];
    for (int q = 0; q < 10; q++) {  // 8 flag bytes -> 8 bits with one multiply
      uint64_t acc = 0;
      for (int g = 0; g < 8; g++) {
        uint64_t v;
        memcpy(&v, &a8[q * 64 + g * 8], 8);
        acc |= ((v * 0x0102040810204080ull) >> 56) << (g * 8);
      }
      bits[q] = acc;
    }
    bits[10] = 0;
    const unsigned sh = (unsigned)(base & 63);
    const uint64_t q0 = base >> 6;
    const uint64_t M = TW_CAP / 64 - 1;
    if (sh == 0) {
      for (int q = 0; q < 10; q++) acc_[(q0 + q) & M] = bits[q];
    } else {
      const uint64_t keep = acc_[q0 & M] & ((1ull << sh) - 1);
      acc_[q0 & M] = keep | (bits[0] << sh);
      for (int q = 1; q <= 10; q++) acc_[(q0 + q) & M] = (bits[q - 1] >> (64 - sh)) | (bits[q] << sh);
    }
  }

  uint32_t* tw_ = nullptr;
  std::vector<uint32_t> own_tw_, ckpt_;
  std::vector<uint64_t> acc_;
  uint32_t roll_span_ = 0;
  int roll_shift_ = 0;
  // every cross-thread word on its own cache line
  struct alignas(128) PaddedU64 { std::atomic<uint64_t> v{0}; };
  PaddedU64 produced_s_, consumed_s_;
  alignas(128) uint32_t cur_[624];   // generator-private working state, on its own cache lines
  alignas(128) uint64_t consumed_ = 0;
  std::atomic<bool> quit_{false};
  std::thread producer_;
  bool seeded_ = false;
};
