// Host-side MT19937 with CPython `random` semantics (Lib/random.py 3.10, Modules/_randommodule.c):
// the engine's two streams (module-level `random` and `model.random`) are consumed on the host,
// because their consumption is a data-dependent sequential chain (SURVEY.md §7 "Hard parts").
#pragma once
#include <cstdint>
#include <cstring>

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
  inline void regen() {
    const uint32_t M = 0x9908b0dfU;
    int k = 0;
    for (; k < 227; k++) {
      uint32_t y = (mt[k] & 0x80000000U) | (mt[k + 1] & 0x7fffffffU);
      mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1U) ? M : 0U);
    }
    for (; k < 623; k++) {
      uint32_t y = (mt[k] & 0x80000000U) | (mt[k + 1] & 0x7fffffffU);
      mt[k] = mt[k - 227] ^ (y >> 1) ^ ((y & 1U) ? M : 0U);
    }
    uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1U) ? M : 0U);
    idx = 0;
  }
  inline uint32_t next() {
    if (idx >= 624) regen();
    uint32_t y = mt[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
  }
  inline double random() {  // random.random(): 53 bits from two words
    uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
  inline uint32_t randbelow(uint32_t n) {  // Random._randbelow_with_getrandbits, n >= 1
    const int k = 32 - __builtin_clz(n);
    uint32_t r = next() >> (32 - k);
    while (r >= n) r = next() >> (32 - k);
    return r;
  }
  inline int randint(int a, int b) { return a + (int)randbelow((uint32_t)(b - a + 1)); }
};
