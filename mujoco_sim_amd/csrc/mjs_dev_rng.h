// mjs_dev_rng.h — per-env MT19937 streams resident in HBM, bit-identical to numpy's legacy
// RandomState(int seed).uniform(lo, hi), which is what the reference's reset hooks consume
// (environments/dmc2gym.py:126-131 swaps env._random_state; point_reach.py:130-143,
// spaces.py:24-31 draw from it). State is struct-of-arrays: word k of env i at mt[k*N + i], so
// the (rare) 624-word regeneration is coalesced across the 64 lanes of a wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct DevRng {
  uint32_t* mt;  // [624][N]
  int32_t* pos;  // [N]
  int N;
};

__device__ inline void rng_seed_lane(const DevRng& r, int i, uint32_t seed) {
  uint32_t x = seed;
  r.mt[i] = x;
  for (int k = 1; k < 624; k++) {
    x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
    r.mt[(size_t)k * r.N + i] = x;
  }
  r.pos[i] = 624;
}

__device__ inline void rng_twist_lane(const DevRng& r, int i) {
  uint32_t* mt = r.mt + i;
  const size_t N = (size_t)r.N;
#pragma unroll 1
  for (int k = 0; k < 624; k++) {
    int k1 = (k + 1 == 624) ? 0 : k + 1;
    int km = (k + 397 >= 624) ? k + 397 - 624 : k + 397;
    uint32_t y = (mt[k * N] & 0x80000000u) | (mt[k1 * N] & 0x7fffffffu);
    mt[k * N] = mt[km * N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
}

// cursor kept in a register while one reset draws several numbers
struct RngCursor {
  int pos;
};
__device__ inline RngCursor rng_open(const DevRng& r, int i) { return RngCursor{r.pos[i]}; }
__device__ inline void rng_close(const DevRng& r, int i, RngCursor c) { r.pos[i] = c.pos; }

__device__ inline uint32_t rng_u32(const DevRng& r, int i, RngCursor& c) {
  if (c.pos >= 624) {
    rng_twist_lane(r, i);
    c.pos = 0;
  }
  uint32_t y = r.mt[(size_t)c.pos * r.N + i];
  c.pos++;
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
// RandomState.uniform(lo, hi) = lo + (hi - lo) * ((a >> 5) * 2^26 + (b >> 6)) / 2^53. numpy rounds the product and the sum
// separately: no FMA contraction here (HIP's __dmul_rn / __dadd_rn are plain operators and do not prevent it; a contracted
// 0.8 + 0.4 u differs from numpy in the last bit for ~1 draw in 10)
#pragma clang fp contract(off)
__device__ inline double rng_uniform(const DevRng& r, int i, RngCursor& c, double lo, double hi) {
  uint32_t a = rng_u32(r, i, c) >> 5, b = rng_u32(r, i, c) >> 6;
  double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;  // exact: a 2^26 + b < 2^53
  const double span = hi - lo, prod = span * u;
  return lo + prod;
}
#pragma clang fp contract(fast)  // back to hipcc's default (-ffp-contract=fast): "on" would stop cross-statement fusion in everything included later
