/* om_rng.c — ORACLE (test infrastructure): MT19937 exactly as numpy's legacy
 * RandomState(int seed) drives it. The reference swaps the env's RandomState on
 * seed() (environments/dmc2gym.py:126-131) and consumes it with
 * random_state.uniform(lo,hi) in the reset hooks (point_reach.py:130-143,
 * spaces.py:24-31). Published algorithm: Matsumoto & Nishimura 1998 (init_genrand
 * seeding); numpy: uniform = lo + (hi-lo)*((a>>5)*2^26 + (b>>6))/2^53.
 * Pinned against numpy itself in tests/test_oracle_known_answers.py. */
#include "mjs_oracle.h"

void om_rng_seed(om_rng* r, uint32_t seed) {
  r->mt[0] = seed;
  for (int i = 1; i < 624; i++) r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
  r->pos = 624;
}

static void om_rng_twist(om_rng* r) {
  uint32_t* mt = r->mt;
  for (int k = 0; k < 624; k++) {
    uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
    mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  r->pos = 0;
}

uint32_t om_rng_u32(om_rng* r) {
  if (r->pos >= 624) om_rng_twist(r);
  uint32_t y = r->mt[r->pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

double om_rng_double(om_rng* r) {
  uint32_t a = om_rng_u32(r) >> 5, b = om_rng_u32(r) >> 6;
  return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

double om_rng_uniform(om_rng* r, double lo, double hi) { return lo + (hi - lo) * om_rng_double(r); }
