// random.h -- owl::common::LCG<N>: TEA-seeded 32-bit linear congruential generator returning
// floats in [0,1); the generator the OWL samples typedef as `Random`.
#pragma once
#include "owl/common/math/vec.h"

namespace owl {
namespace common {

template <unsigned int N = 4>
struct LCG {
  uint32_t state;
  inline __both__ LCG() {}  // deliberately uninitialised: usable inside per-ray data
  inline __both__ LCG(unsigned a, unsigned b) { init(a, b); }
  inline __both__ LCG(const vec2i &seed) { init((unsigned)seed.x, (unsigned)seed.y); }
  inline __both__ LCG(const vec2ui &seed) { init(seed.x, seed.y); }
  // N rounds of the Tiny Encryption Algorithm mix the two seed words
  inline __both__ void init(unsigned a, unsigned b) {
    unsigned sum = 0;
    for (unsigned i = 0; i < N; i++) {
      sum += 0x9e3779b9u;
      a += ((b << 4) + 0xa341316cu) ^ (b + sum) ^ ((b >> 5) + 0xc8013ea4u);
      b += ((a << 4) + 0xad90777du) ^ (a + sum) ^ ((a >> 5) + 0x7e95761eu);
    }
    state = a;
  }
  inline __both__ float operator()() {
    state = 1664525u * state + 1013904223u;  // Numerical Recipes LCG constants
    return ldexpf((float)state, -32);
  }
};

}  // namespace common
}  // namespace owl
