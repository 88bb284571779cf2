// knn_thresholds.h -- exact interval form of the reference's candidate test, shared by the wave
// and team kernels.
//
// The reference tests a query point q against the box a bounds program wrote for primitive c with
// radius r (samples/s01-trueknn/deviceCode.cu:38-56): fl(c - r) <= q <= fl(c + r) per axis.  For
// fixed (q, r) the set of fp32 values c that pass is an interval [thr_lo, thr_hi], because fp32
// rounding is monotone.  The endpoints are found once per query and radius level by galloping and
// bisecting on the ordered-integer image of fp32; afterwards a candidate costs compares only, and
// the result is bit-identical to the literal test (tests/test_trueknn_gpu.py::
// test_candidate_thresholds_equal_the_literal_box_test).  A per-ulp walk from the obvious guess
// q -+ r is NOT an option: near zero and across binades (q ~ r) the endpoint is 10^8 ulps away.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace owlmi {

// fp32 <-> uint32 keys that ascend with the float order (-inf .. -0, +0 .. +inf)
__device__ __forceinline__ uint32_t f_ord(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f_unord(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
#define OWLMI_ORD_NEG_INF 0x007fffffu /* f_ord(-inf) */
#define OWLMI_ORD_POS_INF 0xff800000u /* f_ord(+inf) */

// Smallest float c (as an ordered key) for which the monotone predicate P holds, searched from
// `guess`; clamped to [-inf, +inf]; OWLMI_ORD_POS_INF + 1 if P holds nowhere.
template <typename P>
__device__ __forceinline__ uint32_t first_true_key(P pred, float guess) {
  uint32_t u = f_ord(guess);
  u = u < OWLMI_ORD_NEG_INF ? OWLMI_ORD_NEG_INF : (u > OWLMI_ORD_POS_INF ? OWLMI_ORD_POS_INF : u);
  uint32_t lo, hi;  // at the end: pred(hi) true (or hi == +inf + 1), pred(lo) false (or lo == -inf)
  if (pred(f_unord(u))) {
    hi = u;
    uint32_t step = 1;
    for (;;) {
      uint32_t room = hi - OWLMI_ORD_NEG_INF;
      if (room == 0) return hi;
      uint32_t s = step < room ? step : room;
      uint32_t t = hi - s;
      if (pred(f_unord(t))) {
        if (t == OWLMI_ORD_NEG_INF) return t;
        hi = t;
        step <<= 1;
      } else {
        lo = t;
        break;
      }
    }
  } else {
    lo = u;
    uint32_t step = 1;
    for (;;) {
      uint32_t room = OWLMI_ORD_POS_INF - lo;
      if (room == 0) return OWLMI_ORD_POS_INF + 1u;
      uint32_t s = step < room ? step : room;
      uint32_t t = lo + s;
      if (pred(f_unord(t))) {
        hi = t;
        break;
      }
      lo = t;
      step <<= 1;
    }
  }
  while (hi - lo > 1u) {
    uint32_t mid = lo + ((hi - lo) >> 1);
    if (pred(f_unord(mid)))
      hi = mid;
    else
      lo = mid;
  }
  return hi;
}

// smallest c with q <= fl(c + r)      (q <= fl(c + r)  <=>  c >= thr_lo(q, r))
__device__ __forceinline__ float thr_lo(float q, float r) {
#pragma clang fp contract(off)
  if (!(q == q)) return INFINITY;
  uint32_t key = first_true_key([=](float c) { return q <= c + r; }, q - r);
  return key > OWLMI_ORD_POS_INF ? INFINITY : f_unord(key);
}
// largest c with fl(c - r) <= q       (fl(c - r) <= q  <=>  c <= thr_hi(q, r))
__device__ __forceinline__ float thr_hi(float q, float r) {
#pragma clang fp contract(off)
  if (!(q == q)) return -INFINITY;
  uint32_t key = first_true_key([=](float c) { return !(c - r <= q); }, q + r);  // first c that fails
  if (key <= OWLMI_ORD_NEG_INF) return -INFINITY;
  return f_unord(key - 1u);
}

// Squared-distance gate for k-list candidates: every d2 whose IEEE-rounded sqrt can be <= w (the
// current k-th distance) passes.  The pre-image of one rounded sqrt value spans at most 3
// consecutive floats of d2; (1 + 2^-21) adds at least 4 ulps to fl(w*w).
__device__ __forceinline__ float knn_gate_from_worst(float w) {
#pragma clang fp contract(off)
  float w2 = w * w;
  return w2 * 1.00000048f;
}

}  // namespace owlmi
