// knn_device.h -- device helpers shared by the TrueKNN kernels: the distance arithmetic, the
// candidate box test and the register-resident k-list.  Each mirrors a line range of the
// reference's intersection program (samples/s01-trueknn/deviceCode.cu) and the decisions recorded
// in oracle/trueknn_oracle.c (read as documentation only; nothing from oracle/ is compiled here).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace owlmi {

// deviceCode.cu:110-113 exactly as written: (dx*dx + dy*dy) + dz*dz, every operation rounded to
// fp32 (decision (3): no contraction -- which products a compiler fuses is not portable), then a
// correctly rounded sqrt.  Contraction is pinned off inside the function whatever -ffp-contract
// says; __builtin_sqrtf is the IEEE-rounded sqrt (HIP's __fsqrt_rn is the approximate v_sqrt_f32
// unless OCML_BASIC_ROUNDED_OPERATIONS is set).
__device__ __forceinline__ float knn_dist2(float cx, float cy, float cz, float ox, float oy, float oz) {
#pragma clang fp contract(off)
  float x = cx - ox, y = cy - oy, z = cz - oz;
  return ((x * x) + (y * y)) + (z * z);
}
__device__ __forceinline__ float knn_sqrt(float d2) { return __builtin_sqrtf(d2); }

// deviceCode.cu:38-56: box of primitive c with radius r, tested against point q (closed box on
// the fp32 values the bounds program writes; r > 0 so lower = c - r, upper = c + r)
__device__ __forceinline__ bool knn_in_box(float cx, float cy, float cz, float r, float qx, float qy, float qz) {
  return (cx - r <= qx) & (qx <= cx + r) & (cy - r <= qy) & (qy <= cy + r) & (cz - r <= qz) & (qz <= cz + r);
}

// A k-list entry is one 64-bit key: fp32 bits of the (non-negative) distance in the high word,
// primitive index in the low word, so unsigned key order == (dist, index) order: the order the
// reference's strict '<' insertion (deviceCode.cu:116-134) produces when candidates arrive by
// ascending index.
__device__ __forceinline__ uint64_t knn_key(float dist, int32_t prim) {
  return ((uint64_t)__float_as_uint(dist) << 32) | (uint32_t)prim;
}
// hostCode.cpp:127-130 initial slot {ind=-1, dist=FLOAT_MAX}; low word 0 so that a candidate at
// distance exactly FLT_MAX is rejected like the reference's 'distance < maxDist'
#define KNN_EMPTY_KEY 0x7f7fffff00000000ull
__device__ __forceinline__ float knn_key_dist(uint64_t key) { return __uint_as_float((uint32_t)(key >> 32)); }
__device__ __forceinline__ int32_t knn_key_prim(uint64_t key) {
  return key == KNN_EMPTY_KEY ? -1 : (int32_t)(uint32_t)key;
}

template <int K>
struct KList {
  uint64_t key[K];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < K; j++) key[j] = KNN_EMPTY_KEY;
  }
  __device__ __forceinline__ uint64_t worst() const { return key[K - 1]; }
  // sorted insert, fully unrolled so the list stays in VGPRs (no dynamic indexing)
  __device__ __forceinline__ void insert(uint64_t c) {
    if (c < key[K - 1]) {
#pragma unroll
      for (int j = K - 1; j > 0; --j) {
        bool shift = c < key[j - 1];
        key[j] = shift ? key[j - 1] : (c < key[j] ? c : key[j]);
      }
      key[0] = c < key[0] ? c : key[0];
    }
  }
};

}  // namespace owlmi
