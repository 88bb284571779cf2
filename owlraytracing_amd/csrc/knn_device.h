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

// Rows whose order depends on how bit-identical distances are ordered (KList::has_ties below) are
// flagged by whichever kernel finishes them and redone by tie_fix_kernel (trueknn_team.hip):
// tie[slot] = 1 + the level the query finished at (bit 7: see `edge` below), counters[kTieCounter] counts them, and the first
// kTieListCap slots are also listed so that the usual handful needs no compaction pass.
// counters[kTieCounter + 1]: tie_fix_kernel's work cursor, [kTieCounter + 2]: rows it had to leave, [kTieCounter + 3]: rows that
// stood after a look at what is written (see knn_flag_tie).
constexpr int kTieCounter = 32;  // no kernel's own counter reset reaches this far
constexpr int kCounters = 40;
constexpr int kTieListCap = 4096;
// End-of-wave / end-of-workgroup statistics of the TrueKNN kernels live in STRIPES behind RT-DBSCAN's words of the counter array
// (Engine::kStatBase): kStatStripes stripes of kStatStride words, a cache line each, a workgroup adds to the stripe of its index
// and the host folds them.  All on one line they were a third of the lane kernel's time (10 M points, k = 10: 44.9 ms with, 29.7
// without its seven atomics per wave, 1.1 M a launch at ~12 ns each) and 1.5 % of the packet kernel's.
constexpr int kStatStripes = 32, kStatStride = 16, kStatBase = kCounters + 32 * 8 + 8 * 32;
// edge: one of the tied candidates may be the best one LEFT OUT of the row (or the kernel cannot tell): only a walk finds it.
// Without it every tie lies between two written entries, and tie_fix_kernel first looks whether the pairs' candidates became
// candidates at the same level (coincident points -- duplicates of a data set -- always do): then the row stands as it is.
// edge = 2: bit 7 of tie[slot] says so already (the packet kernel's SELECT pass sets it where it sees such a tie, long before
// the row is finished: team_pass) -- a byte with bit 7 alone is not a flag.
__device__ __forceinline__ void knn_note_tie_edge(uint8_t *tie, int32_t slot) {
  atomicOr(reinterpret_cast<unsigned int *>(tie + (slot & ~3)), 0x80u << (8 * (slot & 3)));
}
__device__ __forceinline__ void knn_flag_tie(uint8_t *tie, int32_t *tie_list, unsigned long long *counters, int32_t slot, int level,
                                             int edge = 1) {
  if (level >= 127) return;  // (tknnSolve caps max_rounds at 127)
  if (edge == 2)  // (an atomic on the byte's word: the note came from another lane, through memory)
    atomicOr(reinterpret_cast<unsigned int *>(tie + (slot & ~3)), (unsigned int)(1 + level) << (8 * (slot & 3)));
  else
    tie[slot] = (uint8_t)((1 + level) | (edge ? 0x80 : 0));
  const unsigned long long pos = atomicAdd(&counters[kTieCounter], 1ull);
  if (pos < (unsigned long long)kTieListCap) tie_list[pos] = slot;
}

template <int K>
struct KList {
  uint64_t key[K];
  // smallest distance (bits; distances are >= 0, so bits order like values) among the candidates this
  // list had no room for: equals the last entry's distance iff a candidate tied with it stayed out
  uint32_t left_out;
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < K; j++) key[j] = KNN_EMPTY_KEY;
    left_out = 0xffffffffu;
  }
  __device__ __forceinline__ uint64_t worst() const { return key[K - 1]; }
  // sorted insert, fully unrolled so the list stays in VGPRs (no dynamic indexing)
  __device__ __forceinline__ void insert(uint64_t c) {
    if (c < key[K - 1]) {
      const uint32_t out = (uint32_t)(key[K - 1] >> 32);
      left_out = out < left_out ? out : left_out;
#pragma unroll
      for (int j = K - 1; j > 0; --j) {
        bool shift = c < key[j - 1];
        key[j] = shift ? key[j - 1] : (c < key[j] ? c : key[j]);
      }
      key[0] = c < key[0] ? c : key[0];
    } else {
      const uint32_t out = (uint32_t)(c >> 32);
      left_out = out < left_out ? out : left_out;
    }
  }
  // Does the row of the first k entries depend on how bit-identical fp32 distances are ordered?
  // True if two of entries 0..k have the same distance (entry k: the best candidate left out of the
  // row).  The reference orders such keys by the round in which each was first a candidate
  // (deviceCode.cu:77-85 keeps what is listed), the lists here by index: flagged rows are redone by
  // tie_fix_kernel (trueknn_team.hip).  False positives are harmless.  Candidates the callers drop
  // at their gates are strictly farther than the k-th entry (knn_gate_from_worst) and never tie.
  __device__ __forceinline__ bool has_ties(int k) const {
    bool t = false;
#pragma unroll
    for (int j = 1; j < K; j++) t |= (j <= k) & ((uint32_t)(key[j] >> 32) == (uint32_t)(key[j - 1] >> 32));
    t |= (k >= K) & (left_out == (uint32_t)(key[K - 1] >> 32));
    return t;
  }
};

}  // namespace owlmi
