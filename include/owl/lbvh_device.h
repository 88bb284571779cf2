/*
 * lbvh_device.h -- device-visible layout of the MI355X LBVH that stands in for the OptiX
 * acceleration structure (reference call sites: owl/UserGeomGroup.cpp:161-217 optixAccelBuild,
 * owl/include/owl/owl_device.h:150-174 optixTrace).  Shared by the engine kernels
 * (owlraytracing_amd/csrc) and by user device programs compiled against owl/owl_device.h.
 *
 * Tree: Karras radix tree over Morton-sorted primitives.  Internal node i (0 <= i < n-1) covers
 * the sorted range [min(i,other), max(i,other)] and splits it after position `split`:
 *   left  child covers [first, split]   -> internal node `split`   unless first == split (leaf)
 *   right child covers [split+1, last]  -> internal node `split+1` unless last == split+1 (leaf)
 * Node 0 is the root.  References to children / ropes use one int32:
 *   ref >= 0  internal node index;  ref < 0  leaf (sorted primitive ~ref);  LBVH_END  traversal done.
 * rope_node[i] / rope_leaf[p] = where a depth-first, left-first walk continues after skipping
 * the subtree (stackless traversal).
 */
#pragma once
#include <stdint.h>

#define LBVH_END ((int32_t)0x80000000)

struct LbvhNode { /* 32 bytes: two 16-byte loads */
  float lo[3];
  int32_t split;
  float hi[3];
  int32_t other;
};

struct LbvhPoint { /* 16 bytes: sorted point + index it had in the caller's buffer */
  float x, y, z;
  int32_t id;
};

struct LbvhBox { /* 24 bytes = owl::box3f, what a bounds program writes */
  float lo[3];
  float hi[3];
};

/* 64-ary box pyramid over fixed blocks of LBVH_BLOCK Morton-consecutive points ("wide" view of the
 * same sorted order, one node = the 64 children a wave tests in one step):
 *   level[0][b] = box of points [LBVH_BLOCK*b, LBVH_BLOCK*(b+1));  level[l][j] = box of
 *   level[l-1][64j .. 64j+63];  the top level has <= 64 entries. */
#define LBVH_BLOCK 16
#define LBVH_WIDE_LEVELS 6
struct LbvhWideView {
  const LbvhBox *level[LBVH_WIDE_LEVELS];
  int32_t count[LBVH_WIDE_LEVELS];
  int32_t levels;
};

struct LbvhView {
  const LbvhNode *nodes;    /* n-1 */
  const int32_t *rope_node; /* n-1 */
  const int32_t *rope_leaf; /* n */
  const LbvhPoint *points;  /* n, Morton order (point sets)            | one of these two */
  const LbvhBox *boxes;     /* n, Morton order (general primitive AABBs) | is non-null       */
  const int32_t *prim_id;   /* n: caller's primitive index of sorted slot */
  int32_t n;
  int32_t root; /* 0, or ~0 when n == 1 */
  const int32_t *nan_count; /* device: how many of the LAST sorted points have a NaN coordinate (point sets) */
};

#if defined(__HIPCC__)
__device__ __forceinline__ int32_t lbvh_first(int32_t i, int32_t other) { return i < other ? i : other; }
__device__ __forceinline__ int32_t lbvh_last(int32_t i, int32_t other) { return i < other ? other : i; }
__device__ __forceinline__ int32_t lbvh_left_ref(int32_t i, const LbvhNode &nd) {
  return lbvh_first(i, nd.other) == nd.split ? ~nd.split : nd.split;
}
__device__ __forceinline__ int32_t lbvh_right_ref(int32_t i, const LbvhNode &nd) {
  return lbvh_last(i, nd.other) == nd.split + 1 ? ~(nd.split + 1) : nd.split + 1;
}
#endif
