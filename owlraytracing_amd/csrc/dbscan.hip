// dbscan.hip -- RT-DBSCAN on the MI355X LBVH.
//
// The reference tree has no RT-DBSCAN source (samples/s02-rtdbscan is absent); its README only
// says distance computations go to the RT cores and "other clustering operations" to shader
// cores.  The spec implemented here is written down in oracle/dbscan_oracle.c (neighbourhood
// includes the point, fp32 distance arithmetic of the TrueKNN intersection program, clusters
// numbered by ascending smallest core index, border -> lowest adjacent cluster) and equals
// sklearn.cluster.DBSCAN wherever no pair sits within rounding of eps.
//
// Same machinery as TrueKNN: points are primitives with box c +- eps (deviceCode.cu:38-56
// pattern), every point is also a query, the "intersection program" does the true sphere test.
// Three traversal launches over the point LBVH (one query per lane, stackless ropes):
//   1. core flags: count neighbours, stop at minPts unless counts were asked for
//   2. union: every core point unites with each core neighbour of smaller index (lock-free
//      union-find; the smaller index stays root, so a root is its cluster's smallest core index)
//   3. labels: roots ranked by an exclusive scan; border points take the smallest adjacent root
// TIGHT NODES keep step 2 from costing O(n x neighbours) in dense sets (BASELINE config 3: 5 000
// neighbours per point).  A tree node whose box diagonal is below eps holds points that are pairwise
// within eps, so its core points are one cluster: every core point unites with the first core point
// of the first tight node on its own root path, and a traversal that meets a tight node settles it
// as a whole -- farthest corner within eps: unite with that representative; nearest face beyond
// eps: nothing; otherwise test its core points one by one until the first within eps.  The
// components, and so the labels, are those of the full neighbour graph.  Step 3 and the full
// neighbour counts use the same node tests (a node inside the sphere is counted, not walked).
// Union-find reads/writes go through agent-scope atomics: a workgroup's L1 (and another XCD's L2)
// would otherwise keep serving a stale parent and a failed CAS could retry forever.
#include "trueknn_engine.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include <cstring>

#ifndef TKNN_DIAG_BUILD
#define TKNN_DIAG_BUILD 0  // make DIAG=1 -> libowl_mi355x_diag.so: TKNN_DB_DIAG switches parts of the union pass off (times only)
#endif

namespace owlmi {
namespace {

constexpr int kDbBlock = 256;

struct DbArgs {
  LbvhView bvh;
  float eps;
  float eps_wide;  // eps * (1 + 1e-6): the box prefilter must not cut what the rounded sphere test accepts
  int min_pts;
  int want_counts;
  int keep_core;         // tknnDbscanAuto: slots already flagged core by a round with a smaller eps stay core (N(p) only grows)
  uint8_t *core_sorted;  // per sorted slot
  uint8_t *core;         // per caller index (may be null)
  int32_t *counts;       // per caller index (may be null)
  int32_t *parent;       // union-find over SORTED SLOTS (the smaller slot stays root): neighbours in space are neighbours in memory
  int32_t *min_row;      // per slot, meaningful at roots: the smallest row (caller index) among the cluster's core points
  int32_t *rank;         // per caller index: cluster label of the cluster whose smallest core row it is
  int32_t *labels;       // per caller index
  const int32_t *next_core;  // per sorted slot (+1 sentinel): first core slot at or after it, n if none
  float eps_in2, eps_out2;   // eps^2 (1 -+ 1e-5): below / above these, fp32 distance arithmetic cannot disagree
  // work counters, per traversal kernel k (0 core flags, 1 unions, 2 labels / assign): [2k] tree nodes
  // tested, [2k + 1] points whose distance to a query was computed (12 algorithmic bytes each)
  unsigned long long *stats;
  int32_t *group_of;  // per sorted slot, written by the core-flag kernel (see db_group_kernel); null: not wanted
  int32_t *near_node;  // per sorted slot, or null: the node a few levels above the point's group (written by the core-flag kernel for the noise probe)
  int short_way;  // db_group_union_kernel's settle: 0: the long way only (measurements)
  int chunk;  // packets per chunk dealt to an XCD (db_group_union_kernel)
  int scan_budget;  // steps the quick scan of a probe may take before the subtrees are asked (db_group_union_kernel)
  float reach, near_lo2, near_hi2;  // this pass of db_group_union_kernel: groups whose nearest faces are near_lo2 < d^2 <= near_hi2 apart, reach >= sqrt(near_hi2)
  // second pass of db_group_union_kernel: uni[node] >= 0: every core point under the node is in the set that slot was the root
  // of when the first pass had ended (db_uniform_kernel); negative: not known to be one set
  const int32_t *block_paths;  // Lbvh::block_paths_device(), or null: where a slot's walk down to its group may start
  const int32_t *split_owner;  // Lbvh::split_owner_device()
  int32_t *uni;       // per internal node
  int32_t *uni_leaf;  // per sorted slot, written for the listed groups that are single points only
  int32_t *pk_diag;  // diagnostic library, TKNN_DB_DIAG & 512: per packet of the timed pass: ticks, extent (1e-6), rounds, settles
  unsigned long long *diag_out;  // diagnostic library, TKNN_DB_DIAG & 512: the group-union kernel's wave time by part ([0] loads [1] tests [2] settles [3] pushes [4] packet set-up, s_memtime ticks) and [5] rounds [6] settles [7] packets
  int diag;  // TKNN_DB_DIAG, diagnostic library only (results are wrong when set): 1 = no probes, 2 = no unions between groups, 4 = none inside groups, 8 = walk lengths on stderr, 16 = report a stack overflow (the result is right: the call falls back), 32 = probes by scanning only (the result is right)
};

// a workgroup's counts into the kernel's two counters: wave sums, one LDS atomic per wave, one global
// atomic per workgroup (every thread of the workgroup must call it).  `stats` = a.stats + 2 * (which kernel): the counters
// are kept in 32 stripes of 8 words, a workgroup adds to the stripe of its index, the host sums them.
constexpr int kDbStripes = 32;
// the group-union kernel's per-XCD packet cursors: a cache line each (side by side their atomics -- one per packet, its answer
// awaited, and up to eight per wave that finds the lists used up -- queue at one place: trueknn_team.hip, kXcdCounter)
constexpr int kDbCursorStride = 32;
__device__ __forceinline__ void db_add_stats(unsigned long long *stats, unsigned long long *blk, uint32_t nodes, uint32_t points) {
  stats += (blockIdx.x & (kDbStripes - 1)) * 8;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nodes += __shfl_xor(nodes, off);
    points += __shfl_xor(points, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&blk[0], (unsigned long long)nodes);
    atomicAdd(&blk[1], (unsigned long long)points);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], blk[0]);
    atomicAdd(&stats[1], blk[1]);
  }
}

// the same for a kernel whose waves are independent and whose LDS is counted in waves per CU: one global atomic per wave
__device__ __forceinline__ void db_add_stats_wave(unsigned long long *stats, uint32_t nodes, uint32_t points) {
  stats += ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kDbStripes - 1)) * 8;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nodes += __shfl_xor(nodes, off);
    points += __shfl_xor(points, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&stats[0], (unsigned long long)nodes);
    atomicAdd(&stats[1], (unsigned long long)points);
  }
}

// squared distances from q to the farthest and the nearest point of a box
__device__ __forceinline__ void box_dist2(const LbvhNode &nd, const LbvhPoint &q, float &far2, float &near2) {
  const float ax = fmaxf(fabsf(q.x - nd.lo[0]), fabsf(q.x - nd.hi[0])), ay = fmaxf(fabsf(q.y - nd.lo[1]), fabsf(q.y - nd.hi[1])),
              az = fmaxf(fabsf(q.z - nd.lo[2]), fabsf(q.z - nd.hi[2]));
  const float bx = fmaxf(fmaxf(nd.lo[0] - q.x, q.x - nd.hi[0]), 0.f), by = fmaxf(fmaxf(nd.lo[1] - q.y, q.y - nd.hi[1]), 0.f),
              bz = fmaxf(fmaxf(nd.lo[2] - q.z, q.z - nd.hi[2]), 0.f);
  far2 = ax * ax + ay * ay + az * az;
  near2 = bx * bx + by * by + bz * bz;
}
// per-axis reach of a union pass's box prefilter from the pass's bound on the squared distance of nearest faces: at
// least its square root (sqrtf is correctly rounded; two parts in 10^6 cover that rounding and the squares' in near2)
inline float db_reach_of(float near_hi2) { return sqrtf(near_hi2) * 1.000002f; }
__device__ __forceinline__ bool node_is_tight(const LbvhNode &nd, float eps_in2) {
  const float ex = nd.hi[0] - nd.lo[0], ey = nd.hi[1] - nd.lo[1], ez = nd.hi[2] - nd.lo[2];
  return ex * ex + ey * ey + ez * ez <= eps_in2;  // NaN boxes (none: fit ignores NaN points) would be "not tight"
}

__device__ __forceinline__ int32_t uf_load(int32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int32_t uf_find(int32_t *parent, int32_t x) {
  for (;;) {
    int32_t p = uf_load(parent + x);
    if (p == x) return x;
    int32_t g = uf_load(parent + p);
    if (g == p) return p;  // p is a root
    __hip_atomic_store(parent + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path halving
    x = g;
  }
}
__device__ __forceinline__ void uf_unite(int32_t *parent, int32_t a, int32_t b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    const int32_t lo = a < b ? a : b, hi = a < b ? b : a;
    // hook the larger root under the smaller; fails if `hi` stopped being a root meanwhile
    if (atomicCAS(parent + hi, hi, lo) == hi) return;
  }
}

// Walks the tree for the CORE neighbours of q, tight nodes settled as a whole: calls
// f(representative's slot) for every tight node that has a core point within eps of q (its first core
// point stands for all of them) and for every core point within eps reached as a leaf.  `own_slot`
// (or -1) names q's own sorted slot: the tight node holding it is skipped.
// `settled(slot)` may say that the group a tight node's first core point stands for needs no look
// (the union kernel: already in my set), sparing the distance tests and the probe.
template <typename S, typename F>
__device__ __forceinline__ void for_each_core_group(const DbArgs &a, const LbvhPoint &q, int32_t own_slot, S settled, F f,
                                                    uint32_t &node_tests, uint32_t &point_tests) {
  const LbvhView &bvh = a.bvh;
  const float r = a.eps_wide;
  int32_t ref = bvh.root;
  // A tight node the sphere cuts through is PROBED, not scanned: the walk goes on below it (its
  // descendants are tight too: inside -> hit, outside -> skipped) and leaves the subtree through the
  // node's own rope at the first core point found within eps -- one is enough, the node is one group.
  int32_t probe_rep = -1, probe_exit = LBVH_END;
  while (ref != LBVH_END) {
    if (probe_rep >= 0 && ref == probe_exit) probe_rep = -1;  // left the probed subtree without a hit
    if (ref >= 0) {
      const LbvhNode nd = bvh.nodes[ref];
      const int32_t rope = bvh.rope_node[ref];  // with the node, not after it: half the chain of dependent loads
      node_tests++;
      const bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) & (q.y <= nd.hi[1] + r) &
                       (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
      if (!hit) {
        ref = rope;
        continue;
      }
      const bool probing = probe_rep >= 0;
      if (probing || node_is_tight(nd, a.eps_in2)) {
        const int32_t first = lbvh_first(ref, nd.other), last = lbvh_last(ref, nd.other);
        const int32_t s = a.next_core[first];
        if (s > last || (first <= own_slot && own_slot <= last)) {  // no core point here / my own group
          ref = rope;
          continue;
        }
        if (!probing && settled(s)) {
          ref = rope;
          continue;
        }
        float far2, near2;
        box_dist2(nd, q, far2, near2);
        if (far2 <= a.eps_in2) {  // every point of the node is within eps
          f(probing ? probe_rep : s);
          ref = probing ? probe_exit : rope;
          probe_rep = -1;
          continue;
        }
        if (near2 > a.eps_out2) {  // none is
          ref = rope;
          continue;
        }
        if (!probing) {
          probe_rep = s;
          probe_exit = rope;
        }
      }
      ref = lbvh_left_ref(ref, nd);
    } else {
      const int32_t slot = ~ref;
      const uint8_t is_core = a.core_sorted[slot];  // three independent loads at once (nearly every leaf reached is core)
      const LbvhPoint p = bvh.points[slot];
      const int32_t rope = bvh.rope_leaf[slot];
      if (is_core && slot != own_slot) {
        point_tests++;
        if (knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z)) <= a.eps) {
          if (probe_rep >= 0) {
            f(probe_rep);
            ref = probe_exit;
            probe_rep = -1;
            continue;
          }
          f(slot);
        }
      }
      ref = rope;
    }
  }
}

// Core flags.  Every point first looks at its GROUP, the first tight node on its own root path (db_group_kernel): its
// points are pairwise within eps, so if it holds minPts of them the point is core without looking any further.  (Listing the
// points that do have to look and walking for them with full waves, as the label pass does, was tried: on BASELINE config 3
// they are a fifth of all points, a full wave waits for the longest of 64 walks instead of the longest of a dozen, and the
// pass took 2.2 ms instead of 1.45.)
__device__ __forceinline__ void db_core_body(const DbArgs &a, int32_t t, uint32_t &node_tests, uint32_t &point_tests) {
  const LbvhView &bvh = a.bvh;
  if (a.parent) a.parent[t] = t;
  if (a.keep_core && a.core_sorted[t]) return;
  const LbvhPoint q = bvh.points[t];
  int32_t cnt = 0;
  const int stop_at = a.want_counts ? 0x7fffffff : a.min_pts;
  const int32_t clean_end = bvh.n - (bvh.nan_count ? *bvh.nan_count : 0);  // NaN points sort last
  const float r = a.eps_wide;
  int32_t ref = bvh.root;
  // the last nodes above my group on my root path (anc[0]: kNear levels above it).  A point whose group is too small to
  // decide looks for its minPts neighbours there FIRST: they are next to it, a walk from the root spends two dozen steps
  // getting near (BASELINE config 3: a fifth of the points walk, nearly all of them core, 0.9 of the pass's 1.4 ms).
  constexpr int kNear = 4;
  static_assert(kNear == LBVH_PATH_WORDS - 1, "the block paths hold the ring's ancestors");
  int32_t anc[kNear];
#pragma unroll
  for (int j = 0; j < kNear; j++) anc[j] = bvh.root;
  if (!a.want_counts || a.group_of) {
    int32_t node = bvh.root, first = t;
    if (a.block_paths) {
      // Not from the root: from the deepest node that holds the wave's 64 slots (a table of the tree, LBVH_PATH_BLOCK), with
      // that node's four nearest ancestors in the ring -- if it is not tight none of its ancestors is (a parent's box holds
      // the child's), and the walk from the root would have come through here with exactly this ring (config 3: 16 of a
      // point's 23 steps).  If it is tight, the group is that node or one of the ancestors: the highest tight one among the
      // four the table has -- unless the fourth is tight too, then the walk starts at the root after all.
      const int32_t *bp = a.block_paths + (size_t)(t / LBVH_PATH_BLOCK) * LBVH_PATH_WORDS;
      const int32_t a4 = bp[0], a3 = bp[1], a2 = bp[2], a1 = bp[3], lca = bp[4];
      if (!node_is_tight(bvh.nodes[lca], a.eps_in2)) {
        node = lca;
        anc[0] = a4, anc[1] = a3, anc[2] = a2, anc[3] = a1;
      } else {
        const bool t1 = node_is_tight(bvh.nodes[a1], a.eps_in2), t2 = node_is_tight(bvh.nodes[a2], a.eps_in2),
                   t3 = node_is_tight(bvh.nodes[a3], a.eps_in2), t4 = node_is_tight(bvh.nodes[a4], a.eps_in2);
        node_tests += 4;
        if (!t4) node = t3 ? a3 : t2 ? a2 : t1 ? a1 : lca;  // (the ring stays at the root: no subtree to count in first)
      }
      node_tests++;
    }
    if (TKNN_DIAG_BUILD && (a.diag & 128)) node = -1;  // (times only) no descent to the group
    while (node >= 0) {
      const LbvhNode nd = bvh.nodes[node];
      node_tests++;
      if (node_is_tight(nd, a.eps_in2)) {
        first = lbvh_first(node, nd.other);
        const int32_t last = lbvh_last(node, nd.other);
        if (!a.want_counts && last < clean_end && last - first + 1 >= a.min_pts) {
          cnt = last - first + 1;
          ref = LBVH_END;
        }
        break;
      }
#pragma unroll
      for (int j = 0; j + 1 < kNear; j++) anc[j] = anc[j + 1];
      anc[kNear - 1] = node;
      node = t <= nd.split ? lbvh_left_ref(node, nd) : lbvh_right_ref(node, nd);
    }
    // the group's first slot keeps the group's reference (a node, or ~t for a point by itself), the others ~first
    if (a.group_of) a.group_of[t] = t == first ? (node >= 0 ? node : ~t) : ~first;
    if (a.near_node) a.near_node[t] = anc[0];
  }
  if (TKNN_DIAG_BUILD && (a.diag & 64)) ref = LBVH_END;  // (times only) no neighbour count
  // rope walk from `from` until the walk would leave through `until` (the rope of the subtree's root; LBVH_END: the whole tree)
  auto count_from = [&](int32_t from, int32_t until, int32_t skip, int32_t skip_rope) {
    int32_t at = from;
    while (at != until && cnt < stop_at) {
      if (at == skip) {  // the subtree counted already
        at = skip_rope;
        continue;
      }
      if (at >= 0) {
        // the rope is fetched WITH the node, not after the box test has asked for it: a walk is a chain of dependent loads
        // (a wave lives as long as its longest walk), and this halves the chain for four more bytes per step
        const LbvhNode nd = bvh.nodes[at];
        const int32_t rope = bvh.rope_node[at];
        node_tests++;
        const bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) & (q.y <= nd.hi[1] + r) &
                         (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
        if (hit) {
          // a node inside the sphere is counted, not walked
          float far2, near2;
          box_dist2(nd, q, far2, near2);
          const int32_t first = lbvh_first(at, nd.other), last = lbvh_last(at, nd.other);
          // (a node that holds my own slot while a subtree is being stepped over is an ANCESTOR of that subtree -- the subtree
          // itself is stepped over before it gets here, its descendants are never reached --: its points were counted in part
          // already, so it is walked, not added whole)
          const bool over_skipped = skip != LBVH_END && first <= t && t <= last;
          if (far2 <= a.eps_in2 && last < clean_end && !over_skipped) {
            cnt += last - first + 1;
            at = rope;
            continue;
          }
        }
        at = hit ? lbvh_left_ref(at, nd) : rope;
      } else {
        const int32_t slot = ~at;
        const LbvhPoint p = bvh.points[slot];
        point_tests++;
        if (knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z)) <= a.eps) cnt++;
        at = bvh.rope_leaf[slot];
      }
    }
  };
  if (ref != LBVH_END) {
    const int32_t near = anc[0];
    if (!a.want_counts && near != bvh.root && near >= 0) {
      // enough neighbours in the subtree around me: core, whatever else the sphere holds.  Not enough: the count goes on
      // over the rest of the tree (the walk from the root steps over that subtree).
      const int32_t near_rope = bvh.rope_node[near];
      count_from(near, near_rope, LBVH_END, LBVH_END);
      if (cnt < a.min_pts) count_from(bvh.root, LBVH_END, near, near_rope);  // the rest of the tree
    } else {
      count_from(bvh.root, LBVH_END, LBVH_END, LBVH_END);
    }
  }
  const uint8_t is_core = cnt >= a.min_pts;
  a.core_sorted[t] = is_core;
  // results are indexed by ROW (the point's position in the caller's buffer, prim_id of the sorted
  // slot), not by the id an engine built with tknnBuildIds reports; the union-find by sorted slot
  const int32_t row = bvh.prim_id[t];
  if (a.core && !(TKNN_DIAG_BUILD && (a.diag & 256))) a.core[row] = is_core;
  if (a.counts) a.counts[row] = cnt;
}

// (`block_count`, or null: the workgroup's number of core slots, which the kernels behind this one place their lists by --
// a launch of its own over all the flags otherwise, db_flag_count_kernel)
__global__ void __launch_bounds__(kDbBlock) db_core_kernel(DbArgs a, int32_t *block_count) {
  __shared__ unsigned long long blk_stats[2];
  __shared__ int32_t wave_count[kDbBlock / 64];
  if (threadIdx.x < 2) blk_stats[threadIdx.x] = 0ull;
  __syncthreads();
  uint32_t node_tests = 0, point_tests = 0;
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n) db_core_body(a, t, node_tests, point_tests);
  if (block_count) {  // (a lane reads the flag it has just written)
    const unsigned long long m = __ballot(t < a.bvh.n && a.core_sorted[t] != 0);
    if ((threadIdx.x & 63) == 0) wave_count[threadIdx.x >> 6] = __popcll(m);
  }
  db_add_stats(a.stats + 0, blk_stats, node_tests, point_tests);  // (a barrier inside)
  if (block_count && threadIdx.x == 0) {
    int32_t total = 0;
    for (int w = 0; w < kDbBlock / 64; w++) total += wave_count[w];
    block_count[blockIdx.x] = total;
  }
}

// next_core[s] = first core slot >= s (n if none): with rank[s] = number of core slots before s (an
// exclusive sum of the flags) and pos[r] = slot of the r-th core point, next_core[s] = pos[rank[s]]
// (the flags are summed as they are read: one byte per slot in, one word out)
struct DbFlagOf {
  __host__ __device__ int32_t operator()(uint8_t c) const { return c; }
};
typedef hipcub::TransformInputIterator<int32_t, DbFlagOf, const uint8_t *> DbFlagIter;
// (and, if asked: the slots that are NOT core, listed in slot order for the label pass -- slot t is the (t - rank[t])-th of
// them, no atomics -- with their number)
__global__ void __launch_bounds__(kDbBlock) db_core_pos_kernel(DbArgs a, const int32_t *rank, int32_t *pos, int32_t *others,
                                                               unsigned long long *n_others) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t >= a.bvh.n) return;
  const int32_t r = rank[t];
  const bool is_core = a.core_sorted[t] != 0;
  if (is_core)
    pos[r] = t;
  else if (others)
    others[t - r] = t;
  if (t == a.bvh.n - 1) {
    const int32_t n_core = r + (is_core ? 1 : 0);
    pos[n_core] = 0x7f7f7f7f;  // "none" (clamped by db_next_core_kernel): the one place a slot behind the last core point looks at
    if (others) *n_others = (unsigned long long)(a.bvh.n - n_core);
  }
}
// The same three steps without a rank per slot (tknnDbscan; a library sum over the n flags took 0.05 ms and a 40 MB array
// written once and read twice): a count of the core flags per workgroup of 256 slots, a sum over those n / 256 counts, and
// the two kernels below find a slot's rank as its workgroup's place + the core slots before it in the workgroup.
__device__ __forceinline__ int32_t db_rank_in_block(bool flag, int32_t block_place, int32_t *wave_count, int32_t &block_total) {
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_count[wave] = __popcll(m);
  __syncthreads();
  int32_t r = block_place;
  block_total = 0;
  for (int w = 0; w < kDbBlock / 64; w++) {
    if (w < wave) r += wave_count[w];
    block_total += wave_count[w];
  }
  return r + __popcll(m & ((1ull << lane) - 1ull));
}
__global__ void __launch_bounds__(kDbBlock) db_flag_count_kernel(DbArgs a, int32_t *block_count) {
  __shared__ int32_t wave_count[kDbBlock / 64];
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  int32_t total;
  (void)db_rank_in_block(t < a.bvh.n && a.core_sorted[t] != 0, 0, wave_count, total);
  if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kDbBlock) db_core_pos_blocks_kernel(DbArgs a, const int32_t *block_place, int32_t *pos, int32_t *others,
                                                                      unsigned long long *n_others) {
  __shared__ int32_t wave_count[kDbBlock / 64];
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  const bool in = t < a.bvh.n, is_core = in && a.core_sorted[t] != 0;
  int32_t total;
  const int32_t r = db_rank_in_block(is_core, block_place[blockIdx.x], wave_count, total);
  if (!in) return;
  if (is_core)
    pos[r] = t;
  else if (others)
    others[t - r] = t;
  if (t == a.bvh.n - 1) {
    const int32_t n_core = r + (is_core ? 1 : 0);
    pos[n_core] = 0x7f7f7f7f;  // "none" (clamped below): the one place a slot behind the last core point looks at
    if (others) *n_others = (unsigned long long)(a.bvh.n - n_core);
  }
}
__global__ void __launch_bounds__(kDbBlock) db_next_core_blocks_kernel(DbArgs a, const int32_t *block_place, const int32_t *pos, int32_t *next_core) {
  __shared__ int32_t wave_count[kDbBlock / 64];
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  const bool in = t < a.bvh.n;
  int32_t total;
  const int32_t r = db_rank_in_block(in && a.core_sorted[t] != 0, block_place[blockIdx.x], wave_count, total);
  if (!in) return;
  const int32_t v = pos[r];  // a slot after the last core point has rank = number of core points: pos[] holds "none" there
  next_core[t] = v < a.bvh.n ? v : a.bvh.n;
  if (t == a.bvh.n - 1) next_core[a.bvh.n] = a.bvh.n;
}
__global__ void __launch_bounds__(kDbBlock) db_next_core_kernel(DbArgs a, const int32_t *rank, const int32_t *pos,
                                                                int32_t *next_core) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t > a.bvh.n) return;
  if (t == a.bvh.n) {
    next_core[t] = a.bvh.n;
    return;
  }
  // a slot after the last core point has rank = number of core points: pos[] holds n there
  const int32_t v = pos[rank[t]];
  next_core[t] = v < a.bvh.n ? v : a.bvh.n;
}

__device__ __forceinline__ void db_union_body(const DbArgs &a, int32_t t, unsigned long long *seen, int kPairs, uint32_t &node_tests,
                                              uint32_t &point_tests) {
  const LbvhView &bvh = a.bvh;
  const LbvhPoint q = bvh.points[t];
  // the first tight node on my own root path: its core points are one cluster, held together by its
  // first core point, which also stands for me in the unions below
  int32_t mine = t;
  int32_t node = bvh.root;
  while (node >= 0) {
    const LbvhNode nd = bvh.nodes[node];
    node_tests++;
    if (node_is_tight(nd, a.eps_in2)) {
      const int32_t s = a.next_core[lbvh_first(node, nd.other)];  // <= t: I am core and inside
      if (s != t) {
        mine = s;
        uf_unite(a.parent, t, mine);
      }
      break;
    }
    node = t <= nd.split ? lbvh_left_ref(node, nd) : lbvh_right_ref(node, nd);
  }
  auto slot_of = [&](int32_t other, unsigned long long &key) -> unsigned long long * {
    const uint32_t lo = (uint32_t)min(mine, other), hi = (uint32_t)max(mine, other);
    key = ((unsigned long long)hi << 32) | lo;
    const uint32_t h = (lo * 0x9e3779b1u) ^ (hi * 0x85ebca6bu);
    return seen + ((h >> 11) & (kPairs - 1));
  };
  int32_t my_root = uf_find(a.parent, mine);
  for_each_core_group(
      a, q, t,
      [&](int32_t other) -> bool {  // is that group in my set already?  (then there is nothing to find out)
        if (other == mine) return true;
        unsigned long long key;
        unsigned long long *slot = slot_of(other, key);
        if (*(volatile unsigned long long *)slot == key) return true;
        if (uf_find(a.parent, other) != my_root) {
          my_root = uf_find(a.parent, my_root);  // my root may have been hooked under another meanwhile
          if (uf_find(a.parent, other) != my_root) return false;
        }
        *(volatile unsigned long long *)slot = key;
        return true;
      },
      [&](int32_t other) {
        if (other == mine || other == t) return;
        unsigned long long key;
        unsigned long long *slot = slot_of(other, key);
        if (*(volatile unsigned long long *)slot == key) return;
        *(volatile unsigned long long *)slot = key;
        uf_unite(a.parent, mine, other);
        my_root = uf_find(a.parent, mine);
      },
      node_tests, point_tests);
}

__global__ void __launch_bounds__(kDbBlock) db_union_kernel(DbArgs a) {
  // The 256 Morton-consecutive points of a workgroup mostly share a tight node and meet the same
  // neighbouring tight nodes: every (my group, other group) pair would be united hundreds of times,
  // each a pair of union-find walks through global atomics.  A direct-mapped LDS table of the pairs
  // this workgroup has already taken care of drops the repeats (a stale or raced entry only costs a
  // redundant unite; the pair that set an entry is united by the lane that set it).
  constexpr int kPairs = 2048;
  __shared__ unsigned long long seen[kPairs];
  __shared__ unsigned long long blk_stats[2];
  for (int i = threadIdx.x; i < kPairs; i += kDbBlock) seen[i] = ~0ull;
  if (threadIdx.x < 2) blk_stats[threadIdx.x] = 0ull;
  __syncthreads();
  uint32_t node_tests = 0, point_tests = 0;
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n && a.core_sorted[t]) db_union_body(a, t, seen, kPairs, node_tests, point_tests);
  db_add_stats(a.stats + 2, blk_stats, node_tests, point_tests);
}

// ---- unions by GROUP: one walk per maximal tight node ------------------------------------------------------------------
// Every point lies in exactly one maximal tight node (a node whose box diagonal is below eps and whose parent's is not;
// a leaf under a non-tight parent is one by itself): a GROUP.  A group's core points are one cluster, so the clusters are
// the components of the graph of groups with an edge (A, B) iff some core point of A lies within eps of some core point
// of B.  The per-point walk above pays a tree descent per POINT to the same ~100 neighbouring groups every other point of
// its group reaches too (BASELINE config 3: 500 node tests per point, 40 of the call's 44 ms).  Here each group walks once,
// box against tree, and only over the slots AFTER its own (the rope of its node is where a depth-first walk continues
// behind it): a pair of groups is looked at exactly once, from its earlier end.
//   group kernel: every point finds its group on its own root path; core points unite with the group's first core point;
//                 the point at the group's first slot lists the group if it has a core point.
//   union kernel: one lane per listed group.  A later group B in reach: no box point within eps (nearest faces) -> nothing;
//                 same set already -> nothing; farthest corners within eps -> unite; otherwise PROBE: core points of A that
//                 are within eps of B's box against B's core points, by the spec's distance arithmetic, until the first hit.
__device__ __forceinline__ void box_box_dist2(const float *alo, const float *ahi, const float *blo, const float *bhi, float &far2,
                                              float &near2) {
  float f = 0.f, g = 0.f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float fa = fmaxf(fabsf(ahi[c] - blo[c]), fabsf(bhi[c] - alo[c]));
    const float ne = fmaxf(fmaxf(blo[c] - ahi[c], alo[c] - bhi[c]), 0.f);
    f += fa * fa;
    g += ne * ne;
  }
  far2 = f;
  near2 = g;
}

// per slot: the group's reference (node, or ~slot of a single point) at the group's first slot if the group has a core
// point, LBVH_END everywhere else -- compacted into the list, in slot order, in three steps: this kernel counts each
// workgroup's entries, an exclusive sum over those n / 256 counts gives the workgroups' places, db_group_list_kernel writes
// the entries (a library select over the n slots took 0.09 ms of a 3 ms call).  The core-flag
// kernel has left every point's group in group_of (it walks the point's root path anyway).
// (The kernel also fills, as it streams by, what later launches want filled: a fill is a launch of its own otherwise.)
__global__ void __launch_bounds__(kDbBlock) db_group_kernel(DbArgs a, int32_t *group_at, int32_t *fill_uni, int32_t *fill_min_row,
                                                            int32_t *block_count) {
  __shared__ int32_t wave_count[kDbBlock / 64];
  const LbvhView &bvh = a.bvh;
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  int32_t out = LBVH_END;
  if (t < bvh.n) {
    fill_uni[t] = -1;              // db_uniform_kernel: nobody has arrived
    fill_min_row[t] = 0x7f7f7f7f;  // db_flatten_kernel: above every row
    const int32_t g = a.group_of[t];
    const bool leads = g >= 0 || g == ~t;
    const int32_t first = leads ? t : ~g;
    const int32_t s = a.next_core[first];
    // s < t, both core, one group: united by a plain store -- these are the call's first unions, every slot is still its
    // own root, s stays one throughout this kernel (only later slots are hooked, under it), and parent[t] is written by
    // nobody else
    if (a.core_sorted[t] && s != t && !(a.diag & 4)) a.parent[t] = s;
    if (leads) {
      const int32_t last = g >= 0 ? lbvh_last(g, bvh.nodes[g].other) : t;
      if (s <= last) out = g;
    }
    group_at[t] = out;
  }
  const unsigned long long m = __ballot(out != LBVH_END);
  if ((threadIdx.x & 63) == 0) wave_count[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t c = 0;
    for (int w = 0; w < kDbBlock / 64; w++) c += wave_count[w];
    block_count[blockIdx.x] = c;
  }
}
__global__ void __launch_bounds__(kDbBlock) db_group_list_kernel(DbArgs a, const int32_t *group_at, const int32_t *block_place, int32_t *groups,
                                                                 unsigned long long *total) {
  __shared__ int32_t wave_count[kDbBlock / 64];
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  const int32_t g = t < a.bvh.n ? group_at[t] : LBVH_END;
  const unsigned long long m = __ballot(g != LBVH_END);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_count[wave] = __popcll(m);
  __syncthreads();
  int32_t place = block_place[blockIdx.x];
  for (int w = 0; w < wave; w++) place += wave_count[w];
  if (g != LBVH_END) groups[place + __popcll(m & ((1ull << lane) - 1ull))] = g;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kDbBlock - 1) *total = (unsigned long long)(place + __popcll(m));  // the list's length
}

// Between the two passes of the group-union kernel: which tree nodes hold core points of ONE set only?  After the first pass
// (pairs of groups that nearly touch) a cluster's dense part is one set, and the second pass, which walks the tree with the
// full reach, would still go down to every group around a packet only to find it in the packet's own set.  One lane per
// listed group climbs from the group towards the root: a parent whose other child has no core point inherits the set; at a
// parent with two children that have some, the first to arrive leaves its set there and stops, the second goes on if both
// sets are the same (a side that is not one set never arrives, and the parent stays "not known").  Roots do not move
// during this kernel (no unions), so two lanes of one set see the same root.
//   uni[node]: -1 nobody arrived; <= -2: one child arrived with set -2 - value; >= 0: one set, this root
__global__ void __launch_bounds__(kDbBlock) db_uniform_kernel(DbArgs a, const int32_t *groups, const unsigned long long *n_groups) {
  const LbvhView &bvh = a.bvh;
  const long long g = (long long)blockIdx.x * kDbBlock + threadIdx.x;
  if (g >= (long long)*n_groups) return;
  const int32_t G = groups[g];
  int32_t first, last, at = G;  // the subtree climbed so far: its slots, and its node (or ~slot)
  if (G >= 0) {
    const int32_t other = bvh.nodes[G].other;
    first = lbvh_first(G, other), last = lbvh_last(G, other);
  } else {
    first = last = ~G;
  }
  const int32_t set = uf_find(a.parent, a.next_core[first]);
  if (G >= 0)
    a.uni[G] = set;
  else
    a.uni_leaf[first] = set;
  const int32_t n = bvh.n;
  while (first != 0 || last != n - 1) {
    // parent: a left child ends where its parent splits, a right child begins right after
    int32_t up = -1;
    bool is_left = false;
    if (at >= 0) {
      is_left = at == last;
      up = a.split_owner[is_left ? at : at - 1];
    } else {
      if (first < n - 1) {
        const int32_t o = a.split_owner[first];
        if (lbvh_first(o, bvh.nodes[o].other) == first) up = o, is_left = true;
      }
      if (up < 0) up = a.split_owner[first - 1];
    }
    const LbvhNode nd = bvh.nodes[up];
    const int32_t up_first = lbvh_first(up, nd.other), up_last = lbvh_last(up, nd.other);
    const int32_t sib_first = is_left ? nd.split + 1 : up_first, sib_last = is_left ? up_last : nd.split;
    if (a.next_core[sib_first] <= sib_last) {  // the other child has core points: both must arrive, with one set
      const int32_t old = atomicExch(a.uni + up, -2 - set);
      if (old == -1 || -2 - old != set) break;
    }
    a.uni[up] = set;
    at = up, first = up_first, last = up_last;
  }
}

// Persistent waves, each working on one PACKET of 64 consecutive groups of the list at a time (Morton neighbours: they
// meet the same nodes).  The wave walks the tree ONCE for the packet, in two alternating roles:
//   lanes = nodes: up to 64 references are popped from the wave's LDS stack and their boxes (a leaf's is its point)
//                  loaded at once and put into LDS;
//   lanes = groups: every popped box, read from LDS by all lanes, is tested against each lane's own group widened by the
//                  reach.  A node no group reaches is dropped.  A tight node (the walk only descends through nodes that are
//                  not: it is maximal, a group) or a core leaf is collected, in the lane's own LDS column, by every group
//                  that has it in reach (nearest faces within eps) and lies before it; when a lane has kDbBuf waiting, the
//                  wave settles what all lanes have collected: first core slots, rows and parent pointers fetched for all
//                  at once, then one after the other.
//   lanes = nodes: the nodes some group reaches that are not tight push their children.
// (History, BASELINE config 3, union pass: a walk per POINT 40 ms; a walk per group, one lane each, 5 ms -- 320 node visits
// per group, each a 64-byte sector from memory because 8 000 waves at different places of a 320 MB tree share nothing in a
// 4 MB L2: 15 GB per launch; one rope walk per packet, wave-uniform, 1.8 ms per launch -- 1 300 dependent loads in a row;
// the stack walk needs some 40 rounds of loads per packet; with the popped nodes tested against the packet's bounding box
// instead of its 64 groups, a packet across a jump of the Z-curve walked half the tree: one wave, 4 ms.)
#ifndef TKNN_DB_BUF
#define TKNN_DB_BUF 8
#endif
constexpr int kDbBuf = TKNN_DB_BUF;
#ifndef TKNN_DB_BOXES
#define TKNN_DB_BOXES 2
#endif
constexpr int kDbBoxes = TKNN_DB_BOXES;  // bounding boxes per packet for the prefilter of popped nodes (64 / kDbBoxes lanes each)
#ifndef TKNN_DB_UNION_BLOCK
#define TKNN_DB_UNION_BLOCK 128  // threads per workgroup of the group-union kernel (its waves are independent)
#define TKNN_DB_WAVES 5          // waves per SIMD its register allocation aims at
#endif
constexpr int kDbUnionBlock = TKNN_DB_UNION_BLOCK;
#ifndef TKNN_DB_STACK
#define TKNN_DB_STACK 512  // (a build with 320 exercises the depth-first mode and the overflow fallback: tests)
#endif
constexpr int kDbStack = TKNN_DB_STACK;  // references per wave; popped one at a time (depth first) when fewer than 256 places are left
constexpr int kDbCand = 64;     // the boxes popped in one round, per wave
struct DbCand {
  float lo[3];
  int32_t ref;    // node, or ~slot of a single point
  float hi[3];
  int32_t other;  // the other end of the node's slot range (a leaf: its slot)
};
__device__ __forceinline__ void db_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ void __launch_bounds__(kDbUnionBlock) __attribute__((amdgpu_waves_per_eu(TKNN_DB_WAVES))) db_group_union_kernel(DbArgs a, const int32_t *groups, const unsigned long long *n_groups,
                                                                  unsigned long long *next_packet, unsigned long long *overflow) {
  __shared__ int32_t buf_ref[kDbBuf * kDbUnionBlock];    // [entry][thread]: the group's node (or ~slot of a single point)
  __shared__ int32_t buf_other[kDbBuf * kDbUnionBlock];  // the other end of its slot range | far-corners-within-eps << 31
  __shared__ int32_t stack_all[(kDbUnionBlock / 64) * kDbStack];
  __shared__ __align__(16) DbCand cand_all[(kDbUnionBlock / 64) * kDbCand];
  int32_t *stack = stack_all + (threadIdx.x >> 6) * kDbStack;
  DbCand *cand = cand_all + (threadIdx.x >> 6) * kDbCand;
  const LbvhView &bvh = a.bvh;
  uint32_t node_tests = 0, point_tests = 0;
  if ((a.diag & 16) && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(overflow, 1ull);  // diagnostic library: take the host's fallback
  const long long total = (long long)*n_groups;
  const long long packets = (total + 63) / 64;
  const int lane = threadIdx.x & 63;
  int32_t *my_ref = buf_ref + threadIdx.x, *my_other = buf_other + threadIdx.x;
  const float r = a.reach;
  typedef float t_f4 __attribute__((ext_vector_type(4)));
  const int xcc = (int)__builtin_amdgcn_s_getreg(6164 /* hwreg(HW_REG_XCC_ID, 0, 4) */) & 7;
  const long long chunk = a.chunk;  // packets per chunk dealt to an XCD
  uint32_t seg_empty = 0;           // XCDs whose chunks are used up (wave-uniform)
  float alo[3], ahi[3];
  int32_t a_last = 0, a_core = 0, mine = 0, my_root = 0;
  // B = a tight node or a single leaf with box blo..bhi, first core slot b_core, last slot b_last: an edge?
  // A scan: my core points that are within eps of B's box against B's core points, until the first hit.  With a budget
  // (steps of either loop) it is the first thing tried -- between neighbouring groups of a dense region the first pairs
  // looked at are within eps -- and gives up undecided (-1) when the budget is spent; without one it is exact (1 / 0).
  auto probe = [&](const float *blo, const float *bhi, int32_t b_core, int32_t b_last, int budget) -> int {
    for (int32_t i = a_core; i <= a_last; i = a.next_core[i + 1]) {
      if (budget-- == 0) return -1;
      const LbvhPoint p = bvh.points[i];
      const float bx = fmaxf(fmaxf(blo[0] - p.x, p.x - bhi[0]), 0.f), by = fmaxf(fmaxf(blo[1] - p.y, p.y - bhi[1]), 0.f),
                  bz = fmaxf(fmaxf(blo[2] - p.z, p.z - bhi[2]), 0.f);
      if (bx * bx + by * by + bz * bz > a.eps_out2) continue;  // nothing of B within eps of p
      for (int32_t j = b_core; j <= b_last; j = a.next_core[j + 1]) {
        if (budget-- == 0) return -1;
        const LbvhPoint q = bvh.points[j];
        point_tests++;
        if (knn_sqrt(knn_dist2(q.x, q.y, q.z, p.x, p.y, p.z)) <= a.eps) return 1;
      }
    }
    return 0;
  };
  // The same question answered down the two groups' own subtrees: a pair of sub-nodes is dropped when its nearest
  // faces are beyond eps or one side has no core point, settles the question when its farthest corners are within eps,
  // and is split otherwise (the side with the longer diagonal) -- a few dozen steps where the scan above may look at
  // every point of a group of thousands before it meets one near the other group (50 M heavy-tailed 2-D points with
  // duplicates: 21 ms for the union pass with the scan alone).  1 / 0: edge / none; -1: the stack is full, scan instead.
  int32_t a_ref = 0;
  auto tree_probe = [&](int32_t b_ref) -> int {
    constexpr int kDepth = 24;
    volatile int32_t sx[kDepth], sy[kDepth];  // (volatile: private memory, not 48 registers)
    int top = 0;
    sx[top] = a_ref, sy[top] = b_ref, top++;
    while (top > 0) {
      top--;
      const int32_t x = sx[top], y = sy[top];
      float xlo[3], xhi[3], ylo[3], yhi[3];
      int32_t x_first, x_last, y_first, y_last, x_split = 0, y_split = 0;
      if (x >= 0) {
        const LbvhNode nd = bvh.nodes[x];
        for (int c = 0; c < 3; c++) xlo[c] = nd.lo[c], xhi[c] = nd.hi[c];
        x_first = lbvh_first(x, nd.other), x_last = lbvh_last(x, nd.other), x_split = nd.split;
      } else {
        const LbvhPoint q = bvh.points[~x];
        xlo[0] = xhi[0] = q.x, xlo[1] = xhi[1] = q.y, xlo[2] = xhi[2] = q.z;
        x_first = x_last = ~x;
      }
      if (y >= 0) {
        const LbvhNode nd = bvh.nodes[y];
        for (int c = 0; c < 3; c++) ylo[c] = nd.lo[c], yhi[c] = nd.hi[c];
        y_first = lbvh_first(y, nd.other), y_last = lbvh_last(y, nd.other), y_split = nd.split;
      } else {
        const LbvhPoint q = bvh.points[~y];
        ylo[0] = yhi[0] = q.x, ylo[1] = yhi[1] = q.y, ylo[2] = yhi[2] = q.z;
        y_first = y_last = ~y;
      }
      if (a.next_core[x_first] > x_last || a.next_core[y_first] > y_last) continue;  // a side without a core point
      float far2, near2;
      box_box_dist2(xlo, xhi, ylo, yhi, far2, near2);
      if (near2 > a.eps_out2) continue;
      if (far2 <= a.eps_in2) return 1;
      if (x < 0 && y < 0) {  // two core points within rounding of eps: the spec's arithmetic decides
        point_tests++;
        if (knn_sqrt(knn_dist2(ylo[0], ylo[1], ylo[2], xlo[0], xlo[1], xlo[2])) <= a.eps) return 1;
        continue;
      }
      if (top + 2 > kDepth) return -1;
      const float dx = (xhi[0] - xlo[0]) * (xhi[0] - xlo[0]) + (xhi[1] - xlo[1]) * (xhi[1] - xlo[1]) + (xhi[2] - xlo[2]) * (xhi[2] - xlo[2]);
      const float dy = (yhi[0] - ylo[0]) * (yhi[0] - ylo[0]) + (yhi[1] - ylo[1]) * (yhi[1] - ylo[1]) + (yhi[2] - ylo[2]) * (yhi[2] - ylo[2]);
      if (x >= 0 && (y < 0 || dx >= dy)) {
        sx[top] = x_first == x_split ? ~x_split : x_split, sy[top] = y, top++;
        sx[top] = x_last == x_split + 1 ? ~(x_split + 1) : x_split + 1, sy[top] = y, top++;
      } else {
        sx[top] = x, sy[top] = y_first == y_split ? ~y_split : y_split, top++;
        sx[top] = x, sy[top] = y_last == y_split + 1 ? ~(y_split + 1) : y_split + 1, top++;
      }
    }
    return 0;
  };
  int waiting = 0;
  [[maybe_unused]] uint32_t pk_long = 0;  // (timing) entries of this packet that went the long way, over the lanes: summed at the end
  [[maybe_unused]] unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm_mark = 0;
  const bool timing = TKNN_DIAG_BUILD && (a.diag & 512) && !((a.diag & 2048) && a.near_lo2 < 0.f) && !((a.diag & 4096) && a.near_lo2 >= 0.f);  // 2048 / 4096: the second / first pass only
#define DB_LAP(i)                                                  \
  do {                                                             \
    if (timing) {                                                  \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      tm[i] += now_ - tm_mark;                                     \
      tm_mark = now_;                                              \
    }                                                              \
  } while (0)
  // what the lanes have collected: two rounds of loads for all of it, then one group after the other
  auto settle = [&]() {
    DB_LAP(1);
    if (timing) tm[6]++;
    // (every load of a round is issued whether its entry exists or not, from a place that does: a load under a condition is
    // waited for before the next one is issued -- eight round trips to memory instead of one)
    int32_t b_core[kDbBuf], b_last[kDbBuf], par[kDbBuf], at[kDbBuf];
#pragma unroll
    for (int w = 0; w < kDbBuf; w++) {
      const int32_t x = my_ref[w * kDbUnionBlock], o = my_other[w * kDbUnionBlock] & 0x7fffffff;
      const int32_t end = x >= 0 ? x : ~x;
      const bool is = w < waiting;
      b_last[w] = is ? max(end, o) : -1;
      at[w] = is ? min(end, o) : 0;
    }
#pragma unroll
    for (int w = 0; w < kDbBuf; w++) b_core[w] = a.next_core[at[w]];
#pragma unroll
    for (int w = 0; w < kDbBuf; w++) {
      if (!(w < waiting)) b_core[w] = 0x7fffffff;
      at[w] = b_core[w] <= b_last[w] ? b_core[w] : 0;
    }
#pragma unroll
    for (int w = 0; w < kDbBuf; w++) par[w] = uf_load(a.parent + at[w]);
    const int32_t my_up = uf_load(a.parent + my_root);  // (my root may have been hooked under another since I last looked)
#pragma unroll
    for (int w = 0; w < kDbBuf; w++) at[w] = par[w];  // >= 0: parents of places that exist
    // a third round: the parents' parents -- with path halving a slot is rarely more than two steps below its root, and a
    // group whose ROOT is mine needs no look at all (most of what a lane collects in the second pass, where the sets have
    // grown together but the first core slots still point at roots of the first pass)
#pragma unroll
    for (int w = 0; w < kDbBuf; w++) par[w] = uf_load(a.parent + at[w]);
    if (my_up != my_root) my_root = uf_find(a.parent, my_up);
    uint32_t todo = 0;  // collected groups that have a core point and are not in my set as far as these loads can tell
#pragma unroll
    for (int w = 0; w < kDbBuf; w++) {
      if (!(b_core[w] <= b_last[w])) continue;  // no core point in it (or no entry)
      if (par[w] != at[w]) par[w] = uf_find(a.parent, par[w]);  // (more than two steps: the chain, rarely)
      if (par[w] != my_root) todo |= 1u << w;  // par[w]: the group's root
    }
    if (a.diag & 2) todo = 0;
    // The short way first: if the farthest corners are within eps the edge is known, and one compare-and-swap hooks the
    // larger root under the smaller.  (A root read here may be hooked under another by a different wave by now: the swap then
    // fails, or hooks under a slot that is no root any more but IS in my set; the first is left to the long way below, the
    // second is fine.)  Tried and dropped: deciding the other pairs here too by ONE distance, first core point against first
    // core point -- in the first pass that is an edge for half of them, and the pass got 0.3 ms SLOWER (2.13 against 1.82 ms
    // for both).
#pragma unroll
    for (int w = 0; w < kDbBuf; w++) {
      if (!(todo & (1u << w)) || !a.short_way || my_other[w * kDbUnionBlock] >= 0) continue;
      const int32_t ro = par[w];
      if (ro == my_root) {  // (joined by an earlier entry of this very loop)
        todo &= ~(1u << w);
        continue;
      }
      const int32_t lo = min(ro, my_root), hi = max(ro, my_root);
      if (atomicCAS(a.parent + hi, hi, lo) == hi) {
        todo &= ~(1u << w);
        my_root = lo;
      }
    }
    // the rest one after the other (not unrolled: the probe's private stack would be kept once per copy)
    while (todo) {
      const int w = __ffs((int)todo) - 1;
      todo &= todo - 1u;
      if (timing) pk_long++;
      const int32_t B = my_ref[w * kDbUnionBlock], packed = my_other[w * kDbUnionBlock];
      const int32_t end = B >= 0 ? B : ~B, o = packed & 0x7fffffff;
      const int32_t other_last = max(end, o), other = a.next_core[min(end, o)];  // its first core slot stands for the group
      if (uf_find(a.parent, other) == my_root) continue;
      my_root = uf_find(a.parent, my_root);  // my root may have been hooked under another meanwhile
      if (uf_find(a.parent, other) == my_root) continue;
      bool edge = packed < 0;
      if (!edge && !(a.diag & 1)) {
        // a few pairs by scanning, then down the two subtrees, and the whole scan if that runs out of stack
        float blo[3], bhi[3];
        if (B >= 0) {
          const LbvhNode nd = bvh.nodes[B];
          for (int c = 0; c < 3; c++) blo[c] = nd.lo[c], bhi[c] = nd.hi[c];
        } else {
          const LbvhPoint q = bvh.points[~B];
          blo[0] = bhi[0] = q.x, blo[1] = bhi[1] = q.y, blo[2] = bhi[2] = q.z;
        }
        int found = probe(blo, bhi, other, other_last, a.scan_budget);
        if (found < 0 && !(a.diag & 32)) found = tree_probe(B);
        if (found < 0) found = probe(blo, bhi, other, other_last, -1);
        edge = found > 0;
      }
      if (edge) {
        uf_unite(a.parent, mine, other);
        my_root = uf_find(a.parent, mine);
      }
    }
    waiting = 0;
    DB_LAP(2);
  };
  if (timing) tm_mark = __builtin_amdgcn_s_memtime();
  [[maybe_unused]] const unsigned long long wave_t0 = tm_mark;
  for (;;) {
    // ---- take a packet.  Each XCD has its own L2: the list is dealt to the XCDs in chunks of packets, a wave takes from
    // the chunks of the XCD it runs on and from the others' when those are used up.
    long long packet = -1;
    for (int t = 0; t < 8 && packet < 0; t++) {
      const int from = (xcc + t) & 7;
      if (seg_empty & (1u << from)) continue;
      unsigned long long v = 0;
      if (t > 0) {  // another XCD's packets (mine are used up): a look before the turn -- cursors only grow, a load does not queue
        if (lane == 0) v = __hip_atomic_load(next_packet + from * kDbCursorStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = __shfl(v, 0);
        if ((((long long)v / chunk) * 8 + from) * chunk + (long long)v % chunk >= packets) {
          seg_empty |= 1u << from;
          continue;
        }
      }
      if (lane == 0) v = atomicAdd(next_packet + from * kDbCursorStride, 1ull);
      v = __shfl(v, 0);
      const long long p = (((long long)v / chunk) * 8 + from) * chunk + (long long)v % chunk;
      if (p < packets)
        packet = p;
      else
        seg_empty |= 1u << from;  // places only grow: this XCD's chunks are used up
    }
    if (packet < 0) break;
    [[maybe_unused]] const unsigned long long packet_t0 = timing ? __builtin_amdgcn_s_memtime() : 0ull;
    [[maybe_unused]] const unsigned long long pk_rounds0 = tm[5], pk_settles0 = tm[6], pk_settle_t0 = tm[2];
    [[maybe_unused]] const uint32_t pk_points0 = point_tests;
    pk_long = 0;
    // ---- lanes = the packet's groups
    const long long g = packet * 64 + lane;
    const bool have = g < total;
    int32_t a_first = 0x7fffffff;
    a_last = 0x7fffffff;  // a lane without a group: nothing lies after it
    alo[0] = alo[1] = alo[2] = ahi[0] = ahi[1] = ahi[2] = 0.f;
    if (have) {
      const int32_t G = groups[g];
      a_ref = G;
      if (G >= 0) {
        const LbvhNode nd = bvh.nodes[G];
        for (int c = 0; c < 3; c++) alo[c] = nd.lo[c], ahi[c] = nd.hi[c];
        a_first = lbvh_first(G, nd.other);
        a_last = lbvh_last(G, nd.other);
      } else {
        const LbvhPoint p = bvh.points[~G];
        alo[0] = ahi[0] = p.x, alo[1] = ahi[1] = p.y, alo[2] = ahi[2] = p.z;
        a_first = a_last = ~G;
      }
      a_core = a.next_core[a_first];  // <= a_last: listed groups have a core point
      mine = a_core;
      my_root = uf_find(a.parent, mine);
    }
    // second pass: the one set all the packet's groups were in when the first pass had ended, if they were (the inside of
    // a cluster): a node whose core points were all in that set (a.uni) has nothing to offer to any of them.  Both sides
    // of the comparison are names of that moment (db_uniform_kernel) -- the roots move while this pass unites sets.
    int32_t pk_root = -1;
    if (a.uni) {
      const int32_t set0 = !have ? -1 : a_ref >= 0 ? a.uni[a_ref] : a.uni_leaf[~a_ref];
      const int32_t r0 = __builtin_amdgcn_readfirstlane(set0);
      if (__ballot(have && set0 != r0) == 0ull) pk_root = r0;
      if (TKNN_DIAG_BUILD && (a.diag & 1024) && lane == 0) atomicAdd(&a.diag_out[pk_root >= 0 ? 8 : 9], 1ull);
    }
    // nothing before the packet's earliest group end can lie after any of its groups
    int32_t low = a_last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) low = min(low, __shfl_xor(low, off));
    // Bounding boxes of the packet's groups, one per 32 lanes (Morton neighbours; two boxes hug a packet that lies across
    // a jump of the Z curve better than one), widened by the reach: a popped node that meets neither can be reached by no
    // group and is dropped where it is popped -- 64 nodes per instruction -- instead of costing a turn of the test loop
    // below, which deals with one node at a time (47 % of the kernel's wave time, half of it on nodes no group reaches).
    float u_lo[kDbBoxes][3], u_hi[kDbBoxes][3];
    {
      float l[3], h[3];
      for (int c = 0; c < 3; c++) {
        // (a margin of a millionth of the magnitudes involved: fl(e.lo - r) <= a.hi, the groups' own test, and
        // e.lo <= fl(a.hi + r) can differ in the last place; the prefilter must never drop what a group would take)
        const float slack = 1e-6f * (fabsf(alo[c]) + fabsf(ahi[c]) + r);
        l[c] = have ? (alo[c] - r) - slack : INFINITY;
        h[c] = have ? (ahi[c] + r) + slack : -INFINITY;
#pragma unroll
        for (int off = 32 / kDbBoxes; off > 0; off >>= 1) {
          l[c] = fminf(l[c], __shfl_xor(l[c], off));
          h[c] = fmaxf(h[c], __shfl_xor(h[c], off));
        }
#pragma unroll
        for (int half = 0; half < kDbBoxes; half++) {
          u_lo[half][c] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(l[c]), (64 / kDbBoxes) * half));
          u_hi[half][c] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h[c]), (64 / kDbBoxes) * half));
        }
      }
    }
    // ---- the packet's walk
    const unsigned long long below = (1ull << lane) - 1ull;
    int top = 1;
    if (lane == 0) stack[0] = bvh.root;
    db_wave_sync();
    DB_LAP(4);
    if (timing) tm[7]++;
    while (top > 0) {
      if (timing) tm[5]++;
      // lanes = nodes: pop, load, stage
      const int width = top > kDbStack - 256 ? 1 : 64;
      const int n_pop = min(top, width);
      const bool valid = lane < n_pop;
      const int32_t ref = valid ? stack[top - 1 - lane] : bvh.root;
      top -= n_pop;
      const bool is_node = ref >= 0;
      const int32_t slot = is_node ? 0 : ~ref;
      const t_f4 *src = is_node ? (const t_f4 *)(bvh.nodes + ref) : (const t_f4 *)(bvh.points + slot);
      const t_f4 v0 = src[0], v1 = src[is_node ? 1 : 0];
      // (loads that only some lanes or some packets need are issued all the same, beside the boxes: under a condition they
      // would be a second and a third round trip to memory in every round)
      const uint8_t leaf_core = a.core_sorted[slot];
      int32_t node_set = -1;
      if (a.uni) node_set = (is_node ? a.uni : a.uni_leaf)[is_node ? ref : slot];  // (a leaf that is not core: whatever is there)
      const bool any_core = is_node || leaf_core != 0;
      if (valid) node_tests++;
      const int32_t split = __float_as_int(v0.w), b_other = is_node ? __float_as_int(v1.w) : slot;
      const int32_t my_end = is_node ? ref : slot;
      const int32_t b_first = min(my_end, b_other), b_last = max(my_end, b_other);
      const float ex = v1.x - v0.x, ey = v1.y - v0.y, ez = v1.z - v0.z;
      const bool tight = !is_node || ex * ex + ey * ey + ez * ez <= a.eps_in2;  // node_is_tight
      {
        DbCand &e = cand[lane];
        e.lo[0] = v0.x, e.lo[1] = v0.y, e.lo[2] = v0.z;
        e.ref = ref;
        e.hi[0] = v1.x, e.hi[1] = v1.y, e.hi[2] = v1.z;
        e.other = b_other;
      }
      // nodes nothing can come of: past the packet's last use, a leaf that is not core, out of every group's reach
      bool near_packet = false;
#pragma unroll
      for (int half = 0; half < kDbBoxes; half++)
        near_packet |= (v0.x <= u_hi[half][0]) & (u_lo[half][0] <= v1.x) & (v0.y <= u_hi[half][1]) & (u_lo[half][1] <= v1.y) &
                       (v0.z <= u_hi[half][2]) & (u_lo[half][2] <= v1.z);
      if (TKNN_DIAG_BUILD && (a.diag & 1024) && a.uni) {
        const unsigned long long sk = __ballot(valid && pk_root >= 0 && node_set == pk_root), al = __ballot(valid), un = __ballot(valid && is_node && a.uni[ref] >= 0);
        if (lane == 0) atomicAdd(&a.diag_out[10], (unsigned long long)__popcll(sk)), atomicAdd(&a.diag_out[11], (unsigned long long)__popcll(al)), atomicAdd(&a.diag_out[12], (unsigned long long)__popcll(un));
      }
      const unsigned long long m_live = __ballot(valid && any_core && b_last > low && near_packet && !(pk_root >= 0 && node_set == pk_root)), m_tight = __ballot(tight);
      db_wave_sync();
      DB_LAP(0);
      // lanes = groups
      unsigned long long m_open = 0ull;  // nodes some group reaches and that are not tight
      for (unsigned long long todo = m_live; todo;) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const DbCand e = cand[j];
        // (the staged node is the same in every lane: its slot range as scalars)
        const int32_t e_ref = __builtin_amdgcn_readfirstlane(e.ref), e_other = __builtin_amdgcn_readfirstlane(e.other);
        const int32_t e_end = e_ref >= 0 ? e_ref : ~e_ref;
        const int32_t e_first = min(e_end, e_other), e_last = max(e_end, e_other);
        // within reach along every axis: the largest separation of the two boxes against r -- subtractions and maxima instead
        // of six compares and their mask operations (a third of this loop's slow instructions; as conservative as
        // fl(e.lo - r) <= a.hi: a difference that is <= r before rounding is <= r after it)
        const float sx = fmaxf(e.lo[0] - ahi[0], alo[0] - e.hi[0]), sy = fmaxf(e.lo[1] - ahi[1], alo[1] - e.hi[1]),
                    sz = fmaxf(e.lo[2] - ahi[2], alo[2] - e.hi[2]);
        const bool hit = (fmaxf(fmaxf(sx, sy), sz) <= r) & (e_last > a_last);
        if (__ballot(hit) == 0ull) continue;
        if (!((m_tight >> j) & 1ull)) {
          m_open |= 1ull << j;
          continue;
        }
        float far2, near2;
        box_box_dist2(alo, ahi, e.lo, e.hi, far2, near2);
        if (hit && e_first > a_last && near2 <= a.near_hi2 && near2 > a.near_lo2) {
          my_ref[waiting * kDbUnionBlock] = e.ref;
          my_other[waiting * kDbUnionBlock] = e.other | (far2 <= a.eps_in2 ? (int32_t)0x80000000 : 0);
          waiting++;
        }
        if (__ballot(waiting == kDbBuf) != 0ull) settle();
      }
      DB_LAP(1);
      // lanes = nodes: children (the left one only if it reaches past the packet's earliest group)
      const bool kids = (m_open >> lane) & 1ull;
      const bool left_too = kids && split > low;
      const unsigned long long m_left = __ballot(left_too);
      const int n_left = __popcll(m_left), n_right = __popcll(m_open);
      if (top + n_left + n_right > kDbStack) {  // (cannot happen below a tree depth of some 250 levels; the host then falls back)
        if (lane == 0) atomicAdd(overflow, 1ull);
        top = 0;
        break;
      }
      if (left_too) stack[top + __popcll(m_left & below)] = b_first == split ? ~split : split;                              // lbvh_left_ref
      if (kids) stack[top + n_left + __popcll(m_open & below)] = b_last == split + 1 ? ~(split + 1) : split + 1;  // lbvh_right_ref
      top += n_left + n_right;
      db_wave_sync();
      DB_LAP(3);
    }
    if (__ballot(waiting > 0) != 0ull) settle();
    if (timing && lane == 0 && a.pk_diag) {
      float ext = 0.f;
      for (int c = 0; c < 3; c++) ext = fmaxf(ext, fmaxf(u_hi[0][c], u_hi[kDbBoxes - 1][c]) - fminf(u_lo[0][c], u_lo[kDbBoxes - 1][c]));
      int32_t *d = a.pk_diag + packet * 4;
      d[0] = (int32_t)(__builtin_amdgcn_s_memtime() - packet_t0), d[1] = (int32_t)(ext * 1e6f), d[2] = (int32_t)(tm[5] - pk_rounds0), d[3] = (int32_t)(tm[6] - pk_settles0);
    }
    if (timing && lane == 0) {  // the longest packet, and how many take more than twice / four times 2^20 ticks
      const unsigned long long el = __builtin_amdgcn_s_memtime() - packet_t0;
      atomicMax(&a.diag_out[12], el);
      if (el > (1ull << 20)) atomicAdd(&a.diag_out[10], 1ull);
      if (el > (1ull << 21)) atomicAdd(&a.diag_out[11], 1ull);
    }
    if (timing) {
      uint32_t lw = pk_long, pt = point_tests - pk_points0;
      for (int off = 32; off > 0; off >>= 1) lw += __shfl_xor(lw, off), pt += __shfl_xor(pt, off);
      const unsigned long long el = __builtin_amdgcn_s_memtime() - packet_t0;
      if (lane == 0 && el > (1ull << 21)) {
        atomicAdd(&a.diag_out[16], tm[5] - pk_rounds0), atomicAdd(&a.diag_out[17], tm[6] - pk_settles0);
        atomicAdd(&a.diag_out[18], (unsigned long long)lw), atomicAdd(&a.diag_out[19], (unsigned long long)pt);
        atomicAdd(&a.diag_out[9], tm[2] - pk_settle_t0);
        float ext = 0.f;
        for (int c = 0; c < 3; c++) ext = fmaxf(ext, fmaxf(u_hi[0][c], u_hi[kDbBoxes - 1][c]) - fminf(u_lo[0][c], u_lo[kDbBoxes - 1][c]));
(void)ext;
      }
    }
  }
  if (timing && lane == 0) {
    for (int i = 0; i < 8; i++) atomicAdd(&a.diag_out[i], tm[i]);
    const unsigned long long el = __builtin_amdgcn_s_memtime() - wave_t0;  // how evenly the packets fill the waves: mean / longest wave
    atomicAdd(&a.diag_out[13], el), atomicMax(&a.diag_out[14], el), atomicAdd(&a.diag_out[15], 1ull);
  }
#undef DB_LAP
  if (a.diag & 8) {  // [6] most walk steps of a wave, [7] their sum
    if (lane == 0) {
      atomicMax(&a.stats[6], (unsigned long long)node_tests);
      atomicAdd(&a.stats[7], (unsigned long long)node_tests);
    }
  }
  db_add_stats_wave(a.stats + 2, node_tests, point_tests);  // (no LDS words: 16 384 bytes per workgroup are ten workgroups per CU, 16 400 are nine)
}

// min_row[root] = min(min_row[root], row).  The value only falls, so a read that finds it at or below `row` settles the matter
// without an atomic: a cluster of millions of points would otherwise queue an atomic per wave on ONE address (50 M 2-D
// points, one giant cluster: 6 ms for the pass; with the read first, a handful of atomics get through).
__device__ __forceinline__ void db_min_row(int32_t *cell, int32_t row) {
  if (__hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > row) atomicMin(cell, row);
}

// after the unions: every core slot points at its root, and the root learns the smallest ROW of its cluster -- clusters are
// numbered by that (the spec: ascending smallest core index).  A wave's slots mostly share one root: one atomic per wave
// and root, not per point.
// Four consecutive slots per thread, their words fetched as one 16-byte load per array: a thread with one slot has one
// 4-byte load in flight per step of its chain (flag -> parent -> parent's parent -> row), and half a million such threads on
// the device do not fill the memory system (0.15 ms for 0.2 GB).
constexpr int kDbPer = 4;
__global__ void __launch_bounds__(kDbBlock) db_flatten_kernel(DbArgs a, int32_t *zero) {
  const int32_t n = a.bvh.n;
  const long long t0 = ((long long)blockIdx.x * kDbBlock + threadIdx.x) * kDbPer;
  int32_t par[kDbPer], prim[kDbPer];
  bool core[kDbPer];
  if (t0 + kDbPer <= n) {
    const int4 p4 = *reinterpret_cast<const int4 *>(a.parent + t0), r4 = *reinterpret_cast<const int4 *>(a.bvh.prim_id + t0);
    const uchar4 c4 = *reinterpret_cast<const uchar4 *>(a.core_sorted + t0);
    par[0] = p4.x, par[1] = p4.y, par[2] = p4.z, par[3] = p4.w;
    prim[0] = r4.x, prim[1] = r4.y, prim[2] = r4.z, prim[3] = r4.w;
    core[0] = c4.x != 0, core[1] = c4.y != 0, core[2] = c4.z != 0, core[3] = c4.w != 0;
#pragma unroll
    for (int k = 0; k < kDbPer; k++) zero[t0 + k] = 0;  // the root flags of the next launch (the group list lived here; its place is 16-byte aligned only if n is a multiple of four)
  } else {
#pragma unroll
    for (int k = 0; k < kDbPer; k++) {
      const bool in = t0 + k < n;
      const long long t = in ? t0 + k : 0;
      par[k] = a.parent[t], prim[k] = a.bvh.prim_id[t], core[k] = in && a.core_sorted[t] != 0;
      if (in) zero[t] = 0;
    }
  }
  // (plain loads and stores: the unions are over -- what this launch reads is final or, where another lane's store below has
  // or has not arrived, a root or an older ancestor in the same tree: the walk ends at the root either way)
  int32_t up[kDbPer], root[kDbPer];
#pragma unroll
  for (int k = 0; k < kDbPer; k++) up[k] = a.parent[core[k] ? par[k] : 0];
#pragma unroll
  for (int k = 0; k < kDbPer; k++) {
    root[k] = par[k];
    if (core[k] && up[k] != par[k]) {
      int32_t x = up[k], p = a.parent[x];
      while (p != x) x = p, p = a.parent[x];
      root[k] = x;
    }
    if (core[k] && root[k] != par[k]) a.parent[t0 + k] = root[k];
  }
  // the thread's slots of its first core slot's root as one (root, smallest row); the others (a boundary between clusters
  // inside four slots) send their own
  int32_t my_root = -1, row = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < kDbPer; k++) {
    if (!core[k]) continue;
    if (my_root < 0) my_root = root[k];
    if (root[k] == my_root)
      row = min(row, prim[k]);
    else
      db_min_row(a.min_row + root[k], prim[k]);
  }
  const int lane = threadIdx.x & 63;
  // the wave's two most common cases first -- one root for all its slots, or two -- as one atomic each; slots of further
  // roots (sparse regions: many small clusters per wave, hardly two slots on one address) send their own
  bool pending = my_root >= 0;
  for (int round = 0; round < 2; round++) {
    const unsigned long long todo = __ballot(pending);
    if (!todo) break;
    const int j = __ffsll((long long)todo) - 1;
    const int32_t r_j = __shfl(my_root, j);
    const bool same = pending && my_root == r_j;
    int32_t m = same ? row : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off));
    if (lane == j) db_min_row(a.min_row + r_j, m);
    pending = pending && !same;
  }
  if (pending) db_min_row(a.min_row + my_root, row);
}
// roots flag their cluster's smallest row; an exclusive scan over the rows then numbers the clusters
__global__ void __launch_bounds__(kDbBlock) db_root_kernel(DbArgs a, int32_t *is_first_row) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n && a.core_sorted[t] && a.parent[t] == t) is_first_row[a.min_row[t]] = 1;
}

// The points that are not core (few -- BASELINE config 3: 0.13 % --, scattered over the waves, listed in slot order since the
// core flags were known: db_core_pos_kernel) take the lowest-numbered cluster among their core neighbours.  Finding those
// neighbours needs the core flags only, not the clusters: the walks -- chains of dependent loads by a few hundred waves, 0.25 ms
// on config 3 -- run BESIDE the group unions on a stream of their own and leave, per listed point, the core neighbours they met
// (one per tight node, db for_each_core_group; a point that is not core has fewer than minPts - 1 of them: `per` - 1 places, and
// a count of -1 if that should ever not be enough).  Lists that would not fit the room (`capacity` words; a set that is mostly
// noise) are not written: the label kernel then walks itself, as it does for a count of -1.
__global__ void __launch_bounds__(kDbBlock) db_border_walk_kernel(DbArgs a, const int32_t *pending, const unsigned long long *n_pending, int32_t *lists,
                                                                  int per, long long capacity) {
  __shared__ unsigned long long blk_stats[2];
  if (threadIdx.x < 2) blk_stats[threadIdx.x] = 0ull;
  __syncthreads();
  uint32_t node_tests = 0, point_tests = 0;
  const long long total = (long long)*n_pending;
  if (total * per <= capacity) {
    for (long long i = (long long)blockIdx.x * kDbBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kDbBlock) {
      const LbvhPoint q = a.bvh.points[pending[i]];
      int32_t *mine = lists + i * per;
      int32_t cnt = 0;
      for_each_core_group(
          a, q, -1, [](int32_t) { return false; },
          [&](int32_t other) {
            if (cnt < per - 1) mine[1 + cnt] = other;
            cnt++;
          },
          node_tests, point_tests);
      mine[0] = cnt <= per - 1 ? cnt : -1;
    }
  }
  db_add_stats(a.stats + 4, blk_stats, node_tests, point_tests);
}

// Labels: core points take their cluster's number (and every point's core flag goes to its row); the listed points take the
// smallest number among the core neighbours db_border_walk_kernel has left them (or walk now).  ONE launch: the first
// `walk_blocks` workgroups serve the list (grid-stride), the rest stream -- two scattered stores per point.
// `by_slot` (or null): the labels stay BY SLOT -- a core slot's number (>= 0; four consecutive slots one 16-byte store), a
// listed slot's -1 (noise) or -3 - label (border) -- and db_rows_from_slots_kernel carries them to the caller's rows through the
// tree's inverse permutation with coalesced stores.  The scatter wrote a partial sector per point and array: rocprofv3 counted
// 648 MB written for 50 MB of output at BASELINE config 3.
__global__ void __launch_bounds__(kDbBlock) db_label_kernel(DbArgs a, const int32_t *pending, const unsigned long long *n_pending, int walk_blocks,
                                                            const int32_t *lists, int per, long long capacity, int32_t *by_slot) {
  __shared__ unsigned long long blk_stats[2];
  if (threadIdx.x < 2) blk_stats[threadIdx.x] = 0ull;
  __syncthreads();
  uint32_t node_tests = 0, point_tests = 0;
  if ((int)blockIdx.x >= walk_blocks) {
    // four consecutive slots per thread (db_flatten_kernel: 16-byte loads, four chains in flight); a core slot's parent IS its
    // root since that kernel
    const int32_t n = a.bvh.n;
    const long long t0 = ((long long)(blockIdx.x - walk_blocks) * kDbBlock + threadIdx.x) * kDbPer;
    int32_t row[kDbPer], root[kDbPer];
    bool in[kDbPer], core[kDbPer];
    if (t0 + kDbPer <= n) {
      const int4 p4 = *reinterpret_cast<const int4 *>(a.parent + t0), r4 = *reinterpret_cast<const int4 *>(a.bvh.prim_id + t0);
      const uchar4 c4 = *reinterpret_cast<const uchar4 *>(a.core_sorted + t0);
      root[0] = p4.x, root[1] = p4.y, root[2] = p4.z, root[3] = p4.w;
      row[0] = r4.x, row[1] = r4.y, row[2] = r4.z, row[3] = r4.w;
      core[0] = c4.x != 0, core[1] = c4.y != 0, core[2] = c4.z != 0, core[3] = c4.w != 0;
      in[0] = in[1] = in[2] = in[3] = true;
    } else {
#pragma unroll
      for (int k = 0; k < kDbPer; k++) {
        in[k] = t0 + k < n;
        const long long t = in[k] ? t0 + k : 0;
        root[k] = a.parent[t], row[k] = a.bvh.prim_id[t], core[k] = in[k] && a.core_sorted[t] != 0;
      }
    }
    int32_t first_row[kDbPer], number[kDbPer];
#pragma unroll
    for (int k = 0; k < kDbPer; k++) first_row[k] = a.min_row[core[k] ? root[k] : 0];
#pragma unroll
    for (int k = 0; k < kDbPer; k++) number[k] = a.rank[core[k] ? first_row[k] : 0];
    if (by_slot) {
      // (a slot that is not core belongs to the list part of this launch: not touched here)
      if (t0 + kDbPer <= n && core[0] && core[1] && core[2] && core[3]) {
        *reinterpret_cast<int4 *>(by_slot + t0) = make_int4(number[0], number[1], number[2], number[3]);
      } else {
#pragma unroll
        for (int k = 0; k < kDbPer; k++)
          if (in[k] && core[k]) by_slot[t0 + k] = number[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < kDbPer; k++) {
        if (!in[k]) continue;
        if (a.core) a.core[row[k]] = core[k];
        if (core[k]) a.labels[row[k]] = number[k];
      }
    }
  } else {
    const long long total = (long long)*n_pending;
    const bool listed = lists != nullptr && total * per <= capacity;
    for (long long i = (long long)blockIdx.x * kDbBlock + threadIdx.x; i < total; i += (long long)walk_blocks * kDbBlock) {
      const int32_t t = pending[i];
      int32_t first_row = -1;  // of my cluster: the smallest over the clusters of my core neighbours (the lowest label)
      auto take = [&](int32_t other) {
        const int32_t m = a.min_row[uf_find(a.parent, other)];
        if (first_row < 0 || m < first_row) first_row = m;
      };
      const int32_t cnt = listed ? lists[i * per] : -1;
      if (cnt >= 0) {
        for (int32_t j = 0; j < cnt; j++) take(lists[i * per + 1 + j]);
      } else {
        const LbvhPoint q = a.bvh.points[t];
        for_each_core_group(a, q, -1, [](int32_t) { return false; }, take, node_tests, point_tests);
      }
      const int32_t label = first_row < 0 ? -1 : a.rank[first_row];
      if (by_slot)
        by_slot[t] = label < 0 ? -1 : -3 - label;
      else
        a.labels[a.bvh.prim_id[t]] = label;
    }
  }
  db_add_stats(a.stats + 4, blk_stats, node_tests, point_tests);
}
// rows <- slots: four consecutive ROWS per thread; their slots come from the tree's inverse permutation (one 16-byte load), the
// slots' words are four scattered 4-byte reads of an array the launch before has just written (40 MB at 10 M points: it sits
// in the 256 MB of cache in front of the memory), the rows' labels and core flags leave as one 16-byte and one 4-byte store
__global__ void __launch_bounds__(kDbBlock) db_rows_from_slots_kernel(DbArgs a, const int32_t *row_slot, const int32_t *by_slot) {
  const int32_t n = a.bvh.n;
  const long long r0 = ((long long)blockIdx.x * kDbBlock + threadIdx.x) * kDbPer;
  if (r0 >= n) return;
  int32_t v[kDbPer];
  if (r0 + kDbPer <= n) {
    const int4 s4 = *reinterpret_cast<const int4 *>(row_slot + r0);
    v[0] = by_slot[s4.x], v[1] = by_slot[s4.y], v[2] = by_slot[s4.z], v[3] = by_slot[s4.w];
    int32_t lab[kDbPer];
#pragma unroll
    for (int k = 0; k < kDbPer; k++) lab[k] = v[k] >= -1 ? v[k] : -3 - v[k];
    *reinterpret_cast<int4 *>(a.labels + r0) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    if (a.core) *reinterpret_cast<uchar4 *>(a.core + r0) = make_uchar4(v[0] >= 0, v[1] >= 0, v[2] >= 0, v[3] >= 0);
  } else {
    for (long long r = r0; r < n; r++) {
      const int32_t w = by_slot[row_slot[r]];
      a.labels[r] = w >= -1 ? w : -3 - w;
      if (a.core) a.core[r] = w >= 0;
    }
  }
}

// tknnDbscanAssign: the caller has decided the label of every core point (>= 0; < 0: not core); a
// core point keeps it, any other point takes the smallest label among its core neighbours, or -1
__global__ void __launch_bounds__(kDbBlock) db_core_from_labels_kernel(DbArgs a, const int32_t *core_label) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n) a.core_sorted[t] = core_label[a.bvh.prim_id[t]] >= 0;
}
__global__ void __launch_bounds__(kDbBlock) db_assign_kernel(DbArgs a, const int32_t *core_label) {
  __shared__ unsigned long long blk_stats[2];
  if (threadIdx.x < 2) blk_stats[threadIdx.x] = 0ull;
  __syncthreads();
  uint32_t node_tests = 0, point_tests = 0;
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n) {
    const LbvhPoint q = a.bvh.points[t];
    const int32_t row = a.bvh.prim_id[t];
    int32_t best = core_label[row];
    if (best < 0) {
      best = -1;
      for_each_core_group(
          a, q, -1, [](int32_t) { return false; },
          [&](int32_t other) {
            const int32_t l = core_label[a.bvh.prim_id[other]];  // a tight node's core points share one label: its first stands for all
            if (best < 0 || l < best) best = l;
          },
          node_tests, point_tests);
    }
    a.labels[row] = best;
  }
  db_add_stats(a.stats + 4, blk_stats, node_tests, point_tests);
}

// ---- "eps auto-grown" (tknnDbscanAuto; spec: oracle/dbscan_oracle.c, dbref_dbscan_auto) ----------------------------
// A round of the growth loop only has to COUNT the noise points: is there a core point within eps of a point that is
// not core itself?  One traversal with an early exit; a subtree without a core point (next_core) is skipped, a node
// wholly inside the sphere that holds one settles the question.  Points found not to be noise never are again (a core
// point stays core as eps grows), so later rounds probe the remaining noise only.
__device__ __forceinline__ bool db_has_core_neighbour(const DbArgs &a, const LbvhPoint &q, int32_t from, int32_t until, int32_t skip,
                                                      int32_t skip_rope, uint32_t &node_tests, uint32_t &point_tests) {
  const LbvhView &bvh = a.bvh;
  const float r = a.eps_wide;
  int32_t ref = from;
  while (ref != until) {
    if (ref == skip) {  // the subtree searched already
      ref = skip_rope;
      continue;
    }
    if (ref >= 0) {
      const LbvhNode nd = bvh.nodes[ref];
      const int32_t rope = bvh.rope_node[ref];  // with the node: half the chain of dependent loads
      node_tests++;
      const bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) & (q.y <= nd.hi[1] + r) &
                       (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
      if (!hit || a.next_core[lbvh_first(ref, nd.other)] > lbvh_last(ref, nd.other)) {  // out of reach, or no core point below
        ref = rope;
        continue;
      }
      float far2, near2;
      box_dist2(nd, q, far2, near2);
      if (far2 <= a.eps_in2) return true;  // all of it within eps, and a core point among it
      if (near2 > a.eps_out2) {
        ref = rope;
        continue;
      }
      ref = lbvh_left_ref(ref, nd);
    } else {
      const int32_t slot = ~ref;
      const uint8_t is_core = a.core_sorted[slot];
      const LbvhPoint p = bvh.points[slot];
      const int32_t rope = bvh.rope_leaf[slot];
      if (is_core) {
        point_tests++;
        if (knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z)) <= a.eps) return true;
      }
      ref = rope;
    }
  }
  return false;
}
// ... first in the subtree a few levels above the point's own group (the core-flag kernel has left its root in near_node): a
// point that is not noise has its core neighbour next to it, and a walk from the root spends two dozen steps getting there
__device__ __forceinline__ bool db_has_core_neighbour(const DbArgs &a, const LbvhPoint &q, int32_t t, uint32_t &node_tests, uint32_t &point_tests) {
  const LbvhView &bvh = a.bvh;
  const int32_t near = a.near_node ? a.near_node[t] : bvh.root;
  if (near < 0 || near == bvh.root) return db_has_core_neighbour(a, q, bvh.root, LBVH_END, LBVH_END, LBVH_END, node_tests, point_tests);
  const int32_t near_rope = bvh.rope_node[near];
  if (db_has_core_neighbour(a, q, near, near_rope, LBVH_END, LBVH_END, node_tests, point_tests)) return true;
  return db_has_core_neighbour(a, q, bvh.root, LBVH_END, near, near_rope, node_tests, point_tests);  // the rest of the tree
}

// noise[slot] (per sorted slot): in, unless first_round: 1 = was noise in the round before; out: 1 = is noise now.
// stats[0..1] += node / point tests, stats[6] += points still noise.
__global__ void __launch_bounds__(kDbBlock) db_noise_probe_kernel(DbArgs a, uint8_t *noise, int first_round) {
  __shared__ unsigned long long blk_stats[2], blk_noise[2];
  if (threadIdx.x < 2) blk_stats[threadIdx.x] = blk_noise[threadIdx.x] = 0ull;
  __syncthreads();
  uint32_t node_tests = 0, point_tests = 0, still = 0;
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n) {
    if (a.core_sorted[t]) {
      noise[t] = 0;
    } else if (first_round || noise[t]) {
      still = db_has_core_neighbour(a, a.bvh.points[t], t, node_tests, point_tests) ? 0u : 1u;
      noise[t] = (uint8_t)still;
    }
  }
  db_add_stats(a.stats + 0, blk_stats, node_tests, point_tests);
  db_add_stats(a.stats + 6, blk_noise, still, 0u);
}

// per-slot flags to the caller's rows
__global__ void __launch_bounds__(kDbBlock) db_rows_kernel(DbArgs a, const uint8_t *by_slot, uint8_t *by_row) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n) by_row[a.bvh.prim_id[t]] = by_slot[t];
}

}  // namespace

// tknnSegmentMin (sharded RT-DBSCAN, owlraytracing_amd/distributed.py): out[seg[i]] = min(out[seg[i]], val[i]) over the
// elements with seg[i] >= 0 -- the smallest global id of every local cluster, the one per-point step of the label propagation
// (everything after it works on clusters and halo rows).  Few segments and ten million elements is the hard case -- a
// scatter-reduce queues its atomics on a few dozen addresses (two ranks, BASELINE config 3: 160 ms of a 174 ms step) --; as in
// db_min_row the value only falls, so an element that reads a smaller or equal one is done, and a wave first settles the two
// segments most of its lanes hold with one atomic each.
__global__ void __launch_bounds__(kDbBlock) db_segment_min_kernel(const int32_t *seg, const long long *val, long long n, long long *out) {
  const long long i = (long long)blockIdx.x * kDbBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int32_t my_seg = -1;
  long long my_val = 0x7fffffffffffffffll;
  if (i < n) {
    my_seg = seg[i];
    if (my_seg >= 0) my_val = val[i];
  }
  bool pending = my_seg >= 0;
  if (pending && __hip_atomic_load(out + my_seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= my_val) pending = false;
  for (int round = 0; round < 2; round++) {
    const unsigned long long todo = __ballot(pending);
    if (!todo) break;
    const int j = __ffsll((long long)todo) - 1;
    const int32_t s_j = __shfl(my_seg, j);
    const bool same = pending && my_seg == s_j;
    long long m = same ? my_val : 0x7fffffffffffffffll;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const long long o = __shfl_xor(m, off);
      m = o < m ? o : m;
    }
    if (lane == j) atomicMin(out + s_j, m);
    pending = pending && !same;
  }
  if (pending) atomicMin(out + my_seg, my_val);
}
void db_segment_min(const int32_t *d_seg, const int64_t *d_val, int64_t n, int64_t *d_out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(db_segment_min_kernel, dim3((unsigned)((n + kDbBlock - 1) / kDbBlock)), dim3(kDbBlock), 0, s, d_seg, (const long long *)d_val, (long long)n,
                     (long long *)d_out);
  OWLMI_HIP(hipGetLastError());
}

// the work counters of the launches so far: the stripes to the host (synchronises the stream), summed into h_counters_[0..7]
void Engine::db_read_stats(hipStream_t s) {
  OWLMI_HIP(hipMemcpyAsync(h_counters_ + 16, counters_ + kCounters, kDbStripes * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < 8; i++) {
    unsigned long long sum = 0;
    for (int st = 0; st < kDbStripes; st++) sum += h_counters_[16 + st * 8 + i];
    h_counters_[i] = sum;
  }
}

// tknnDbscanNoise: one growth round of the auto-eps loop without the loop -- core flags, then the noise probe of every
// point that is not core.  Returns the number of noise points; d_noise is indexed by row.
int64_t Engine::dbscan_noise(float eps, int min_pts, uint8_t *d_noise, hipStream_t s) {
  const int64_t n = bvh_.size();
  size_t scan_bytes = 0;
  OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, (int)n, s));
  {
    size_t flag_scan_bytes = 0;
    OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, flag_scan_bytes, DbFlagIter(nullptr, DbFlagOf()), (int32_t *)nullptr, (int)n, s));
    scan_bytes = std::max(scan_bytes, flag_scan_bytes);
  }
  const size_t need = (((size_t)n * (4 + 4 + 4 + 4 + 1 + 1)) + 32 + 255) / 256 * 256;  // as dbscan_auto
  if (need + scan_bytes > wave_ws_bytes_) {
    if (wave_ws_) (void)hipFree(wave_ws_);
    wave_ws_ = nullptr;
    OWLMI_HIP(hipMalloc(&wave_ws_, need + scan_bytes));
    wave_ws_bytes_ = need + scan_bytes;
  }
  char *ws = (char *)wave_ws_;
  DbArgs a;
  std::memset(&a, 0, sizeof a);
  a.bvh = bvh_.view();
  a.block_paths = (getenv("TKNN_DB_PATHS") && atoi(getenv("TKNN_DB_PATHS")) == 0) ? nullptr : bvh_.block_paths_device();  // (0: measurements)
  a.min_pts = min_pts;
  a.eps = eps;
  a.eps_wide = eps * 1.000001f;
  a.eps_in2 = eps * eps * (1.0f - 1e-5f);
  a.eps_out2 = eps * eps * (1.0f + 1e-5f);
  a.parent = nullptr;            // no unions here: the slots' place holds ...
  a.near_node = (int32_t *)ws;   // ... where each point's noise probe starts (db_has_core_neighbour)
  int32_t *core_rank = (int32_t *)(ws + (size_t)n * 4);
  a.rank = (int32_t *)(ws + (size_t)n * 8);                   // n + 1 entries: flags, then positions
  int32_t *next_core = (int32_t *)(ws + (size_t)n * 12 + 4);  // n + 1 entries
  a.core_sorted = (uint8_t *)(ws + (size_t)n * 16 + 8);
  uint8_t *noise = a.core_sorted + n;
  a.next_core = next_core;
  a.stats = counters_ + kCounters;  // striped (db_add_stats)
  void *scan_tmp = ws + need;
  const unsigned blocks = (unsigned)((n + kDbBlock - 1) / kDbBlock), blocks1 = (unsigned)((n + 1 + kDbBlock - 1) / kDbBlock);
  OWLMI_HIP(hipMemsetAsync(counters_, 0, kCounterWords * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(db_core_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, (int32_t *)nullptr);
  {
    int32_t *pos = a.rank;
    OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, DbFlagIter(a.core_sorted, DbFlagOf()), core_rank, (int)n, s));
    hipLaunchKernelGGL(db_core_pos_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, core_rank, pos, (int32_t *)nullptr, (unsigned long long *)nullptr);
    hipLaunchKernelGGL(db_next_core_kernel, dim3(blocks1), dim3(kDbBlock), 0, s, a, core_rank, pos, next_core);
  }
  hipLaunchKernelGGL(db_noise_probe_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, noise, 1);
  hipLaunchKernelGGL(db_rows_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, noise, d_noise);
  OWLMI_HIP(hipGetLastError());
  db_read_stats(s);
  return (int64_t)h_counters_[6];
}

void Engine::dbscan(float eps, int min_pts, int32_t *d_labels, uint8_t *d_core, int32_t *d_counts,
                    tknnDbscanInfo *info, hipStream_t s, const int32_t *core_label) {
  const int64_t n = bvh_.size();
  // scratch: core flags per slot, parent, root flags, ranks, next_core (+ two sentinels), smallest rows
  const size_t min_row_at = ((size_t)n * 17 + 8 + 15) / 16 * 16;
  const size_t block_places_bytes = (((size_t)n / kDbBlock + 2) * 8 + 15) / 16 * 16;
  const size_t max_packets = (size_t)n / 64 + 1;  // of db_group_union_kernel (their diagnostic records: the diagnostic library only)
  const size_t need = (min_row_at + (size_t)n * 16 + block_places_bytes + (TKNN_DIAG_BUILD ? max_packets * 16 : 0) + 255) / 256 * 256;  // ... the list of the slots that are not core, their core neighbours
  size_t scan_bytes = 0;
  OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, (int)n, s));
  {
    size_t flag_scan_bytes = 0;
    OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, flag_scan_bytes, DbFlagIter(nullptr, DbFlagOf()), (int32_t *)nullptr, (int)n, s));
    scan_bytes = std::max(scan_bytes, flag_scan_bytes);
  }
  if (need + scan_bytes > wave_ws_bytes_) {
    if (wave_ws_) (void)hipFree(wave_ws_);
    wave_ws_ = nullptr;
    OWLMI_HIP(hipMalloc(&wave_ws_, need + scan_bytes));
    wave_ws_bytes_ = need + scan_bytes;
  }
  char *ws = (char *)wave_ws_;
  DbArgs a;
  std::memset(&a, 0, sizeof a);
  a.bvh = bvh_.view();
  a.block_paths = (getenv("TKNN_DB_PATHS") && atoi(getenv("TKNN_DB_PATHS")) == 0) ? nullptr : bvh_.block_paths_device();  // (0: measurements)
  a.eps = eps;
  a.eps_wide = eps * 1.000001f;
  a.min_pts = min_pts;
  a.want_counts = d_counts != nullptr;
  a.keep_core = 0;
  a.diag = (TKNN_DIAG_BUILD && getenv("TKNN_DB_DIAG")) ? atoi(getenv("TKNN_DB_DIAG")) : 0;  // the diagnostic library only
  a.parent = (int32_t *)ws;
  int32_t *is_root = (int32_t *)(ws + (size_t)n * 4);
  a.rank = (int32_t *)(ws + (size_t)n * 8);  // n + 1 entries
  int32_t *next_core = (int32_t *)(ws + (size_t)n * 12 + 4);  // n + 1 entries
  a.core_sorted = (uint8_t *)(ws + (size_t)n * 16 + 8);
  a.min_row = (int32_t *)(ws + min_row_at);
  int32_t *not_core = a.min_row + n;  // in slot order; its length in counters_[19]
  int32_t *uni = not_core + 2 * (size_t)n;  // db_uniform_kernel's per-node sets (behind border_lists)
  int32_t *block_places = uni + n;          // per workgroup of 256 slots: its listed groups; behind them, their places in the list
  int32_t *border_lists = not_core + n;  // `border_per` words per listed point, if they fit n words: a count and the core neighbours
  const int border_per = std::max(2, min_pts);
  const bool side = !core_label && !(getenv("TKNN_DB_SIDE") && atoi(getenv("TKNN_DB_SIDE")) == 0);  // (0: measurements without the side stream)
  if (side && !db_side_) {
    OWLMI_HIP(hipStreamCreateWithFlags(&db_side_, hipStreamNonBlocking));
    OWLMI_HIP(hipEventCreateWithFlags(&ev_side_a_, hipEventDisableTiming));
    OWLMI_HIP(hipEventCreateWithFlags(&ev_side_b_, hipEventDisableTiming));
  }
  a.next_core = next_core;
  a.eps_in2 = eps * eps * (1.0f - 1e-5f);
  a.eps_out2 = eps * eps * (1.0f + 1e-5f);
  void *scan_tmp = ws + need;
  a.core = d_core;
  a.counts = d_counts;
  a.labels = d_labels;
  const char *union_env = getenv("TKNN_DBSCAN_UNION");  // "point": the per-point union walk (A/B measurements, tests); read per call
  const bool per_point = db_force_point_ || (union_env && std::strcmp(union_env, "point") == 0);
  // every point's group, for the union pass; kept in the caller's label array, which is written last
  a.group_of = core_label || per_point ? nullptr : d_labels;
  const unsigned blocks = (unsigned)((n + kDbBlock - 1) / kDbBlock);
  a.diag_out = counters_ + 20;
  a.stats = counters_ + kCounters;  // striped (db_add_stats); [0..5]: node / point tests of the three traversal kernels
  if (db_side_pending_) {
    // a call that threw between its side launch and its rejoin has left db_border_walk_kernel running on the workspace
    // and the counters this call is about to reset: nothing of this call before that kernel has ended
    OWLMI_HIP(hipStreamWaitEvent(s, ev_side_b_, 0));
    db_side_pending_ = false;
  }
  OWLMI_HIP(hipMemsetAsync(counters_, 0, kCounterWords * sizeof(unsigned long long), s));  // ... [19]: length of the label pass's list
  const unsigned walk_grid = blocks < 2048u ? blocks : 2048u;  // grid-stride over lists whose lengths only the device knows
  hipEvent_t e0 = ev_a_, e1 = ev_b_;
  OWLMI_HIP(hipEventRecord(e0, s));
  if (core_label) {
    hipLaunchKernelGGL(db_core_from_labels_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, core_label);
  } else {
    // the flags BY ROW are written by the label kernel, which scatters to the rows anyway: a one-byte store at the
    // caller's row from this kernel cost a partial sector per point (rocprofv3: 426 MB written for 10 M points)
    DbArgs c = a;
    c.core = nullptr;
    hipLaunchKernelGGL(db_core_kernel, dim3(blocks), dim3(kDbBlock), 0, s, c, block_places);
  }
  OWLMI_HIP(hipEventRecord(ev_c_, s));  // end of the core-flag traversal
  {
    // next_core: flags -> exclusive sum (rank of a slot among the core slots) -> slot of the r-th core
    // point -> first core slot at or after each slot.  is_root / rank are free until the unions are done.
    int32_t *pos = a.rank;
    if (core_label) hipLaunchKernelGGL(db_flag_count_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, block_places);  // (else: db_core_kernel has counted)
    OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, block_places, block_places + blocks, (int)blocks, s));
    hipLaunchKernelGGL(db_core_pos_blocks_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, block_places + blocks, pos, core_label ? (int32_t *)nullptr : not_core, counters_ + 19);
    hipLaunchKernelGGL(db_next_core_blocks_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, block_places + blocks, pos, next_core);
  }
  if (side) {  // the walks of the points that are not core, beside everything up to the label kernel
    OWLMI_HIP(hipEventRecord(ev_side_a_, s));
    OWLMI_HIP(hipStreamWaitEvent(db_side_, ev_side_a_, 0));
    hipLaunchKernelGGL(db_border_walk_kernel, dim3(walk_grid), dim3(kDbBlock), 0, db_side_, a, not_core, counters_ + 19, border_lists, border_per, (long long)n);
    OWLMI_HIP(hipEventRecord(ev_side_b_, db_side_));
    db_side_pending_ = true;  // (until some stream waits for it: below, or at the top of the next call if this one throws in between)
  }
  if (core_label) {
    hipLaunchKernelGGL(db_assign_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, core_label);
    OWLMI_HIP(hipGetLastError());
    OWLMI_HIP(hipEventRecord(e1, s));
    OWLMI_HIP(hipStreamSynchronize(s));
    db_read_stats(s);
    if (info) {
      float ms = 0;
      OWLMI_HIP(hipEventElapsedTime(&ms, e0, e1));
      std::memset(info, 0, sizeof *info);
      info->clusters = -1;
      info->solve_ms = ms;
      info->label_ms = ms;
      info->node_tests = (int64_t)h_counters_[4];
      info->point_tests = (int64_t)h_counters_[5];
      info->label_point_tests = (int64_t)h_counters_[5];
    }
    return;
  }
  int union_launches = 1;
  bool between_passes = false;  // ev_g_ .. ev_h_: the cursors' reset and db_uniform_kernel, not part of union_ms
  if (per_point) {
    OWLMI_HIP(hipEventRecord(ev_d_, s));
    hipLaunchKernelGGL(db_union_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a);
  } else {
    // per-slot group references where the ranks go afterwards, the list itself (slot order = Morton order: neighbours in
    // the list are neighbours in space) where the root flags go; its length in counters_[8], the XCDs' cursors behind the statistics' stripes
    int32_t *group_at = a.rank, *groups = is_root;
    unsigned long long *n_groups = counters_ + 8;
    unsigned long long *cursors = counters_ + kCounters + kDbStripes * 8;  // (behind the statistics' stripes, kDbCursorStride apart)
    // (... and the count of stack overflows in [17]; all zero since the call's first memset)
    a.chunk = getenv("TKNN_DB_CHUNK") ? std::max(1, atoi(getenv("TKNN_DB_CHUNK"))) : 64;
    a.short_way = getenv("TKNN_DB_SHORT") ? atoi(getenv("TKNN_DB_SHORT")) : 1;
    a.scan_budget = getenv("TKNN_DB_SCAN") ? std::max(0, atoi(getenv("TKNN_DB_SCAN"))) : 12;
    hipLaunchKernelGGL(db_group_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, group_at, uni, a.min_row, block_places);
    OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, block_places, block_places + blocks, (int)blocks, s));
    hipLaunchKernelGGL(db_group_list_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, group_at, block_places + blocks, groups, n_groups);
    // persistent lanes: as many workgroups as the device holds at once (the list's length is known on the device only)
    // (per engine: the CU count and the occupancy are those of THIS engine's device)
    if (db_union_resident_ == 0) {
      int per_cu = 0;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device_) != hipSuccess) {
        db_union_resident_ = 256 * 4;
      } else {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)db_group_union_kernel, kDbUnionBlock, 0) != hipSuccess) per_cu = 4;
        // nine workgroups = 18 waves per CU are what fits (16 KB of LDS per workgroup = 13 granules of 1 280 bytes: the occupancy
        // query says ten).  BASELINE config 3, both passes, since the packet cursors have cache lines of their own (round 4):
        // 1.46 / 1.34 / 1.25 ms with seven / eight / nine; ten (a build with seven buffered unions per lane: 12 granules) 1.27
        // (TKNN_DB_PER_CU: measurements)
        db_union_resident_ = prop.multiProcessorCount * std::max(1, std::min(per_cu, getenv("TKNN_DB_PER_CU") ? atoi(getenv("TKNN_DB_PER_CU")) : 9));
      }
    }
    const int resident = db_union_resident_;
    unsigned grid = blocks < (unsigned)resident ? blocks : (unsigned)resident;
    if (getenv("TKNN_DB_GRID")) grid = std::max(1, std::min((int)grid, atoi(getenv("TKNN_DB_GRID"))));  // measurements
    // Two passes: first the pairs of groups that (nearly) touch -- in a dense region their probes hit at once, and when the
    // pass is over a cluster's groups are one set --, then the pairs further apart, nearly all of which then are in one
    // set already and cost a parent read instead of a probe that would have to look at many point pairs to find one close
    // enough (or none).  One pass settles everything as soon as it is met: 4 times the point tests, 14 ms instead of 4.
    const float split = getenv("TKNN_DB_SPLIT") ? (float)atof(getenv("TKNN_DB_SPLIT")) : 0.25f;
    // ONE number bounds both passes: pass 1 takes near2 <= near_hi2, pass 2 near2 > that same value, and the per-axis
    // reach of a pass's box prefilter is derived from its near_hi2 (>= its square root, DbArgs), so that no pair whose
    // faces are between split * eps_wide and sqrt(near_hi2) apart along one axis falls between the passes
    // (reach = split * eps (1 + 1e-6) against near2 <= split^2 eps^2 (1 + 1e-5) left a 34-ulp window).
    if (TKNN_DIAG_BUILD && (a.diag & 512)) {
      a.pk_diag = (int32_t *)((char *)block_places + block_places_bytes);
      OWLMI_HIP(hipMemsetAsync(a.pk_diag, 0, max_packets * 16, s));
    }
    a.near_lo2 = -1.f;
    a.near_hi2 = split * split * a.eps_out2;
    a.reach = db_reach_of(a.near_hi2);
    OWLMI_HIP(hipEventRecord(ev_d_, s));
    if (split < 1.f) {
      union_launches = 2;
      hipLaunchKernelGGL(db_group_union_kernel, dim3(grid), dim3(kDbUnionBlock), 0, s, a, groups, n_groups, cursors, n_groups + 9);
      OWLMI_HIP(hipMemsetAsync(cursors, 0, 8 * kDbCursorStride * sizeof(unsigned long long), s));
      a.near_lo2 = a.near_hi2;
      OWLMI_HIP(hipEventRecord(ev_g_, s));
      between_passes = true;
      if (n > 1 && !(getenv("TKNN_DB_UNIFORM") && atoi(getenv("TKNN_DB_UNIFORM")) == 0)) {  // (0: measurements without it)
        // which nodes hold one set only, now that the groups that touch are united (min_row's place is free until the unions are done)
        a.uni = uni;          // (filled by db_group_kernel)
        a.uni_leaf = a.rank;  // (the select has used the per-slot group references up)
        a.split_owner = bvh_.split_owner_device();
        hipLaunchKernelGGL(db_uniform_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, groups, n_groups);
      }
      OWLMI_HIP(hipEventRecord(ev_h_, s));
    }
    a.near_hi2 = a.eps_out2;
    a.reach = db_reach_of(a.near_hi2);
    hipLaunchKernelGGL(db_group_union_kernel, dim3(grid), dim3(kDbUnionBlock), 0, s, a, groups, n_groups, cursors, n_groups + 9);
  }
  OWLMI_HIP(hipEventRecord(ev_e_, s));
  if (per_point) OWLMI_HIP(hipMemsetAsync(a.min_row, 0x7f, (size_t)n * sizeof(int32_t), s));  // 0x7f7f7f7f: above every row (else: db_group_kernel)
  const unsigned blocks_per = (unsigned)(((n + kDbPer - 1) / kDbPer + kDbBlock - 1) / kDbBlock);  // kDbPer slots per thread
  hipLaunchKernelGGL(db_flatten_kernel, dim3(blocks_per), dim3(kDbBlock), 0, s, a, is_root);
  hipLaunchKernelGGL(db_root_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, is_root);
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, is_root, a.rank, (int)n, s));
  // number of clusters = rank[n-1] + is_root[n-1]
  int32_t last[2] = {0, 0};
  OWLMI_HIP(hipMemcpyAsync(&last[0], a.rank + (n - 1), 4, hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipMemcpyAsync(&last[1], is_root + (n - 1), 4, hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipEventRecord(ev_f_, s));
  if (side) {
    OWLMI_HIP(hipStreamWaitEvent(s, ev_side_b_, 0));
    db_side_pending_ = false;
  }
  // labels by slot first and a gather to the rows (default), or scattered straight to the rows (TKNN_DB_LABEL=scatter: A/B).
  // The caller's arrays must allow 16-byte stores for the gather (hipMalloc'd ones do).
  const char *label_env = getenv("TKNN_DB_LABEL");
  const int32_t *row_slot = bvh_.row_slot_device();
  const bool gather = row_slot && !(label_env && std::strcmp(label_env, "scatter") == 0) && ((uintptr_t)d_labels % 16 == 0) && (!d_core || (uintptr_t)d_core % 4 == 0);
  int32_t *by_slot = gather ? is_root : nullptr;  // (the root flags have been summed: their place is free)
  hipLaunchKernelGGL(db_label_kernel, dim3(walk_grid + blocks_per), dim3(kDbBlock), 0, s, a, not_core, counters_ + 19, (int)walk_grid,
                     side ? border_lists : (const int32_t *)nullptr, border_per, (long long)n, by_slot);
  if (gather) hipLaunchKernelGGL(db_rows_from_slots_kernel, dim3(blocks_per), dim3(kDbBlock), 0, s, a, row_slot, by_slot);
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipEventRecord(e1, s));
  OWLMI_HIP(hipMemcpyAsync(h_counters_ + 8, counters_ + 8, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));   // groups
  OWLMI_HIP(hipMemcpyAsync(h_counters_ + 9, counters_ + 17, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));  // stack overflows
  OWLMI_HIP(hipMemcpyAsync(h_counters_ + 10, counters_ + 19, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));  // slots that are not core
  db_read_stats(s);
  OWLMI_HIP(hipStreamSynchronize(s));
  if (!per_point && h_counters_[9] != 0 && !db_force_point_) {
    // a packet's walk ran out of stack (a tree some 250 levels deep): the whole call again with the per-point unions
    db_force_point_ = true;
    try {
      dbscan(eps, min_pts, d_labels, d_core, d_counts, info, s, core_label);
    } catch (...) {
      db_force_point_ = false;
      throw;
    }
    db_force_point_ = false;
    return;
  }
  if (info) {
    float ms = 0;
    std::memset(info, 0, sizeof *info);
    OWLMI_HIP(hipEventElapsedTime(&ms, e0, e1));
    info->clusters = last[0] + last[1];
    info->solve_ms = ms;
    OWLMI_HIP(hipEventElapsedTime(&info->core_ms, e0, ev_c_));
    OWLMI_HIP(hipEventElapsedTime(&info->union_ms, ev_d_, ev_e_));
    if (between_passes) {  // union_ms: the two launches of the union kernel
      float between = 0;
      OWLMI_HIP(hipEventElapsedTime(&between, ev_g_, ev_h_));
      info->union_ms -= between;
    }
    OWLMI_HIP(hipEventElapsedTime(&info->label_ms, ev_f_, e1));
    info->node_tests = (int64_t)(h_counters_[0] + h_counters_[2] + h_counters_[4]);
    info->point_tests = (int64_t)(h_counters_[1] + h_counters_[3] + h_counters_[5]);
    info->core_point_tests = (int64_t)h_counters_[1];
    info->union_point_tests = (int64_t)h_counters_[3];
    info->label_point_tests = (int64_t)h_counters_[5];
    info->union_launches = union_launches;
    info->union_node_tests = (int64_t)h_counters_[2];
    info->groups = per_point ? 0 : (int64_t)h_counters_[8];
    if ((a.diag & 512) && a.pk_diag) {
      // what the packets of the timed pass took, and when the launch would end if the waves took them longest first
      // (TKNN_DB_DUMP=<file>: four int32 per packet -- ticks, extent in 1e-6, rounds, settles -- for scripts/db_packet_stats.py)
      const size_t np = ((size_t)h_counters_[8] + 63) / 64;
      std::vector<int32_t> d(np * 4);
      OWLMI_HIP(hipMemcpy(d.data(), a.pk_diag, np * 16, hipMemcpyDeviceToHost));
      if (const char *dump = getenv("TKNN_DB_DUMP")) {
        if (FILE *f = std::fopen(dump, "wb")) {
          std::fwrite(d.data(), 16, np, f);
          std::fclose(f);
        }
      }
      std::vector<size_t> order(np);
      for (size_t i = 0; i < np; i++) order[i] = i;
      std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return d[x * 4] > d[y * 4]; });
      double sum = 0;
      for (size_t i = 0; i < np; i++) sum += d[i * 4];
      const int machines = (int)db_union_resident_ * (kDbUnionBlock / 64);
      std::vector<double> load(machines, 0.0);
      for (size_t i = 0; i < np; i++) load[std::min_element(load.begin(), load.end()) - load.begin()] += d[order[i] * 4];
      std::fprintf(stderr, "[dbscan] %zu packets, %d waves: sum %.3g ticks, mean per wave %.0f, longest packet %d; longest-first would end at %.0f ticks\n", np, machines, sum,
                   sum / machines, np ? d[order[0] * 4] : 0, *std::max_element(load.begin(), load.end()));
      for (int i = 0; i < 6 && (size_t)i < np; i++)
        std::fprintf(stderr, "[dbscan]   packet %zu: %d ticks, extent %.4f, %d rounds, %d settles\n", order[i], d[order[i] * 4], d[order[i] * 4 + 1] * 1e-6, d[order[i] * 4 + 2], d[order[i] * 4 + 3]);
    }
    if (a.diag & 512) {
      unsigned long long t[20];
      OWLMI_HIP(hipMemcpy(t, counters_ + 20, sizeof t, hipMemcpyDeviceToHost));
      std::fprintf(stderr, "[dbscan] union kernel: %llu waves, mean time in the kernel %.0f ticks, longest %llu ticks (%.0f %% of it filled on average), in packets %.0f ticks\n", t[15],
                   (double)t[13] / (double)t[15], t[14], 100.0 * (double)t[13] / (double)t[15] / (double)t[14], (double)(t[0] + t[1] + t[2] + t[3] + t[4]) / (double)t[15]);
      if (!(a.diag & 1024)) std::fprintf(stderr, "[dbscan] union kernel: longest packet %llu ticks; %llu packets above 2^20 ticks, %llu above 2^21\n", t[12], t[10], t[11]);
      if (!(a.diag & 1024) && t[11]) std::fprintf(stderr, "[dbscan] union kernel: the packets above 2^21 ticks, each on average: %.1f rounds, %.1f settles (%.0f ticks in them), %.1f entries the long way, %.1f point tests\n",
                   (double)t[16] / t[11], (double)t[17] / t[11], (double)t[9] / t[11], (double)t[18] / t[11], (double)t[19] / t[11]);
      const double tot = (double)(t[0] + t[1] + t[2] + t[3] + t[4]);
      std::fprintf(stderr, "[dbscan] union kernel wave time: loads %.1f%%  tests %.1f%%  settles %.1f%%  pushes %.1f%%  packet set-up %.1f%%;  %.1f rounds and %.1f settles per packet, %llu packet walks, %.1f us per packet walk (s_memtime at 100 MHz)\n",
                   100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot, 100 * t[4] / tot, (double)t[5] / (double)t[7], (double)t[6] / (double)t[7], t[7], tot / 100.0 / (double)t[7]);
    }
    if (getenv("TKNN_DB_VERBOSE"))
      std::fprintf(stderr, "[dbscan] node / point tests: core flags %llu / %llu, unions %llu / %llu, labels %llu / %llu; %llu slots not core\n", h_counters_[0],
                   h_counters_[1], h_counters_[2], h_counters_[3], h_counters_[4], h_counters_[5], h_counters_[10]);
    if (a.diag & 1024) {
      unsigned long long t[5];
      OWLMI_HIP(hipMemcpy(t, counters_ + 28, sizeof t, hipMemcpyDeviceToHost));
      std::fprintf(stderr, "[dbscan] second union pass: %llu packets of one set, %llu mixed; %llu nodes popped, %llu of them dropped as the packet's own set, %llu known to be of one set\n", t[0], t[1], t[3], t[2], t[4]);
    }
    if (a.diag & 8)
      std::fprintf(stderr, "[dbscan] groups %llu  union-phase node tests %llu  longest walk %llu  mean of the waves' longest %.0f\n", h_counters_[8],
                   h_counters_[2], h_counters_[6], (double)h_counters_[7] / ((double)((h_counters_[8] + 63) / 64) + 1e-9));
  }
}


// tknnDbscanAuto: the growth loop of the spec -- rounds of (core flags of the points not core yet, noise probe of the
// points still noise) with eps doubling, then ONE full clustering at the eps that brought the noise under the bound.
void Engine::dbscan_auto(float eps0, int min_pts, double max_noise, int max_rounds, int32_t *d_labels, uint8_t *d_core,
                         tknnDbscanAutoInfo *info, hipStream_t s) {
  const int64_t n = bvh_.size();
  const int64_t bound = (int64_t)std::floor(max_noise * (double)n);
  // scratch of the probe rounds: parent (unused but written by the core kernel), rank / flags, core ranks, next_core,
  // core flags per slot, noise flags per slot
  size_t scan_bytes = 0;
  OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, (int)n, s));
  {
    size_t flag_scan_bytes = 0;
    OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, flag_scan_bytes, DbFlagIter(nullptr, DbFlagOf()), (int32_t *)nullptr, (int)n, s));
    scan_bytes = std::max(scan_bytes, flag_scan_bytes);
  }
  const size_t need = (((size_t)n * (4 + 4 + 4 + 4 + 1 + 1)) + 32 + 255) / 256 * 256;
  if (need + scan_bytes > wave_ws_bytes_) {
    if (wave_ws_) (void)hipFree(wave_ws_);
    wave_ws_ = nullptr;
    OWLMI_HIP(hipMalloc(&wave_ws_, need + scan_bytes));
    wave_ws_bytes_ = need + scan_bytes;
  }
  char *ws = (char *)wave_ws_;
  DbArgs a;
  std::memset(&a, 0, sizeof a);
  a.bvh = bvh_.view();
  a.block_paths = (getenv("TKNN_DB_PATHS") && atoi(getenv("TKNN_DB_PATHS")) == 0) ? nullptr : bvh_.block_paths_device();  // (0: measurements)
  a.min_pts = min_pts;
  a.keep_core = 1;
  a.parent = nullptr;           // no unions in the growth rounds: the slots' place holds ...
  a.near_node = (int32_t *)ws;  // ... where each point's noise probe starts (db_has_core_neighbour)
  int32_t *core_rank = (int32_t *)(ws + (size_t)n * 4);
  a.rank = (int32_t *)(ws + (size_t)n * 8);                     // n + 1 entries: flags, then positions
  int32_t *next_core = (int32_t *)(ws + (size_t)n * 12 + 4);  // n + 1 entries
  a.core_sorted = (uint8_t *)(ws + (size_t)n * 16 + 8);
  uint8_t *noise = a.core_sorted + n;
  a.next_core = next_core;
  a.stats = counters_ + kCounters;  // striped (db_add_stats)
  void *scan_tmp = ws + need;
  const unsigned blocks = (unsigned)((n + kDbBlock - 1) / kDbBlock), blocks1 = (unsigned)((n + 1 + kDbBlock - 1) / kDbBlock);
  OWLMI_HIP(hipMemsetAsync(a.core_sorted, 0, (size_t)n, s));
  OWLMI_HIP(hipEventRecord(ev_a_, s));
  float eps = eps0;
  int rounds = 0;
  int64_t noise_now = n;
  bool reached = false;
  for (int t = 0; t < max_rounds; t++) {
    a.eps = eps;
    a.eps_wide = eps * 1.000001f;
    a.eps_in2 = eps * eps * (1.0f - 1e-5f);
    a.eps_out2 = eps * eps * (1.0f + 1e-5f);
    OWLMI_HIP(hipMemsetAsync(counters_, 0, kCounterWords * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(db_core_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, (int32_t *)nullptr);
    {
      int32_t *pos = a.rank;
        OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, DbFlagIter(a.core_sorted, DbFlagOf()), core_rank, (int)n, s));
      hipLaunchKernelGGL(db_core_pos_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, core_rank, pos, (int32_t *)nullptr, (unsigned long long *)nullptr);
      hipLaunchKernelGGL(db_next_core_kernel, dim3(blocks1), dim3(kDbBlock), 0, s, a, core_rank, pos, next_core);
    }
    hipLaunchKernelGGL(db_noise_probe_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, noise, t == 0 ? 1 : 0);
    OWLMI_HIP(hipGetLastError());
    db_read_stats(s);
    rounds = t + 1;
    noise_now = (int64_t)h_counters_[6];
    if (noise_now <= bound) {
      reached = true;
      break;
    }
    if (t + 1 < max_rounds) eps = eps * 2.0f;  // hostCode.cpp:321
  }
  OWLMI_HIP(hipEventRecord(ev_b_, s));
  OWLMI_HIP(hipEventSynchronize(ev_b_));
  float probe_ms = 0;
  OWLMI_HIP(hipEventElapsedTime(&probe_ms, ev_a_, ev_b_));
  // the clusters of the final eps (also when the rounds ran out: the caller gets the last round's labelling and an error)
  tknnDbscanInfo last;
  std::memset(&last, 0, sizeof last);
  dbscan(eps, min_pts, d_labels, d_core, nullptr, &last, s);
  if (info) {
    std::memset(info, 0, sizeof *info);
    info->last = last;
    info->rounds = rounds;
    info->eps = eps;
    info->noise = noise_now;
    info->probe_ms = probe_ms;
  }
  if (!reached) throw RoundsExceeded{};
}

}  // namespace owlmi
