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

#include <cstring>

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
  int32_t *parent;       // per caller index
  int32_t *rank;         // per caller index: cluster label of a root
  int32_t *labels;       // per caller index
  const int32_t *next_core;  // per sorted slot (+1 sentinel): first core slot at or after it, n if none
  float eps_in2, eps_out2;   // eps^2 (1 -+ 1e-5): below / above these, fp32 distance arithmetic cannot disagree
  // work counters, per traversal kernel k (0 core flags, 1 unions, 2 labels / assign): [2k] tree nodes
  // tested, [2k + 1] points whose distance to a query was computed (12 algorithmic bytes each)
  unsigned long long *stats;
};

// a workgroup's counts into the kernel's two counters: wave sums, one LDS atomic per wave, one global
// atomic per workgroup (every thread of the workgroup must call it)
__device__ __forceinline__ void db_add_stats(unsigned long long *stats, unsigned long long *blk, uint32_t nodes, uint32_t points) {
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

// squared distances from q to the farthest and the nearest point of a box
__device__ __forceinline__ void box_dist2(const LbvhNode &nd, const LbvhPoint &q, float &far2, float &near2) {
  const float ax = fmaxf(fabsf(q.x - nd.lo[0]), fabsf(q.x - nd.hi[0])), ay = fmaxf(fabsf(q.y - nd.lo[1]), fabsf(q.y - nd.hi[1])),
              az = fmaxf(fabsf(q.z - nd.lo[2]), fabsf(q.z - nd.hi[2]));
  const float bx = fmaxf(fmaxf(nd.lo[0] - q.x, q.x - nd.hi[0]), 0.f), by = fmaxf(fmaxf(nd.lo[1] - q.y, q.y - nd.hi[1]), 0.f),
              bz = fmaxf(fmaxf(nd.lo[2] - q.z, q.z - nd.hi[2]), 0.f);
  far2 = ax * ax + ay * ay + az * az;
  near2 = bx * bx + by * by + bz * bz;
}
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
    if (g != p) __hip_atomic_store(parent + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path halving
    x = p;
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
// f(representative's row) for every tight node that has a core point within eps of q (its first core
// point stands for all of them) and for every core point within eps reached as a leaf.  `own_slot`
// (or -1) names q's own sorted slot: the tight node holding it is skipped.
// `settled(row)` may say that the group a tight node's first core point stands for needs no look
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
      node_tests++;
      const bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) & (q.y <= nd.hi[1] + r) &
                       (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
      if (!hit) {
        ref = bvh.rope_node[ref];
        continue;
      }
      const bool probing = probe_rep >= 0;
      if (probing || node_is_tight(nd, a.eps_in2)) {
        const int32_t first = lbvh_first(ref, nd.other), last = lbvh_last(ref, nd.other);
        const int32_t s = a.next_core[first];
        if (s > last || (first <= own_slot && own_slot <= last)) {  // no core point here / my own group
          ref = bvh.rope_node[ref];
          continue;
        }
        if (!probing && settled(bvh.prim_id[s])) {
          ref = bvh.rope_node[ref];
          continue;
        }
        float far2, near2;
        box_dist2(nd, q, far2, near2);
        if (far2 <= a.eps_in2) {  // every point of the node is within eps
          f(probing ? probe_rep : bvh.prim_id[s]);
          ref = probing ? probe_exit : bvh.rope_node[ref];
          probe_rep = -1;
          continue;
        }
        if (near2 > a.eps_out2) {  // none is
          ref = bvh.rope_node[ref];
          continue;
        }
        if (!probing) {
          probe_rep = bvh.prim_id[s];
          probe_exit = bvh.rope_node[ref];
        }
      }
      ref = lbvh_left_ref(ref, nd);
    } else {
      const int32_t slot = ~ref;
      if (a.core_sorted[slot] && slot != own_slot) {
        const LbvhPoint p = bvh.points[slot];
        point_tests++;
        if (knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z)) <= a.eps) {
          if (probe_rep >= 0) {
            f(probe_rep);
            ref = probe_exit;
            probe_rep = -1;
            continue;
          }
          f(bvh.prim_id[slot]);
        }
      }
      ref = bvh.rope_leaf[slot];
    }
  }
}

__device__ __forceinline__ void db_core_body(const DbArgs &a, int32_t t, uint32_t &node_tests, uint32_t &point_tests) {
  const LbvhView &bvh = a.bvh;
  if (a.keep_core && a.core_sorted[t]) return;
  const LbvhPoint q = bvh.points[t];
  int32_t cnt = 0;
  const int stop_at = a.want_counts ? 0x7fffffff : a.min_pts;
  const int32_t clean_end = bvh.n - (bvh.nan_count ? *bvh.nan_count : 0);  // NaN points sort last
  const float r = a.eps_wide;
  int32_t ref = bvh.root;
  if (!a.want_counts) {
    // the first tight node on my own root path: its points are pairwise within eps, so if it holds
    // minPts of them I am core without looking any further
    int32_t node = bvh.root;
    while (node >= 0) {
      const LbvhNode nd = bvh.nodes[node];
      if (node_is_tight(nd, a.eps_in2)) {
        const int32_t first = lbvh_first(node, nd.other), last = lbvh_last(node, nd.other);
        if (last < clean_end && last - first + 1 >= a.min_pts) {
          cnt = last - first + 1;
          ref = LBVH_END;
        }
        break;
      }
      node = t <= nd.split ? lbvh_left_ref(node, nd) : lbvh_right_ref(node, nd);
    }
  }
  while (ref != LBVH_END && cnt < stop_at) {
    if (ref >= 0) {
      const LbvhNode nd = bvh.nodes[ref];
      node_tests++;
      const bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) & (q.y <= nd.hi[1] + r) &
                       (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
      if (hit) {
        // a node inside the sphere is counted, not walked
        float far2, near2;
        box_dist2(nd, q, far2, near2);
        const int32_t last = lbvh_last(ref, nd.other);
        if (far2 <= a.eps_in2 && last < clean_end) {
          cnt += last - lbvh_first(ref, nd.other) + 1;
          ref = bvh.rope_node[ref];
          continue;
        }
      }
      ref = hit ? lbvh_left_ref(ref, nd) : bvh.rope_node[ref];
    } else {
      const int32_t slot = ~ref;
      const LbvhPoint p = bvh.points[slot];
      point_tests++;
      if (knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z)) <= a.eps) cnt++;
      ref = bvh.rope_leaf[slot];
    }
  }
  const uint8_t is_core = cnt >= a.min_pts;
  a.core_sorted[t] = is_core;
  // results and the union-find are indexed by ROW (the point's position in the caller's buffer,
  // prim_id of the sorted slot), not by the id an engine built with tknnBuildIds reports
  const int32_t row = bvh.prim_id[t];
  if (a.core) a.core[row] = is_core;
  if (a.counts) a.counts[row] = cnt;
  a.parent[row] = row;
}

__global__ void __launch_bounds__(kDbBlock) db_core_kernel(DbArgs a) {
  __shared__ unsigned long long blk_stats[2];
  if (threadIdx.x < 2) blk_stats[threadIdx.x] = 0ull;
  __syncthreads();
  uint32_t node_tests = 0, point_tests = 0;
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n) db_core_body(a, t, node_tests, point_tests);
  db_add_stats(a.stats + 0, blk_stats, node_tests, point_tests);
}

// next_core[s] = first core slot >= s (n if none): with rank[s] = number of core slots before s (an
// exclusive sum of the flags) and pos[r] = slot of the r-th core point, next_core[s] = pos[rank[s]]
__global__ void __launch_bounds__(kDbBlock) db_core_flag_kernel(DbArgs a, int32_t *flag) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n) flag[t] = a.core_sorted[t];
}
__global__ void __launch_bounds__(kDbBlock) db_core_pos_kernel(DbArgs a, const int32_t *rank, int32_t *pos) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n && a.core_sorted[t]) pos[rank[t]] = t;
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
  const int32_t row = bvh.prim_id[t];
  // the first tight node on my own root path: its core points are one cluster, held together by its
  // first core point, which also stands for me in the unions below
  int32_t mine = row;
  int32_t node = bvh.root;
  while (node >= 0) {
    const LbvhNode nd = bvh.nodes[node];
    node_tests++;
    if (node_is_tight(nd, a.eps_in2)) {
      const int32_t s = a.next_core[lbvh_first(node, nd.other)];  // <= t: I am core and inside
      if (s != t) {
        mine = bvh.prim_id[s];
        uf_unite(a.parent, row, mine);
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
        if (other == mine || other == row) return;
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

// after the unions: point every core at its root and flag roots for the ranking scan
__global__ void __launch_bounds__(kDbBlock) db_flatten_kernel(DbArgs a, int32_t *is_root) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t >= a.bvh.n) return;
  const int32_t id = a.bvh.prim_id[t];
  int32_t flag = 0;
  if (a.core_sorted[t]) {
    const int32_t root = uf_find(a.parent, id);
    flag = root == id;
    a.labels[id] = root;  // temporary: root index, replaced by its rank in db_label_kernel
  }
  is_root[id] = flag;
}

__global__ void __launch_bounds__(kDbBlock) db_label_kernel(DbArgs a) {
  __shared__ unsigned long long blk_stats[2];
  if (threadIdx.x < 2) blk_stats[threadIdx.x] = 0ull;
  __syncthreads();
  uint32_t node_tests = 0, point_tests = 0;
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t < a.bvh.n) {
    const LbvhPoint q = a.bvh.points[t];
    const int32_t row = a.bvh.prim_id[t];
    int32_t root = -1;
    if (a.core_sorted[t]) {
      root = uf_find(a.parent, row);
    } else {
      for_each_core_group(
          a, q, -1, [](int32_t) { return false; },
          [&](int32_t other) {
            const int32_t r = uf_find(a.parent, other);
            if (root < 0 || r < root) root = r;
          },
          node_tests, point_tests);
    }
    a.labels[row] = root < 0 ? -1 : a.rank[root];
  }
  db_add_stats(a.stats + 4, blk_stats, node_tests, point_tests);
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
            const int32_t l = core_label[other];  // a tight node's core points share one label: its first stands for all
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
__device__ __forceinline__ bool db_has_core_neighbour(const DbArgs &a, const LbvhPoint &q, uint32_t &node_tests, uint32_t &point_tests) {
  const LbvhView &bvh = a.bvh;
  const float r = a.eps_wide;
  int32_t ref = bvh.root;
  while (ref != LBVH_END) {
    if (ref >= 0) {
      const LbvhNode nd = bvh.nodes[ref];
      node_tests++;
      const bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) & (q.y <= nd.hi[1] + r) &
                       (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
      if (!hit || a.next_core[lbvh_first(ref, nd.other)] > lbvh_last(ref, nd.other)) {  // out of reach, or no core point below
        ref = bvh.rope_node[ref];
        continue;
      }
      float far2, near2;
      box_dist2(nd, q, far2, near2);
      if (far2 <= a.eps_in2) return true;  // all of it within eps, and a core point among it
      if (near2 > a.eps_out2) {
        ref = bvh.rope_node[ref];
        continue;
      }
      ref = lbvh_left_ref(ref, nd);
    } else {
      const int32_t slot = ~ref;
      if (a.core_sorted[slot]) {
        const LbvhPoint p = bvh.points[slot];
        point_tests++;
        if (knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z)) <= a.eps) return true;
      }
      ref = bvh.rope_leaf[slot];
    }
  }
  return false;
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
      still = db_has_core_neighbour(a, a.bvh.points[t], node_tests, point_tests) ? 0u : 1u;
      noise[t] = (uint8_t)still;
    }
  }
  db_add_stats(a.stats + 0, blk_stats, node_tests, point_tests);
  db_add_stats(a.stats + 6, blk_noise, still, 0u);
}

}  // namespace

void Engine::dbscan(float eps, int min_pts, int32_t *d_labels, uint8_t *d_core, int32_t *d_counts,
                    tknnDbscanInfo *info, hipStream_t s, const int32_t *core_label) {
  const int64_t n = bvh_.size();
  // scratch: core flags per slot, parent, root flags, ranks
  const size_t need = (((size_t)n * (1 + 4 + 4 + 4 + 4)) + 16 + 255) / 256 * 256;  // + next_core, + two sentinels
  size_t scan_bytes = 0;
  OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, (int)n, s));
  if (need + scan_bytes > wave_ws_bytes_) {
    if (wave_ws_) (void)hipFree(wave_ws_);
    wave_ws_ = nullptr;
    OWLMI_HIP(hipMalloc(&wave_ws_, need + scan_bytes));
    wave_ws_bytes_ = need + scan_bytes;
  }
  char *ws = (char *)wave_ws_;
  DbArgs a;
  a.bvh = bvh_.view();
  a.eps = eps;
  a.eps_wide = eps * 1.000001f;
  a.min_pts = min_pts;
  a.want_counts = d_counts != nullptr;
  a.keep_core = 0;
  a.parent = (int32_t *)ws;
  int32_t *is_root = (int32_t *)(ws + (size_t)n * 4);
  a.rank = (int32_t *)(ws + (size_t)n * 8);  // n + 1 entries
  int32_t *next_core = (int32_t *)(ws + (size_t)n * 12 + 4);  // n + 1 entries
  a.core_sorted = (uint8_t *)(ws + (size_t)n * 16 + 8);
  a.next_core = next_core;
  a.eps_in2 = eps * eps * (1.0f - 1e-5f);
  a.eps_out2 = eps * eps * (1.0f + 1e-5f);
  void *scan_tmp = ws + need;
  a.core = d_core;
  a.counts = d_counts;
  a.labels = d_labels;
  const unsigned blocks = (unsigned)((n + kDbBlock - 1) / kDbBlock);
  a.stats = counters_;  // [0..5]: node / point tests of the three traversal kernels
  OWLMI_HIP(hipMemsetAsync(counters_, 0, 6 * sizeof(unsigned long long), s));
  hipEvent_t e0 = ev_a_, e1 = ev_b_;
  OWLMI_HIP(hipEventRecord(e0, s));
  if (core_label) {
    hipLaunchKernelGGL(db_core_from_labels_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, core_label);
  } else {
    hipLaunchKernelGGL(db_core_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a);
  }
  OWLMI_HIP(hipEventRecord(ev_c_, s));  // end of the core-flag traversal
  {
    // next_core: flags -> exclusive sum (rank of a slot among the core slots) -> slot of the r-th core
    // point -> first core slot at or after each slot.  is_root / rank are free until the unions are done.
    int32_t *flag = a.rank, *core_rank = is_root, *pos = a.rank;
    const unsigned blocks1 = (unsigned)((n + 1 + kDbBlock - 1) / kDbBlock);
    hipLaunchKernelGGL(db_core_flag_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, flag);
    OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, flag, core_rank, (int)n, s));
    OWLMI_HIP(hipMemsetAsync(pos, 0x7f, ((size_t)n + 1) * sizeof(int32_t), s));  // 0x7f7f7f7f: "none", clamped below
    hipLaunchKernelGGL(db_core_pos_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, core_rank, pos);
    hipLaunchKernelGGL(db_next_core_kernel, dim3(blocks1), dim3(kDbBlock), 0, s, a, core_rank, pos, next_core);
  }
  if (core_label) {
    hipLaunchKernelGGL(db_assign_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, core_label);
    OWLMI_HIP(hipGetLastError());
    OWLMI_HIP(hipEventRecord(e1, s));
    OWLMI_HIP(hipStreamSynchronize(s));
    OWLMI_HIP(hipMemcpyAsync(h_counters_, counters_, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    OWLMI_HIP(hipStreamSynchronize(s));
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
  OWLMI_HIP(hipEventRecord(ev_d_, s));
  hipLaunchKernelGGL(db_union_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a);
  OWLMI_HIP(hipEventRecord(ev_e_, s));
  hipLaunchKernelGGL(db_flatten_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, is_root);
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, is_root, a.rank, (int)n, s));
  OWLMI_HIP(hipEventRecord(ev_f_, s));
  hipLaunchKernelGGL(db_label_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a);
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipEventRecord(e1, s));
  OWLMI_HIP(hipMemcpyAsync(h_counters_, counters_, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  // number of clusters = rank[n-1] + is_root[n-1]
  int32_t last[2] = {0, 0};
  OWLMI_HIP(hipMemcpyAsync(&last[0], a.rank + (n - 1), 4, hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipMemcpyAsync(&last[1], is_root + (n - 1), 4, hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipStreamSynchronize(s));
  if (info) {
    float ms = 0;
    std::memset(info, 0, sizeof *info);
    OWLMI_HIP(hipEventElapsedTime(&ms, e0, e1));
    info->clusters = last[0] + last[1];
    info->solve_ms = ms;
    OWLMI_HIP(hipEventElapsedTime(&info->core_ms, e0, ev_c_));
    OWLMI_HIP(hipEventElapsedTime(&info->union_ms, ev_d_, ev_e_));
    OWLMI_HIP(hipEventElapsedTime(&info->label_ms, ev_f_, e1));
    info->node_tests = (int64_t)(h_counters_[0] + h_counters_[2] + h_counters_[4]);
    info->point_tests = (int64_t)(h_counters_[1] + h_counters_[3] + h_counters_[5]);
    info->core_point_tests = (int64_t)h_counters_[1];
    info->union_point_tests = (int64_t)h_counters_[3];
    info->label_point_tests = (int64_t)h_counters_[5];
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
  a.min_pts = min_pts;
  a.keep_core = 1;
  a.parent = (int32_t *)ws;
  int32_t *core_rank = (int32_t *)(ws + (size_t)n * 4);
  a.rank = (int32_t *)(ws + (size_t)n * 8);                     // n + 1 entries: flags, then positions
  int32_t *next_core = (int32_t *)(ws + (size_t)n * 12 + 4);  // n + 1 entries
  a.core_sorted = (uint8_t *)(ws + (size_t)n * 16 + 8);
  uint8_t *noise = a.core_sorted + n;
  a.next_core = next_core;
  a.stats = counters_;
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
    OWLMI_HIP(hipMemsetAsync(counters_, 0, 8 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(db_core_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a);
    {
      int32_t *flag = a.rank, *pos = a.rank;
      hipLaunchKernelGGL(db_core_flag_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, flag);
      OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, flag, core_rank, (int)n, s));
      OWLMI_HIP(hipMemsetAsync(pos, 0x7f, ((size_t)n + 1) * sizeof(int32_t), s));
      hipLaunchKernelGGL(db_core_pos_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, core_rank, pos);
      hipLaunchKernelGGL(db_next_core_kernel, dim3(blocks1), dim3(kDbBlock), 0, s, a, core_rank, pos, next_core);
    }
    hipLaunchKernelGGL(db_noise_probe_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, noise, t == 0 ? 1 : 0);
    OWLMI_HIP(hipGetLastError());
    OWLMI_HIP(hipMemcpyAsync(h_counters_, counters_, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    OWLMI_HIP(hipStreamSynchronize(s));
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
