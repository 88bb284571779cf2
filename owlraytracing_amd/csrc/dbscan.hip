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
// Union-find reads/writes go through agent-scope atomics: a workgroup's L1 (and another XCD's L2)
// would otherwise keep serving a stale parent and a failed CAS could retry forever.
#include "trueknn_engine.h"

#include <hipcub/hipcub.hpp>

namespace owlmi {
namespace {

constexpr int kDbBlock = 256;

struct DbArgs {
  LbvhView bvh;
  float eps;
  float eps_wide;  // eps * (1 + 1e-6): the box prefilter must not cut what the rounded sphere test accepts
  int min_pts;
  int want_counts;
  uint8_t *core_sorted;  // per sorted slot
  uint8_t *core;         // per caller index (may be null)
  int32_t *counts;       // per caller index (may be null)
  int32_t *parent;       // per caller index
  int32_t *rank;         // per caller index: cluster label of a root
  int32_t *labels;       // per caller index
};

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

// visits every point within the sphere; f(point, slot) returns false to stop early
template <typename F>
__device__ __forceinline__ void for_each_neighbour(const DbArgs &a, const LbvhPoint &q, F f) {
  const LbvhView &bvh = a.bvh;
  const float r = a.eps_wide;
  int32_t ref = bvh.root;
  while (ref != LBVH_END) {
    if (ref >= 0) {
      const LbvhNode nd = bvh.nodes[ref];
      const bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) & (q.y <= nd.hi[1] + r) &
                       (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
      ref = hit ? lbvh_left_ref(ref, nd) : bvh.rope_node[ref];
    } else {
      const int32_t slot = ~ref;
      const LbvhPoint p = bvh.points[slot];
      const float d = knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z));
      if (d <= a.eps)
        if (!f(p, slot)) return;
      ref = bvh.rope_leaf[slot];
    }
  }
}

__global__ void __launch_bounds__(kDbBlock) db_core_kernel(DbArgs a) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t >= a.bvh.n) return;
  const LbvhPoint q = a.bvh.points[t];
  int32_t cnt = 0;
  const int stop_at = a.want_counts ? 0x7fffffff : a.min_pts;
  for_each_neighbour(a, q, [&](const LbvhPoint &, int32_t) {
    cnt++;
    return cnt < stop_at;
  });
  const uint8_t is_core = cnt >= a.min_pts;
  a.core_sorted[t] = is_core;
  if (a.core) a.core[q.id] = is_core;
  if (a.counts) a.counts[q.id] = cnt;
  a.parent[q.id] = q.id;
}

__global__ void __launch_bounds__(kDbBlock) db_union_kernel(DbArgs a) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t >= a.bvh.n || !a.core_sorted[t]) return;
  const LbvhPoint q = a.bvh.points[t];
  for_each_neighbour(a, q, [&](const LbvhPoint &p, int32_t slot) {
    if (p.id < q.id && a.core_sorted[slot]) uf_unite(a.parent, q.id, p.id);
    return true;
  });
}

// after the unions: point every core at its root and flag roots for the ranking scan
__global__ void __launch_bounds__(kDbBlock) db_flatten_kernel(DbArgs a, int32_t *is_root) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t >= a.bvh.n) return;
  const int32_t id = a.bvh.points[t].id;
  int32_t flag = 0;
  if (a.core_sorted[t]) {
    const int32_t root = uf_find(a.parent, id);
    flag = root == id;
    a.labels[id] = root;  // temporary: root index, replaced by its rank in db_label_kernel
  }
  is_root[id] = flag;
}

__global__ void __launch_bounds__(kDbBlock) db_label_kernel(DbArgs a) {
  const int32_t t = blockIdx.x * kDbBlock + threadIdx.x;
  if (t >= a.bvh.n) return;
  const LbvhPoint q = a.bvh.points[t];
  int32_t root = -1;
  if (a.core_sorted[t]) {
    root = uf_find(a.parent, q.id);
  } else {
    for_each_neighbour(a, q, [&](const LbvhPoint &p, int32_t slot) {
      if (a.core_sorted[slot]) {
        const int32_t r = uf_find(a.parent, p.id);
        if (root < 0 || r < root) root = r;
      }
      return true;
    });
  }
  a.labels[q.id] = root < 0 ? -1 : a.rank[root];
}

}  // namespace

void Engine::dbscan(float eps, int min_pts, int32_t *d_labels, uint8_t *d_core, int32_t *d_counts,
                    tknnDbscanInfo *info, hipStream_t s) {
  const int64_t n = bvh_.size();
  // scratch: core flags per slot, parent, root flags, ranks
  const size_t need = (((size_t)n * (1 + 4 + 4 + 4)) + 255) / 256 * 256;
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
  a.parent = (int32_t *)ws;
  int32_t *is_root = (int32_t *)(ws + (size_t)n * 4);
  a.rank = (int32_t *)(ws + (size_t)n * 8);
  a.core_sorted = (uint8_t *)(ws + (size_t)n * 12);
  void *scan_tmp = ws + need;
  a.core = d_core;
  a.counts = d_counts;
  a.labels = d_labels;
  const unsigned blocks = (unsigned)((n + kDbBlock - 1) / kDbBlock);
  hipEvent_t e0 = ev_a_, e1 = ev_b_;
  OWLMI_HIP(hipEventRecord(e0, s));
  hipLaunchKernelGGL(db_core_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a);
  hipLaunchKernelGGL(db_union_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a);
  hipLaunchKernelGGL(db_flatten_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a, is_root);
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, is_root, a.rank, (int)n, s));
  hipLaunchKernelGGL(db_label_kernel, dim3(blocks), dim3(kDbBlock), 0, s, a);
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipEventRecord(e1, s));
  // number of clusters = rank[n-1] + is_root[n-1]
  int32_t last[2] = {0, 0};
  OWLMI_HIP(hipMemcpyAsync(&last[0], a.rank + (n - 1), 4, hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipMemcpyAsync(&last[1], is_root + (n - 1), 4, hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipStreamSynchronize(s));
  if (info) {
    float ms = 0;
    OWLMI_HIP(hipEventElapsedTime(&ms, e0, e1));
    info->clusters = last[0] + last[1];
    info->solve_ms = ms;
  }
}

}  // namespace owlmi
