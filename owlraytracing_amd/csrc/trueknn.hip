// trueknn.hip -- the TrueKNN engine behind include/owlknn.h: LBVH build, the per-round "lane"
// kernel (one query per lane, stackless rope traversal) and the result writers.  The persistent
// wave-packet kernel lives in trueknn_wave.hip.
//
// Reference functions this file replaces (samples/s01-trueknn):
//   deviceCode.cu:140-153  __raygen__rayGen            -> active test + point query per lane
//   owl_device.h:150-174   optixTrace (RT cores)       -> rope traversal of the LBVH
//   deviceCode.cu:62-138   __intersection__Spheres     -> box test + register k-list insert
//   hostCode.cpp:285-340   round loop                  -> Engine::solve_lane
#include "knn_thresholds.h"  // knn_gate_from_worst
#include "trueknn_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace owlmi {

namespace {

constexpr int kLaneBlock = 256;
constexpr int kCountedSubtree = 32;  // smallest subtree the lane kernel tries to count instead of walking

struct LaneRoundArgs {
  LbvhView bvh;
  LbvhView halo;           // second point set searched by every query (n == 0: none)
  int level;               // 0-based radius level of this launch
  int32_t *out_level;      // n, caller order (may be null)
  float radius;
  int k;
  uint8_t *done;           // per sorted slot
  uint8_t *tie;            // per sorted slot: (1 + level) | 0x80 for rows finished with exact-distance ties (knn_flag_tie)
  int32_t *tie_list;
  const int32_t *next_level;  // per sorted slot: first level this query takes part in (may be null)
  int64_t *isect_sorted;   // per sorted slot, accumulated over rounds
  int32_t *out_idx;        // n*k, caller order (may be null)
  float *out_dist;         // n*k (may be null)
  int64_t *out_isect;      // n (may be null)
  tknnNeigh *out_fb;       // n*k (may be null)
  unsigned long long *counters;  // [0] unfinished, [1] node tests, [2] point tests, [3] sum isect, [4] active lanes, [5], [6] candidates of the active lanes, plain and squared
};

template <int K>
__device__ __forceinline__ void write_row(const LaneRoundArgs &a, int32_t row, const KList<K> &list,
                                          int64_t isect) {
  const int k = a.k;
  const int64_t base = (int64_t)row * k;
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (j < k) {
      int32_t prim = knn_key_prim(list.key[j]);
      float d = knn_key_dist(list.key[j]);
      if (a.out_idx) a.out_idx[base + j] = prim;
      if (a.out_dist) a.out_dist[base + j] = d;
      if (a.out_fb) {
        // final state of the reference's frameBuffer: slot 0 carries numNeighbors (0 = done) and
        // the intersection counter, other slots keep their initial {k, 0}  (deviceCode.cu:74,118)
        tknnNeigh e;
        e.ind = prim;
        e.dist = d;
        e.numNeighbors = j == 0 ? 0 : k;
        e.pad_ = 0;
        e.intersections = j == 0 ? isect : 0;
        a.out_fb[base + j] = e;
      }
    }
  }
  if (a.out_isect) a.out_isect[row] = isect;
  if (a.out_level) a.out_level[row] = a.level;
}

// SUBTREES: count fully covered subtrees beyond the gate instead of walking them (see below).  In a
// wave some lane is at a large node most of the time, so the test is paid on most steps: launches
// whose boxes hold few points run without it (Engine::lane_rounds decides per round).
template <int K, bool SUBTREES>
__global__ void __launch_bounds__(kLaneBlock) lane_round_kernel(LaneRoundArgs a) {
  const int32_t t = blockIdx.x * kLaneBlock + threadIdx.x;
  const LbvhView &bvh = a.bvh;
  // deviceCode.cu:148: finished queries launch no ray.  Inactive lanes fall through to the
  // wave-wide counter reduction at the end instead of returning.
  const bool active = t < bvh.n && !a.done[t] && (!a.next_level || a.next_level[t] <= a.level);
  LbvhPoint q = {0.f, 0.f, 0.f, -1};
  if (active) q = bvh.points[t];
  const float r = a.radius;
  KList<K> list;
  list.clear();
  int32_t cnt = 0, others = 0;
  uint32_t node_tests = 0, point_tests = 0;
  float tau2 = INFINITY;  // squared-distance gate from the list's last entry: beyond it nothing enters
  for (int tree = 0; tree < 2; tree++) {
    const LbvhView &tv = tree == 0 ? a.bvh : a.halo;
    if (tv.n <= 0) continue;
    const int32_t clean_end = tv.n - (tv.nan_count ? *tv.nan_count : 0);  // NaN points sort last
    int32_t ref = active ? tv.root : LBVH_END;
    while (ref != LBVH_END) {
      if (ref >= 0) {
        const LbvhNode nd = tv.nodes[ref];
        node_tests++;
        // conservative: any point p in the node has lo <= c_p <= hi, and fp32 rounding is monotone,
        // so fl(c_p - r) >= fl(lo - r) and fl(c_p + r) <= fl(hi + r)
        bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) &
                   (q.y <= nd.hi[1] + r) & (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
        // The same monotonicity the other way round: if even the largest centre passes the lower
        // test and the smallest the upper one, EVERY point of the node is a candidate
        // (deviceCode.cu:74 would count each).  If, besides, none of them can enter the list any
        // more -- the node lies beyond the gate -- the subtree is counted, not walked: a query
        // whose box has grown over a whole cluster costs O(log n) instead of O(cluster).
        // Only tried for subtrees of at least kCountedSubtree points: near the leaves the test would
        // cost as much as the node test itself and save nothing.
        const int32_t first = lbvh_first(ref, nd.other), last = lbvh_last(ref, nd.other);
        const bool inside = SUBTREES && hit && last - first + 1 >= kCountedSubtree && (nd.hi[0] - r <= q.x) & (q.x <= nd.lo[0] + r) &
                                       (nd.hi[1] - r <= q.y) & (q.y <= nd.lo[1] + r) & (nd.hi[2] - r <= q.z) & (q.z <= nd.lo[2] + r);
        if (inside) {
          const float gx = fmaxf(fmaxf(nd.lo[0] - q.x, q.x - nd.hi[0]), 0.f);
          const float gy = fmaxf(fmaxf(nd.lo[1] - q.y, q.y - nd.hi[1]), 0.f);
          const float gz = fmaxf(fmaxf(nd.lo[2] - q.z, q.z - nd.hi[2]), 0.f);
          const float m2 = (gx * gx + gy * gy) + gz * gz;  // <= every point's squared distance, up to rounding
          // 5e-6 covers the roundings of m2 and of the points' own distance arithmetic; a node that
          // holds the query itself has m2 = 0 and is never skipped, so `others` stays right
          if (m2 * 0.999995f > tau2 && last < clean_end) {
            const int32_t c = last - first + 1;
            cnt += c;
            others += c;
            ref = tv.rope_node[ref];
            continue;
          }
        }
        ref = hit ? lbvh_left_ref(ref, nd) : tv.rope_node[ref];
      } else {
        const int32_t slot = ~ref;
        const LbvhPoint p = tv.points[slot];
        point_tests++;
        if (knn_in_box(p.x, p.y, p.z, r, q.x, q.y, q.z)) {
          cnt++;                 // deviceCode.cu:74
          if (p.id != q.id) {    // deviceCode.cu:103
            others++;
            float d = knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z));
            list.insert(knn_key(d, p.id));
            if (SUBTREES) tau2 = knn_gate_from_worst(knn_key_dist(list.worst()));
          }
        }
        ref = tv.rope_leaf[slot];
      }
    }
  }
  int64_t isect = 0;
  bool finished = false;
  if (active) {
    isect = a.isect_sorted[t] + cnt;
    a.isect_sorted[t] = isect;
    finished = others >= a.k;  // k insertions happened <=> numNeighbors reached 0
    if (finished) {
      a.done[t] = 1;
      write_row<K>(a, bvh.prim_id[t], list, isect);
      if (list.has_ties(a.k)) knn_flag_tie(a.tie, a.tie_list, a.counters, t, a.level);
    }
  }
  // wave-aggregated counters (all 64 lanes are here)
  const bool waiting = t < bvh.n && !a.done[t] && !active;  // handed over at a later level
  unsigned long long unfinished = __popcll(__ballot((active && !finished) || waiting));
  unsigned long long traced = __popcll(__ballot(active));
  unsigned long long nt = node_tests, pt = point_tests, si = finished ? (unsigned long long)isect : 0ull;
  // candidates of the active lanes, plain and squared (capped): sum c^2 / sum c is the box population
  // a random candidate TEST of this round saw, the figure that says where the round's work was
  const unsigned long long cc = active ? (unsigned long long)min(cnt, 65535) : 0ull;
  unsigned long long sc = cc, sc2 = cc * cc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nt += __shfl_xor(nt, off);
    pt += __shfl_xor(pt, off);
    si += __shfl_xor(si, off);
    sc += __shfl_xor(sc, off);
    sc2 += __shfl_xor(sc2, off);
  }
  // the workgroup's sums in LDS, then one atomic per counter and workgroup on the workgroup's stripe (kStatBase, knn_device.h)
  __shared__ unsigned long long blk_sum[7];
  if (threadIdx.x < 7) blk_sum[threadIdx.x] = 0ull;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    if (unfinished) atomicAdd(&blk_sum[0], unfinished);
    if (nt) atomicAdd(&blk_sum[1], nt);
    if (pt) atomicAdd(&blk_sum[2], pt);
    if (si) atomicAdd(&blk_sum[3], si);
    if (traced) atomicAdd(&blk_sum[4], traced);
    if (sc) {
      atomicAdd(&blk_sum[5], sc);
      atomicAdd(&blk_sum[6], sc2);
    }
  }
  __syncthreads();
  if (threadIdx.x < 7 && blk_sum[threadIdx.x])
    atomicAdd(&a.counters[kStatBase + (blockIdx.x & (kStatStripes - 1)) * kStatStride + threadIdx.x], blk_sum[threadIdx.x]);
}

template <int K>
void launch_lane(const LaneRoundArgs &a, bool subtrees, hipStream_t s) {
  unsigned blocks = (unsigned)((a.bvh.n + kLaneBlock - 1) / kLaneBlock);
  if (subtrees)
    hipLaunchKernelGGL((lane_round_kernel<K, true>), dim3(blocks), dim3(kLaneBlock), 0, s, a);
  else
    hipLaunchKernelGGL((lane_round_kernel<K, false>), dim3(blocks), dim3(kLaneBlock), 0, s, a);
}

// ---- exact-kNN repair (opt-in, never part of tknnSolve) ----------------
// The reference's rows are box-candidate kNN: a query that finished with radius r_q only ever saw
// points inside its L-inf box, so when its k-th distance d_k exceeds r_q a closer point may sit
// outside the box (15-20 % of rows on uniform data).  Every true neighbour has
// Euclidean distance <= d_k, hence lies in the box of half-width d_k: one more traversal with
// that radius, same (dist, index) order, gives the exact row.
struct RepairArgs {
  LbvhView bvh, halo;
  int k;
  float start_radius;
  const int32_t *levels;  // per caller row: level at which the query finished
  int32_t *idx;           // n*k rows, read (d_k) and rewritten
  float *dist;
  unsigned long long *counters;  // [0] rows repaired
};

template <int K>
__global__ void __launch_bounds__(kLaneBlock) repair_kernel(RepairArgs a) {
  const int32_t t = blockIdx.x * kLaneBlock + threadIdx.x;
  if (t >= a.bvh.n) return;
  const LbvhPoint q = a.bvh.points[t];
  const int32_t row = a.bvh.prim_id[t];
  const int32_t level = a.levels[row];
  if (level < 0) return;
  float rq = a.start_radius;
  for (int i = 0; i < level; i++) rq *= 2;
  const float dk = a.dist[(int64_t)row * a.k + a.k - 1];
  if (!(dk > rq)) return;  // the ball of radius d_k is inside the box already searched
  const float r = dk * 1.000001f;  // the rounded box test must not cut a point at distance d_k
  KList<K> list;
  list.clear();
  for (int tree = 0; tree < 2; tree++) {
    const LbvhView &tv = tree == 0 ? a.bvh : a.halo;
    if (tv.n <= 0) continue;
    int32_t ref = tv.root;
    while (ref != LBVH_END) {
      if (ref >= 0) {
        const LbvhNode nd = tv.nodes[ref];
        const bool hit = (nd.lo[0] - r <= q.x) & (q.x <= nd.hi[0] + r) & (nd.lo[1] - r <= q.y) &
                         (q.y <= nd.hi[1] + r) & (nd.lo[2] - r <= q.z) & (q.z <= nd.hi[2] + r);
        ref = hit ? lbvh_left_ref(ref, nd) : tv.rope_node[ref];
      } else {
        const int32_t slot = ~ref;
        const LbvhPoint p = tv.points[slot];
        if (p.id != q.id && knn_in_box(p.x, p.y, p.z, r, q.x, q.y, q.z))
          list.insert(knn_key(knn_sqrt(knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z)), p.id));
        ref = tv.rope_leaf[slot];
      }
    }
  }
  const int64_t base = (int64_t)row * a.k;
#pragma unroll
  for (int j = 0; j < K; j++)
    if (j < a.k) {
      a.idx[base + j] = knn_key_prim(list.key[j]);
      a.dist[base + j] = knn_key_dist(list.key[j]);
    }
  atomicAdd(&a.counters[0], 1ull);
}

template <int K>
void launch_repair(const RepairArgs &a, hipStream_t s) {
  unsigned blocks = (unsigned)((a.bvh.n + kLaneBlock - 1) / kLaneBlock);
  hipLaunchKernelGGL(repair_kernel<K>, dim3(blocks), dim3(kLaneBlock), 0, s, a);
}

}  // namespace

int list_capacity_for(int k) {
  static const int caps[] = {1, 2, 4, 5, 8, 10, 16, 24, 32, 64};
  for (int c : caps)
    if (k <= c) return c;
  return -1;
}

Engine::Engine() {
  OWLMI_HIP(hipGetDevice(&device_));
  OWLMI_HIP(hipEventCreate(&ev_a_));
  OWLMI_HIP(hipEventCreate(&ev_b_));
  OWLMI_HIP(hipEventCreate(&ev_c_));
  OWLMI_HIP(hipEventCreate(&ev_d_));
  OWLMI_HIP(hipEventCreate(&ev_e_));
  OWLMI_HIP(hipEventCreate(&ev_f_));
  OWLMI_HIP(hipEventCreate(&ev_g_));
  OWLMI_HIP(hipEventCreate(&ev_h_));
  OWLMI_HIP(hipMalloc((void **)&counters_, kCounterWords * sizeof(unsigned long long)));  // (the team kernel's per-XCD packet counters sit in the stripes' words, 32 words apart: trueknn_team.hip) [32]: tie rows
  OWLMI_HIP(hipMalloc((void **)&tie_list_, kTieListCap * sizeof(int32_t)));
  OWLMI_HIP(hipHostMalloc((void **)&h_counters_, (16 + std::max(kDbStripes * 8, kStatStripes * kStatStride)) * sizeof(unsigned long long)));
  if (const char *e = getenv("TKNN_WAVE_FORCE_REDO")) wave_force_redo_ = atoi(e) != 0;
  if (const char *e = getenv("TKNN_LEAF_MAX")) {
    int v = atoi(e);
    if (v >= 1 && v <= 64) wave_leaf_max_ = v;
  }
}

Engine::~Engine() {
  if (done_) (void)hipFree(done_);
  if (isect_sorted_) (void)hipFree(isect_sorted_);
  if (next_level_) (void)hipFree(next_level_);
  if (tie_) (void)hipFree(tie_);
  if (boundary_) (void)hipFree(boundary_);
  if (tie_list_) (void)hipFree(tie_list_);
  if (counters_) (void)hipFree(counters_);
  if (halo_mask_) (void)hipFree(halo_mask_);
  if (slot_list_) (void)hipFree(slot_list_);
  if (h_counters_) (void)hipHostFree(h_counters_);
  if (wave_ws_) (void)hipFree(wave_ws_);
  if (ev_a_) (void)hipEventDestroy(ev_a_);
  if (ev_b_) (void)hipEventDestroy(ev_b_);
  if (ev_c_) (void)hipEventDestroy(ev_c_);
  if (ev_d_) (void)hipEventDestroy(ev_d_);
  if (ev_e_) (void)hipEventDestroy(ev_e_);
  if (ev_f_) (void)hipEventDestroy(ev_f_);
  if (ev_g_) (void)hipEventDestroy(ev_g_);
  if (ev_h_) (void)hipEventDestroy(ev_h_);
  if (ev_side_a_) (void)hipEventDestroy(ev_side_a_);
  if (ev_side_b_) (void)hipEventDestroy(ev_side_b_);
  if (db_side_) (void)hipStreamDestroy(db_side_);
}

void Engine::set_halo(const float *d_xyz, const int32_t *d_ids, int64_t m, hipStream_t s) {
  halo_n_ = 0;
  if (m <= 0) return;
  halo_.build_from_points(d_xyz, m, s, d_ids);
  halo_n_ = m;
}

LbvhView Engine::halo_view() const {
  if (halo_count() > 0) return halo_.view();
  LbvhView v;
  std::memset(&v, 0, sizeof v);
  v.root = LBVH_END;
  return v;
}

void Engine::build(const float *d_xyz, const int32_t *d_ids, int64_t n, tknnBuildInfo *info, hipStream_t s) {
  OWLMI_HIP(hipEventRecord(ev_a_, s));
  halo_n_ = 0;
  boundary_valid_ = false;
  // Per-slot solve state first: if one of these allocations fails (a 100 M-point rebuild on a full
  // card) the engine must not be left "built" with null state arrays behind a stale capacity.
  if (n > state_cap_) {
    state_cap_ = 0;
    if (done_) (void)hipFree(done_);
    if (isect_sorted_) (void)hipFree(isect_sorted_);
    if (next_level_) (void)hipFree(next_level_);
    if (tie_) (void)hipFree(tie_);
    if (boundary_) (void)hipFree(boundary_);
    boundary_ = nullptr;
    tie_ = nullptr;
    done_ = nullptr;
    isect_sorted_ = nullptr;
    next_level_ = nullptr;
    try {
      OWLMI_HIP(hipMalloc((void **)&done_, (size_t)n));
      OWLMI_HIP(hipMalloc((void **)&isect_sorted_, (size_t)n * sizeof(int64_t)));
      OWLMI_HIP(hipMalloc((void **)&next_level_, (size_t)n * sizeof(int32_t)));
      OWLMI_HIP(hipMalloc((void **)&tie_, ((size_t)n + 3) & ~(size_t)3));  // (whole words: knn_flag_tie's atomics work on the byte's word)
      OWLMI_HIP(hipMalloc((void **)&boundary_, (size_t)n));
    } catch (...) {
      bvh_.clear();  // unbuilt: tknnSolve then answers TKNN_E_STATE instead of launching on null arrays
      throw;
    }
    state_cap_ = n;
  }
  try {
    bvh_.build_from_points(d_xyz, n, s, d_ids);
  } catch (...) {
    bvh_.clear();
    throw;
  }
  OWLMI_HIP(hipEventRecord(ev_b_, s));
  OWLMI_HIP(hipMemcpyAsync(scene_, bvh_.scene_device(), 6 * sizeof(float), hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipStreamSynchronize(s));
  OWLMI_HIP(hipEventSynchronize(ev_b_));
  if (info) {
    float ms = 0;
    OWLMI_HIP(hipEventElapsedTime(&ms, ev_a_, ev_b_));
    info->build_ms = ms;
    info->device_bytes = (int64_t)bvh_.device_bytes();
    info->n = (int32_t)n;
  }
}

// points a candidate box of this radius holds at the mean density of the built set (a work estimate)
double Engine::expected_box_population(float radius) const {
  double measure = 1.0;
  int dims = 0;
  for (int a = 0; a < 3; a++) {
    const double e = (double)scene_[3 + a] - (double)scene_[a];
    if (e > 0) {
      measure *= e;
      dims++;
    }
  }
  if (dims == 0 || !(measure > 0)) return (double)bvh_.size();
  return (double)bvh_.size() / measure * std::pow(2.0 * (double)radius, dims);
}

void Engine::solve_lane(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s) { lane_rounds(sa, 0, true, info, s); }

void Engine::continue_lane(const SolveArgs &sa, int first_level, tknnSolveInfo *info, hipStream_t s) {
  lane_rounds(sa, first_level, false, info, s);
}

void Engine::lane_rounds(const SolveArgs &sa, int first_level, bool fresh, tknnSolveInfo *info, hipStream_t s) {
  const int64_t n = bvh_.size();
  const int cap = list_capacity_for(sa.k);
  if (fresh) {
    OWLMI_HIP(hipMemsetAsync(done_, 0, (size_t)n, s));
    OWLMI_HIP(hipMemsetAsync(isect_sorted_, 0, (size_t)n * sizeof(int64_t), s));
    if (sa.d_levels) OWLMI_HIP(hipMemsetAsync(sa.d_levels, 0xff, (size_t)n * sizeof(int32_t), s));
  }
  OWLMI_HIP(hipMemsetAsync(counters_, 0, 16 * sizeof(unsigned long long), s));
  static_assert(kStatBase == owlmi::kStatBase && kStatStripes == owlmi::kStatStripes && kStatStride == owlmi::kStatStride, "one layout (knn_device.h)");
  reset_stat_stripes(s);
  LaneRoundArgs a;
  a.bvh = bvh_.view();
  a.halo = halo_view();
  a.out_level = sa.d_levels;
  a.k = sa.k;
  a.done = done_;
  a.tie = tie_;
  a.tie_list = tie_list_;
  a.isect_sorted = isect_sorted_;
  a.next_level = fresh ? nullptr : next_level_;
  a.out_idx = sa.d_idx;
  a.out_dist = sa.d_dist;
  a.out_isect = sa.d_isect;
  a.out_fb = sa.d_fb;
  a.counters = counters_;
  float radius = sa.start_radius, total_ms = 0;
  int rounds = first_level, launches = 0;
  for (int t = 0; t < first_level; t++) radius *= 2;
  double predicted_candidates = expected_box_population(radius);
  const double growth = expected_box_population(2.0f) / std::max(expected_box_population(1.0f), 1e-300);  // 2^dims
  unsigned long long c1_before = 0, c2_before = 0;
  for (;;) {
    if (rounds >= sa.max_rounds) {
      if (sa.allow_unfinished) break;
      throw RoundsExceeded{};
    }
    a.level = rounds;
    rounds++;
    launches++;
    a.radius = radius;
    // Subtree counting pays when boxes hold hundreds of points.  Stragglers handed over by the team
    // kernel are exactly those queries; otherwise predict from the round before -- the population of
    // the box an average candidate test worked in, times 2^dims -- and for the first round from
    // the scene's mean density.
    const bool subtrees = !fresh || predicted_candidates >= 256.0;
    // (the kernel's counters are striped: [0] of every stripe = unfinished is reset per round, the others accumulate over the rounds)
    OWLMI_HIP(hipMemset2DAsync(counters_ + kStatBase, kStatStride * sizeof(unsigned long long), 0, sizeof(unsigned long long), kStatStripes, s));
    OWLMI_HIP(hipEventRecord(ev_a_, s));
    switch (cap) {
      case 1: launch_lane<1>(a, subtrees, s); break;
      case 2: launch_lane<2>(a, subtrees, s); break;
      case 4: launch_lane<4>(a, subtrees, s); break;
      case 5: launch_lane<5>(a, subtrees, s); break;
      case 8: launch_lane<8>(a, subtrees, s); break;
      case 10: launch_lane<10>(a, subtrees, s); break;
      case 16: launch_lane<16>(a, subtrees, s); break;
      case 24: launch_lane<24>(a, subtrees, s); break;
      case 32: launch_lane<32>(a, subtrees, s); break;
      default: launch_lane<64>(a, subtrees, s); break;
    }
    OWLMI_HIP(hipGetLastError());
    OWLMI_HIP(hipEventRecord(ev_b_, s));
    // hostCode.cpp:310-330: the host decides about another round from the result state
    fetch_stat_stripes(s);
    OWLMI_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 7; i++) {
      h_counters_[i] = 0;
      for (int j = 0; j < kStatStripes; j++) h_counters_[i] += h_counters_[16 + j * kStatStride + i];
    }
    float ms = 0;
    OWLMI_HIP(hipEventElapsedTime(&ms, ev_a_, ev_b_));
    total_ms += ms;
    if (h_counters_[0] == 0) break;
    if (rounds >= sa.max_rounds && sa.allow_unfinished) break;
    {
      // counters [5], [6] accumulate over rounds: this round's share is the difference
      const double c1 = (double)(h_counters_[5] - c1_before), c2 = (double)(h_counters_[6] - c2_before);
      c1_before = h_counters_[5];
      c2_before = h_counters_[6];
      predicted_candidates = (c1 > 0 ? c2 / c1 : 0.0) * growth;
    }
    radius *= 2;  // hostCode.cpp:321 (fp32)
  }
  if (info) {
    info->rounds = rounds;
    info->final_radius = radius;
    info->node_tests = (int64_t)h_counters_[1];
    info->point_tests = (int64_t)h_counters_[2];
    info->total_intersections = (int64_t)h_counters_[3];
    info->total_active_rounds = (int64_t)h_counters_[4];
    info->solve_ms = total_ms;
    info->dominant_kernel_ms = total_ms / std::max(launches, 1);
    info->dominant_kernel_launches = launches;
    info->kernel_used = TKNN_KERNEL_LANE;
    info->list_capacity = cap;
    info->unfinished = (int64_t)h_counters_[0];
  }
}

int64_t Engine::repair_exact(int k, float start_radius, const int32_t *d_levels, int32_t *d_idx, float *d_dist,
                             hipStream_t s) {
  RepairArgs a;
  a.bvh = bvh_.view();
  a.halo = halo_view();
  a.k = k;
  a.start_radius = start_radius;
  a.levels = d_levels;
  a.idx = d_idx;
  a.dist = d_dist;
  a.counters = counters_;
  OWLMI_HIP(hipMemsetAsync(counters_, 0, sizeof(unsigned long long), s));
  switch (list_capacity_for(k)) {
    case 1: launch_repair<1>(a, s); break;
    case 2: launch_repair<2>(a, s); break;
    case 4: launch_repair<4>(a, s); break;
    case 5: launch_repair<5>(a, s); break;
    case 8: launch_repair<8>(a, s); break;
    case 10: launch_repair<10>(a, s); break;
    case 16: launch_repair<16>(a, s); break;
    case 24: launch_repair<24>(a, s); break;
    case 32: launch_repair<32>(a, s); break;
    default: launch_repair<64>(a, s); break;
  }
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipMemcpyAsync(h_counters_, counters_, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipStreamSynchronize(s));
  return (int64_t)h_counters_[0];
}

// tknnSolveOptions.phase == 3: per sorted slot, 1 for the queries an earlier call left unfinished (level -1 at their row)
__global__ void __launch_bounds__(256) unfinished_mask_kernel(const int32_t *levels, const int32_t *prim_id, int64_t n, uint8_t *mask) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) mask[t] = levels[prim_id[t]] < 0 ? 1 : 0;
}

void Engine::solve(const SolveArgs &sa, int kernel, tknnSolveInfo *info, hipStream_t s) {
  if (bigk_supports(sa.k)) {
    // 64 < k <= TKNN_MAX_K: the team walk with the lists in memory, rows final (three-word keys: no tie pass)
    if (kernel != TKNN_KERNEL_AUTO && kernel != TKNN_KERNEL_TEAM)
      throw ArgError{TKNN_E_UNSUPPORTED, "the lane and wave kernels keep their lists in registers: k <= 64 (k up to TKNN_MAX_K: TKNN_KERNEL_AUTO or TKNN_KERNEL_TEAM)"};
    if (sa.d_start_radii && halo_count() > 0)
      throw ArgError{TKNN_E_UNSUPPORTED, "tknnSolveEx: per-query start radii and a halo tree do not combine (the halo is exchanged for ONE radius)"};
    if (sa.phase != 0) {
      if (sa.phase == 3) {
        if (!sa.d_levels) throw ArgError{TKNN_E_ARG, "tknnSolveEx: phase 3 (unfinished queries only) needs the d_levels of the call that left them"};
        const int64_t n = bvh_.size();
        hipLaunchKernelGGL(unfinished_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sa.d_levels, bvh_.view().prim_id, n, boundary_);
        OWLMI_HIP(hipGetLastError());
        boundary_valid_ = false;
      } else if (!boundary_valid_) {
        throw ArgError{TKNN_E_STATE, "tknnSolveEx: phase 1 / 2 need a tknnHaloSelect count pass since the last build (it marks the boundary queries)"};
      }
    }
    struct HaloOffBig {
      bool &flag;
      explicit HaloOffBig(bool &f, bool on) : flag(f) { flag = on; }
      ~HaloOffBig() { flag = false; }
    } halo_off_big(ignore_halo_, sa.phase == 1);
    solve_bigk(sa, info, s);
    return;
  }
  if (kernel == TKNN_KERNEL_TEAM && !team_kernel_supports(sa.k))
    throw ArgError{TKNN_E_UNSUPPORTED, "the team kernels hold up to four neighbours per lane of a 16-lane team: k <= 64"};
  if (kernel == TKNN_KERNEL_AUTO) {
    // team kernels for every k the engine takes (<= 64): they hand what it cannot hold (outliers, dense duplicates, start radii
    // far too large) to lane rounds or the wave kernel by itself; measured fastest from r0 = 2e-5 to
    // r0 = 0.04 on 10 M uniform points and on the clustered sets of profiles/
    if (team_kernel_supports(sa.k))
      kernel = TKNN_KERNEL_TEAM;
    else
      kernel = wave_kernel_available() ? TKNN_KERNEL_WAVE : TKNN_KERNEL_LANE;
  }
  // Rows whose order depends on how bit-identical fp32 distances are ordered: every kernel lists by
  // (dist, index) and flags them in tie_; fix_ties redoes them in the reference's order, by the round in
  // which each neighbour was first a candidate (deviceCode.cu:77-85 -- lists persist over rounds).
  if (sa.d_start_radii && (kernel != TKNN_KERNEL_TEAM || bvh_.size() >= (1ll << 28)))
    throw ArgError{TKNN_E_UNSUPPORTED, "tknnSolveEx: per-query start radii are served by the team kernels only (k <= 64)"};
  if (sa.d_start_radii && halo_count() > 0)
    throw ArgError{TKNN_E_UNSUPPORTED, "tknnSolveEx: per-query start radii and a halo tree do not combine (the halo is exchanged for ONE radius)"};
  if (sa.phase != 0) {
    if (kernel != TKNN_KERNEL_TEAM || bvh_.size() >= (1ll << 28))
      throw ArgError{TKNN_E_UNSUPPORTED, "tknnSolveEx: phases (interior / boundary / unfinished queries) are served by the team kernels only"};
    if (sa.phase == 3) {
      // the queries an earlier call (allow_unfinished) left without a row: d_levels[row] < 0, marked per sorted slot
      if (!sa.d_levels) throw ArgError{TKNN_E_ARG, "tknnSolveEx: phase 3 (unfinished queries only) needs the d_levels of the call that left them"};
      const int64_t n = bvh_.size();
      hipLaunchKernelGGL(unfinished_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sa.d_levels, bvh_.view().prim_id, n, boundary_);
      OWLMI_HIP(hipGetLastError());
      boundary_valid_ = false;  // (the marks of the last tknnHaloSelect are gone)
    } else if (!boundary_valid_) {
      throw ArgError{TKNN_E_STATE, "tknnSolveEx: phase 1 / 2 need a tknnHaloSelect count pass since the last build (it marks the boundary queries)"};
    }
  }
  struct HaloOff {  // phase 1 runs beside tknnSetHalo: it must not look at the halo tree
    bool &flag;
    explicit HaloOff(bool &f, bool on) : flag(f) { flag = on; }
    ~HaloOff() { flag = false; }
  } halo_off(ignore_halo_, sa.phase == 1);
  tknnSolveInfo mine;
  std::memset(&mine, 0, sizeof mine);
  bool solved = false;
  ties_early_ = false;
  if (kernel == TKNN_KERNEL_TEAM) {
    solved = solve_team(sa, &mine, s);  // (resets tie_ and the tie counters together with its own state, one launch)
    if (!solved) kernel = TKNN_KERNEL_WAVE;  // n >= 2^28
  }
  if (!solved) {
    OWLMI_HIP(hipMemsetAsync(tie_, 0, (size_t)bvh_.size(), s));
    OWLMI_HIP(hipMemsetAsync(counters_ + kTieCounter, 0, 3 * sizeof(unsigned long long), s));
    if (kernel == TKNN_KERNEL_WAVE)
      solve_wave(sa, &mine, s);
    else
      solve_lane(sa, &mine, s);
  }
  fix_ties(sa, &mine, s);
  if (info) *info = mine;
}

}  // namespace owlmi

// ------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------
using owlmi::Engine;

static thread_local std::string g_last_error;

struct tknnEngine_t {
  Engine impl;
};

template <typename F>
static int guarded(F &&f) {
  try {
    f();
    return TKNN_OK;
  } catch (const owlmi::HipError &e) {
    g_last_error = e.what;
    return TKNN_E_HIP;
  } catch (const owlmi::RoundsExceeded &) {
    g_last_error = "max_rounds reached with unfinished queries (the reference loops forever here, e.g. n <= k)";
    return TKNN_E_ROUNDS;
  } catch (const owlmi::ArgError &e) {
    g_last_error = e.what;
    return e.code;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return TKNN_E_HIP;
  }
}

// Every engine call runs on the device the engine was created on (the caller's current device at
// tknnCreate), whatever device is current in the calling thread now; the caller's choice is restored.
struct DeviceScope {
  int prev = -1, want;
  explicit DeviceScope(int device) : want(device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != want && hipSetDevice(want) != hipSuccess) throw owlmi::HipError{"hipSetDevice(engine's device) failed"};
  }
  ~DeviceScope() {
    if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
  }
};

template <typename F>
static int guarded_on(tknnEngine e, F &&f) {
  return guarded([&] {
    DeviceScope scope(e->impl.device());
    f();
  });
}

extern "C" {

const char *tknnLastError(void) { return g_last_error.c_str(); }

int tknnDeviceCount(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int tknnCreate(tknnEngine *out) {
  if (!out) {
    g_last_error = "tknnCreate: out is NULL";
    return TKNN_E_ARG;
  }
  *out = nullptr;
  return guarded([&] {
    int n = 0;
    OWLMI_HIP(hipGetDeviceCount(&n));
    if (n <= 0) throw owlmi::HipError{"no HIP device visible: the TrueKNN engine has no CPU fallback"};
    *out = new tknnEngine_t();
  });
}

void tknnDestroy(tknnEngine e) {
  if (!e) return;
  try {
    DeviceScope scope(e->impl.device());
    delete e;
  } catch (...) {
    delete e;
  }
}

int tknnBuildIds(tknnEngine e, const float *d_xyz, const int32_t *d_ids, int64_t n, tknnBuildInfo *info,
                 void *stream) {
  if (!e || !d_xyz || n <= 0 || n >= 0x7fffffffLL) {
    g_last_error = "tknnBuild: need an engine, a device pointer and 0 < n < 2^31-1";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] { e->impl.build(d_xyz, d_ids, n, info, (hipStream_t)stream); });
}

int tknnBuild(tknnEngine e, const float *d_xyz, int64_t n, tknnBuildInfo *info, void *stream) {
  return tknnBuildIds(e, d_xyz, nullptr, n, info, stream);
}

int tknnSetHalo(tknnEngine e, const float *d_xyz, const int32_t *d_ids, int64_t m, void *stream) {
  if (!e || m < 0 || m >= 0x7fffffffLL || (m > 0 && (!d_xyz || !d_ids))) {
    g_last_error = "tknnSetHalo: need an engine and, for m > 0, device pointers to points and ids";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnSetHalo: call tknnBuild first"};
    e->impl.set_halo(d_xyz, d_ids, m, (hipStream_t)stream);
  });
}

int tknnHaloSelect(tknnEngine e, const float *d_boxes, const int32_t *d_box_peer, int32_t nboxes, int32_t npeers,
                   int64_t *d_counts, const int64_t *d_offsets, float *d_rows, void *stream) {
  if (!e || nboxes < 0 || (nboxes > 0 && (!d_boxes || !d_box_peer)) || (!d_rows && !d_counts) || (d_rows && !d_offsets)) {
    g_last_error = "tknnHaloSelect: need boxes with their peers, and counts (count pass) or offsets + rows (write pass)";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnHaloSelect: call tknnBuild first"};
    e->impl.halo_select(d_boxes, d_box_peer, nboxes, npeers, d_counts, d_offsets, d_rows, (hipStream_t)stream);
  });
}

int tknnHaloSelectFixed(tknnEngine e, const float *d_boxes, const int32_t *d_box_peer, int32_t nboxes, int32_t npeers,
                        const int64_t *d_caps, const int64_t *d_offsets, float *d_rows, int64_t *d_counts, void *stream) {
  if (!e || nboxes < 0 || (nboxes > 0 && (!d_boxes || !d_box_peer)) || !d_caps || !d_offsets || !d_rows || !d_counts) {
    g_last_error = "tknnHaloSelectFixed: need boxes with their peers, capacities, offsets, the row buffer and a place for the counts";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnHaloSelectFixed: call tknnBuild first"};
    e->impl.halo_select(d_boxes, d_box_peer, nboxes, npeers, d_counts, d_offsets, d_rows, (hipStream_t)stream, d_caps);
  });
}

int tknnSolve(tknnEngine e, int k, float start_radius, int kernel, int max_rounds, int32_t *d_idx,
              float *d_dist, int64_t *d_intersections, tknnNeigh *d_fb, tknnSolveInfo *info, void *stream) {
  tknnSolveOptions o;
  std::memset(&o, 0, sizeof o);
  o.k = k;
  o.start_radius = start_radius;
  o.kernel = kernel;
  o.max_rounds = max_rounds;
  o.d_idx = d_idx;
  o.d_dist = d_dist;
  o.d_intersections = d_intersections;
  o.d_fb = d_fb;
  return tknnSolveEx(e, &o, info, stream);
}

int tknnSolveEx(tknnEngine e, const tknnSolveOptions *options, tknnSolveInfo *info, void *stream) {
  if (!e || !options) {
    g_last_error = "tknnSolve: engine or options is NULL";
    return TKNN_E_ARG;
  }
  const int k = options->k, kernel = options->kernel, max_rounds = options->max_rounds;
  const float start_radius = options->start_radius;
  int32_t *d_idx = options->d_idx;
  float *d_dist = options->d_dist;
  int64_t *d_intersections = options->d_intersections;
  tknnNeigh *d_fb = options->d_fb;
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnSolve: call tknnBuild first"};
    if (k <= 0) throw owlmi::ArgError{TKNN_E_ARG, "tknnSolve: k must be positive"};
    if (k > TKNN_MAX_K) throw owlmi::ArgError{TKNN_E_UNSUPPORTED, "tknnSolve: k exceeds TKNN_MAX_K"};
    if ((int64_t)k >= e->impl.size() && !options->allow_unfinished)
      throw owlmi::ArgError{TKNN_E_ARG, "tknnSolve: need n > k (the reference never terminates otherwise)"};
    if (!(start_radius > 0.f) || !std::isfinite(start_radius))
      throw owlmi::ArgError{TKNN_E_ARG, "tknnSolve: start_radius must be finite and > 0"};
    if (kernel != TKNN_KERNEL_AUTO && kernel != TKNN_KERNEL_LANE && kernel != TKNN_KERNEL_WAVE &&
        kernel != TKNN_KERNEL_TEAM)
      throw owlmi::ArgError{TKNN_E_ARG, "tknnSolve: unknown kernel selector"};
    owlmi::SolveArgs sa;
    sa.k = k;
    sa.start_radius = start_radius;
    sa.max_rounds = max_rounds > 0 ? std::min(max_rounds, 127) : 64;  // (127: more doublings than fp32 has binades for a radius; the tie flags keep the level in seven bits)
    sa.d_idx = d_idx;
    sa.d_dist = d_dist;
    sa.d_isect = d_intersections;
    sa.d_fb = d_fb;
    sa.d_levels = options->d_levels;
    sa.allow_unfinished = options->allow_unfinished != 0;
    sa.phase = options->phase;
    sa.d_start_radii = options->d_start_radii;
    if (sa.phase < 0 || sa.phase > 3) throw owlmi::ArgError{TKNN_E_ARG, "tknnSolveEx: phase must be 0 (all), 1 (interior), 2 (boundary) or 3 (unfinished)"};
    if (info) std::memset(info, 0, sizeof(*info));
    e->impl.solve(sa, kernel, info, (hipStream_t)stream);
  });
}

int tknnRepairExact(tknnEngine e, int k, float start_radius, const int32_t *d_levels, int32_t *d_idx,
                    float *d_dist, int64_t *repaired, void *stream) {
  if (!e || !d_levels || !d_idx || !d_dist) {
    g_last_error = "tknnRepairExact: engine, levels, idx and dist are required";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnRepairExact: call tknnBuild first"};
    if (k <= 0 || k > TKNN_MAX_K_REGISTERS) throw owlmi::ArgError{TKNN_E_ARG, "tknnRepairExact: k out of range (1 .. 64: the repair pass keeps its lists in registers)"};
    if (!(start_radius > 0.f) || !std::isfinite(start_radius))
      throw owlmi::ArgError{TKNN_E_ARG, "tknnRepairExact: start_radius must be the one the rows were solved with"};
    const int64_t n = e->impl.repair_exact(k, start_radius, d_levels, d_idx, d_dist, (hipStream_t)stream);
    if (repaired) *repaired = n;
  });
}

int tknnDbscan(tknnEngine e, float eps, int min_pts, int32_t *d_labels, uint8_t *d_core, int32_t *d_counts,
               tknnDbscanInfo *info, void *stream) {
  if (!e || !d_labels) {
    g_last_error = "tknnDbscan: engine or labels pointer is NULL";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnDbscan: call tknnBuild first"};
    if (!(eps > 0.f) || !std::isfinite(eps)) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscan: eps must be finite and > 0"};
    if (min_pts < 1) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscan: min_pts must be >= 1"};
    e->impl.dbscan(eps, min_pts, d_labels, d_core, d_counts, info, (hipStream_t)stream);
  });
}

int tknnDbscanAssign(tknnEngine e, float eps, const int32_t *d_core_label, int32_t *d_labels, tknnDbscanInfo *info,
                     void *stream) {
  if (!e || !d_labels || !d_core_label) {
    g_last_error = "tknnDbscanAssign: engine, core labels or labels pointer is NULL";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnDbscanAssign: call tknnBuild first"};
    if (!(eps > 0.f) || !std::isfinite(eps)) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscanAssign: eps must be finite and > 0"};
    e->impl.dbscan(eps, 1, d_labels, nullptr, nullptr, info, (hipStream_t)stream, d_core_label);
  });
}

int tknnDbscanAuto(tknnEngine e, float eps0, int min_pts, double max_noise, int max_rounds, int32_t *d_labels, uint8_t *d_core,
                   tknnDbscanAutoInfo *info, void *stream) {
  if (!e || !d_labels) {
    g_last_error = "tknnDbscanAuto: engine or labels pointer is NULL";
    return TKNN_E_ARG;
  }
  const int rc = guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnDbscanAuto: call tknnBuild first"};
    if (!(eps0 > 0.f) || !std::isfinite(eps0)) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscanAuto: eps0 must be finite and > 0"};
    if (min_pts < 1) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscanAuto: min_pts must be >= 1"};
    if (!(max_noise >= 0.0) || !(max_noise <= 1.0)) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscanAuto: max_noise is a share of the points, 0 .. 1"};
    if (max_rounds < 1) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscanAuto: max_rounds must be >= 1"};
    e->impl.dbscan_auto(eps0, min_pts, max_noise, max_rounds, d_labels, d_core, info, (hipStream_t)stream);
  });
  if (rc == TKNN_E_ROUNDS) g_last_error = "tknnDbscanAuto: max_rounds doublings of eps did not bring the noise under the bound";
  return rc;
}

int tknnDbscanNoise(tknnEngine e, float eps, int min_pts, uint8_t *d_noise, int64_t *noise_count, void *stream) {
  if (!e || !d_noise) {
    g_last_error = "tknnDbscanNoise: engine or flag pointer is NULL";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnDbscanNoise: call tknnBuild first"};
    if (!(eps > 0.f) || !std::isfinite(eps)) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscanNoise: eps must be finite and > 0"};
    if (min_pts < 1) throw owlmi::ArgError{TKNN_E_ARG, "tknnDbscanNoise: min_pts must be >= 1"};
    const int64_t count = e->impl.dbscan_noise(eps, min_pts, d_noise, (hipStream_t)stream);
    if (noise_count) *noise_count = count;
  });
}

int tknnSegmentMin(tknnEngine e, const int32_t *d_segment, const int64_t *d_value, int64_t n, int64_t *d_out, void *stream) {
  if (!e || n < 0 || (n > 0 && (!d_segment || !d_value || !d_out))) {
    g_last_error = "tknnSegmentMin: engine, segments, values and the output are required";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] { owlmi::db_segment_min(d_segment, d_value, n, d_out, (hipStream_t)stream); });
}

int tknnExportTree(tknnEngine e, void *nodes, int32_t *rope_node, int32_t *rope_leaf, int32_t *prim_id,
                   void *stream) {
  if (!e) {
    g_last_error = "tknnExportTree: engine is NULL";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnExportTree: call tknnBuild first"};
    e->impl.tree().download((LbvhNode *)nodes, rope_node, rope_leaf, prim_id, (hipStream_t)stream);
  });
}

int tknnExportTreeTables(tknnEngine e, int32_t *split_owner, int32_t *block_paths, void *stream) {
  if (!e) {
    g_last_error = "tknnExportTreeTables: engine is NULL";
    return TKNN_E_ARG;
  }
  return guarded_on(e, [&] {
    if (!e->impl.built()) throw owlmi::ArgError{TKNN_E_STATE, "tknnExportTreeTables: call tknnBuild first"};
    e->impl.tree().download_tables(split_owner, block_paths, (hipStream_t)stream);
  });
}

int tknnDebugThresholds(const float *d_q, const float *d_r, int64_t n, float *d_lo, float *d_hi, void *stream) {
  if (!d_q || !d_r || !d_lo || !d_hi || n < 0) {
    g_last_error = "tknnDebugThresholds: null pointer";
    return TKNN_E_ARG;
  }
  return guarded([&] {
    owlmi::debug_thresholds(d_q, d_r, n, d_lo, d_hi, (hipStream_t)stream);
    OWLMI_HIP(hipStreamSynchronize((hipStream_t)stream));
  });
}

}  // extern "C"
