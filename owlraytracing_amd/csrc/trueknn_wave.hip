// trueknn_wave.hip -- the persistent wave-packet TrueKNN kernel (TKNN_KERNEL_WAVE).
//
// One wave owns a packet of 64 Morton-consecutive queries (one per lane) and resolves ALL radius
// levels for it inside one launch; waves are persistent and pull packets from an atomic counter.
// Per level:
//   * every lane turns "is candidate c in my box of radius r" (deviceCode.cu:38-56 + the RT-core
//     ray/AABB test) into six thresholds cmin <= c <= cmax that are EXACTLY equivalent to
//     fl(c - r) <= q <= fl(c + r) (monotonicity of fp32 rounding), so the per-candidate test is six
//     compares and no arithmetic;
//   * the wave walks the LBVH cooperatively: an LDS stack of node references, up to 64 nodes popped
//     and box-tested per step (one per lane) against the union of the packet's thresholds,
//     survivors' children pushed with ballot + prefix-count compaction;
//   * subtrees of <= leaf_max points are "leaf ranges": contiguous runs of the Morton-sorted point
//     array.  A range is refined against the 64 individual query boxes (ballot) and then streamed
//     through SCALAR loads (the candidate is wave-uniform), each lane testing it against its own
//     thresholds;
//   * survivors that could enter the lane's k-list are queued in a small per-lane LDS queue and
//     merged into the register-resident sorted list in bursts, so the long insertion sequence runs
//     with most lanes busy instead of once per candidate.
// Lanes whose box held >= k other points are finished (deviceCode.cu:118: numNeighbors reached
// 0) and write their row; the rest double the radius (hostCode.cpp:321) and go again.  Because a
// finished row is the k smallest (dist, index) among the candidates of its final box, recomputing
// the list from scratch at each level gives the same row as the reference's incremental insertion.
#include "trueknn_engine.h"

#include <algorithm>
#include <cstring>

namespace owlmi {

namespace {

constexpr int kWaveBlock = 256;          // 4 independent waves per workgroup
constexpr int kStackCap = 512;           // node references per wave (LDS)
constexpr int kQueueDepth = 8;           // pending k-list candidates per lane (LDS)
constexpr int kWaveLds = kStackCap * 4 + kQueueDepth * 64 * 8;  // 6 KiB per wave

#define OWLMI_AS4 __attribute__((address_space(4)))

struct WaveArgs {
  LbvhView bvh;
  float start_radius;
  int k;
  int max_rounds;
  int leaf_max;
  int32_t ngroups;
  int32_t *out_idx;
  float *out_dist;
  int64_t *out_isect;
  tknnNeigh *out_fb;
  // [0] packet counter  [1] max level+1  [2] node tests  [3] point tests  [4] sum intersections
  // [5] error flags (1: max_rounds reached, 2: stack overflow)
  unsigned long long *counters;
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float next_up(float x) {
  if (!(x < INFINITY)) return x;  // +inf, NaN
  if (x == 0.f) return __uint_as_float(1u);
  uint32_t b = __float_as_uint(x);
  return __uint_as_float(x > 0.f ? b + 1 : b - 1);
}
__device__ __forceinline__ float next_down(float x) {
  if (!(x > -INFINITY)) return x;
  if (x == 0.f) return __uint_as_float(0x80000001u);
  uint32_t b = __float_as_uint(x);
  return __uint_as_float(x > 0.f ? b - 1 : b + 1);
}

// largest c with fl(c - r) <= q   (so that  fl(c - r) <= q  <=>  c <= thr_hi(q, r))
__device__ __forceinline__ float thr_hi(float q, float r) {
#pragma clang fp contract(off)
  float c = q + r;
  if (!(c == c)) return -INFINITY;
  for (;;) {
    float u = next_up(c);
    if (u > c && u - r <= q)
      c = u;
    else
      break;
  }
  while (!(c - r <= q) && c > -INFINITY) c = next_down(c);
  return c;
}
// smallest c with q <= fl(c + r)
__device__ __forceinline__ float thr_lo(float q, float r) {
#pragma clang fp contract(off)
  float c = q - r;
  if (!(c == c)) return INFINITY;
  for (;;) {
    float d = next_down(c);
    if (d < c && q <= d + r)
      c = d;
    else
      break;
  }
  while (!(q <= c + r) && c < INFINITY) c = next_up(c);
  return c;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ int lane_rank(unsigned long long mask) {  // set bits below my lane
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ float bcast_f(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}

template <int K>
struct LaneState {
  KList<K> list;
  float tau2;       // conservative squared-distance gate for the queue
  int qpos;         // entries pending in my LDS queue
  uint32_t cnt;     // candidates in my box this level (self included)   deviceCode.cu:74
  uint32_t others;  // ... excluding myself                              deviceCode.cu:103
};

// squared-distance gate: every d2 with sqrt_rn(d2) <= w must pass ("queue gate")
__device__ __forceinline__ float gate_from_worst(float w) {
#pragma clang fp contract(off)
  float w2 = w * w;
  return w2 * 1.00000048f;  // 1 + 2^-21: at least 4 ulps above fl(w*w)
}

template <int K>
__device__ __forceinline__ void flush_queue(LaneState<K> &st, const uint64_t *queue, int lane) {
  for (int s = 0; s < kQueueDepth; s++) {
    if (__ballot(s < st.qpos) == 0ull) break;
    if (s < st.qpos) {
      uint64_t e = queue[s * 64 + lane];
      float d = knn_sqrt(__uint_as_float((uint32_t)(e >> 32)));
      st.list.insert(knn_key(d, (int32_t)(uint32_t)e));
    }
  }
  st.qpos = 0;
  st.tau2 = gate_from_worst(knn_key_dist(st.list.worst()));
}

template <int K>
__device__ __forceinline__ void emit_row(const WaveArgs &a, int32_t qid, const KList<K> &list, int64_t isect) {
  const int k = a.k;
  const int64_t base = (int64_t)qid * k;
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (j < k) {
      int32_t prim = knn_key_prim(list.key[j]);
      float d = knn_key_dist(list.key[j]);
      if (a.out_idx) a.out_idx[base + j] = prim;
      if (a.out_dist) a.out_dist[base + j] = d;
      if (a.out_fb) {
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
  if (a.out_isect) a.out_isect[qid] = isect;
}

template <int K>
__global__ void __launch_bounds__(kWaveBlock) wave_packet_kernel(WaveArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  int32_t *stack = (int32_t *)(smem + wid * kWaveLds);
  uint64_t *queue = (uint64_t *)(smem + wid * kWaveLds + kStackCap * 4);
  const LbvhView &bvh = a.bvh;
  const OWLMI_AS4 LbvhPoint *cpts = (const OWLMI_AS4 LbvhPoint *)bvh.points;

  unsigned long long my_isect_sum = 0, wave_node_tests = 0, wave_point_tests = 0;
  int wave_levels = 0, wave_err = 0;

  for (;;) {
    int g = 0;
    if (lane == 0) g = (int)atomicAdd(&a.counters[0], 1ull);
    g = __builtin_amdgcn_readfirstlane(g);
    if (g >= a.ngroups) break;

    const int32_t slot = g * 64 + lane;
    bool active = slot < bvh.n;
    LbvhPoint q = {0.f, 0.f, 0.f, -1};
    if (active) q = bvh.points[slot];
    float r = a.start_radius;
    int level = 0;
    int64_t isect = 0;

    for (;;) {  // radius levels (rounds of hostCode.cpp:285-340)
      // exact per-lane candidate thresholds, and their union over the packet
      float lo_x = INFINITY, lo_y = INFINITY, lo_z = INFINITY;
      float hi_x = -INFINITY, hi_y = -INFINITY, hi_z = -INFINITY;
      if (active) {
        lo_x = thr_lo(q.x, r);
        lo_y = thr_lo(q.y, r);
        lo_z = thr_lo(q.z, r);
        hi_x = thr_hi(q.x, r);
        hi_y = thr_hi(q.y, r);
        hi_z = thr_hi(q.z, r);
      }
      const float g_lo_x = wave_min(lo_x), g_lo_y = wave_min(lo_y), g_lo_z = wave_min(lo_z);
      const float g_hi_x = wave_max(hi_x), g_hi_y = wave_max(hi_y), g_hi_z = wave_max(hi_z);

      LaneState<K> st;
      st.list.clear();
      st.tau2 = INFINITY;
      st.qpos = 0;
      st.cnt = 0;
      st.others = 0;

      int sp = 1;
      if (lane == 0) stack[0] = bvh.root;
      wave_lds_sync();

      while (sp > 0) {
        int w = min(64, min(sp, kStackCap - sp));
        if (w <= 0) {  // cannot make room for children: report, the host re-solves with the lane kernel
          wave_err |= 2;
          break;
        }
        int32_t ref = LBVH_END;
        if (lane < w) ref = stack[sp - 1 - lane];
        sp -= w;
        wave_node_tests += (unsigned)w;

        // classify my node: box over centres + covered range
        float n_lo_x = 0, n_lo_y = 0, n_lo_z = 0, n_hi_x = 0, n_hi_y = 0, n_hi_z = 0;
        int32_t first = 0, count = 0, left = LBVH_END, right = LBVH_END;
        bool overlap = false;
        if (lane < w) {
          if (ref >= 0) {
            const LbvhNode nd = bvh.nodes[ref];
            n_lo_x = nd.lo[0];
            n_lo_y = nd.lo[1];
            n_lo_z = nd.lo[2];
            n_hi_x = nd.hi[0];
            n_hi_y = nd.hi[1];
            n_hi_z = nd.hi[2];
            first = lbvh_first(ref, nd.other);
            count = lbvh_last(ref, nd.other) - first + 1;
            left = lbvh_left_ref(ref, nd);
            right = lbvh_right_ref(ref, nd);
          } else {
            const LbvhPoint p = bvh.points[~ref];
            n_lo_x = n_hi_x = p.x;
            n_lo_y = n_hi_y = p.y;
            n_lo_z = n_hi_z = p.z;
            first = ~ref;
            count = 1;
          }
          // some centre in [n_lo, n_hi] may satisfy g_lo <= c <= g_hi on every axis
          overlap = (n_lo_x <= g_hi_x) & (n_hi_x >= g_lo_x) & (n_lo_y <= g_hi_y) & (n_hi_y >= g_lo_y) &
                    (n_lo_z <= g_hi_z) & (n_hi_z >= g_lo_z);
        }
        const bool is_range = overlap && count <= a.leaf_max;
        const bool expand = overlap && !is_range;

        // push children of expanding nodes: ballot + prefix count compaction
        const unsigned long long emask = __ballot(expand);
        if (expand) {
          const int at = sp + 2 * lane_rank(emask);
          stack[at] = right;
          stack[at + 1] = left;
        }
        sp += 2 * __popcll(emask);
        wave_lds_sync();

        // leaf ranges: stream their points past all 64 query boxes
        unsigned long long rmask = __ballot(is_range);
        while (rmask) {
          const int src = __ffsll((long long)rmask) - 1;
          rmask &= rmask - 1;
          const float b_lo_x = bcast_f(n_lo_x, src), b_lo_y = bcast_f(n_lo_y, src), b_lo_z = bcast_f(n_lo_z, src);
          const float b_hi_x = bcast_f(n_hi_x, src), b_hi_y = bcast_f(n_hi_y, src), b_hi_z = bcast_f(n_hi_z, src);
          const bool mine = active & (b_lo_x <= hi_x) & (b_hi_x >= lo_x) & (b_lo_y <= hi_y) & (b_hi_y >= lo_y) &
                            (b_lo_z <= hi_z) & (b_hi_z >= lo_z);
          const unsigned long long takers = __ballot(mine);
          if (takers == 0ull) continue;
          const int r_first = __builtin_amdgcn_readlane(first, src);
          const int r_count = __builtin_amdgcn_readlane(count, src);
          wave_point_tests += (unsigned long long)r_count * (unsigned)__popcll(takers);
          for (int j = 0; j < r_count; j++) {
            const LbvhPoint p = cpts[r_first + j];  // wave-uniform -> scalar load
            const bool in = active & (lo_x <= p.x) & (p.x <= hi_x) & (lo_y <= p.y) & (p.y <= hi_y) &
                            (lo_z <= p.z) & (p.z <= hi_z);
            const bool other = in & (p.id != q.id);
            st.cnt += in ? 1u : 0u;
            st.others += other ? 1u : 0u;
            const float d2 = knn_dist2(p.x, p.y, p.z, q.x, q.y, q.z);
            if (other & (d2 <= st.tau2)) {
              queue[st.qpos * 64 + lane] = ((uint64_t)__float_as_uint(d2) << 32) | (uint32_t)p.id;
              st.qpos++;
            }
            if (__ballot(st.qpos == kQueueDepth) != 0ull) flush_queue<K>(st, queue, lane);
          }
        }
      }
      flush_queue<K>(st, queue, lane);

      bool finished = false;
      if (active) {
        isect += st.cnt;
        finished = st.others >= (uint32_t)a.k;
        if (finished) {
          emit_row<K>(a, q.id, st.list, isect);
          my_isect_sum += (unsigned long long)isect;
        }
      }
      active = active && !finished;
      level++;
      if (__ballot(active) == 0ull || wave_err) break;
      if (level >= a.max_rounds) {
        wave_err |= 1;
        break;
      }
      r = r * 2.0f;  // hostCode.cpp:321
    }
    wave_levels = max(wave_levels, level);
  }

  // once per wave lifetime
  const unsigned long long isum = wave_sum(my_isect_sum);
  if (lane == 0) {
    atomicMax(&a.counters[1], (unsigned long long)wave_levels);
    atomicAdd(&a.counters[2], wave_node_tests);
    atomicAdd(&a.counters[3], wave_point_tests);
    atomicAdd(&a.counters[4], isum);
    if (wave_err) atomicOr(&a.counters[5], (unsigned long long)wave_err);
  }
}

template <int K>
void launch_wave(const WaveArgs &a, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(wave_packet_kernel<K>, dim3(blocks), dim3(kWaveBlock), kWaveBlock / 64 * kWaveLds, s, a);
}

template <int K>
int max_blocks_per_cu() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wave_packet_kernel<K>, kWaveBlock,
                                                   kWaveBlock / 64 * kWaveLds) != hipSuccess)
    nb = 2;
  return std::max(1, nb);
}

}  // namespace

bool Engine::wave_kernel_available() { return true; }

void Engine::solve_wave(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s) {
  const int64_t n = bvh_.size();
  const int cap = list_capacity_for(sa.k);
  WaveArgs a;
  a.bvh = bvh_.view();
  a.start_radius = sa.start_radius;
  a.k = sa.k;
  a.max_rounds = sa.max_rounds;
  a.leaf_max = wave_leaf_max_;
  a.ngroups = (int32_t)((n + 63) / 64);
  a.out_idx = sa.d_idx;
  a.out_dist = sa.d_dist;
  a.out_isect = sa.d_isect;
  a.out_fb = sa.d_fb;
  a.counters = counters_;

  hipDeviceProp_t prop;
  OWLMI_HIP(hipGetDeviceProperties(&prop, device_));
  int per_cu = 1;
  switch (cap) {
    case 1: per_cu = max_blocks_per_cu<1>(); break;
    case 2: per_cu = max_blocks_per_cu<2>(); break;
    case 4: per_cu = max_blocks_per_cu<4>(); break;
    case 5: per_cu = max_blocks_per_cu<5>(); break;
    case 8: per_cu = max_blocks_per_cu<8>(); break;
    case 10: per_cu = max_blocks_per_cu<10>(); break;
    case 16: per_cu = max_blocks_per_cu<16>(); break;
    case 24: per_cu = max_blocks_per_cu<24>(); break;
    case 32: per_cu = max_blocks_per_cu<32>(); break;
    default: per_cu = max_blocks_per_cu<64>(); break;
  }
  const int64_t want = (a.ngroups + kWaveBlock / 64 - 1) / (kWaveBlock / 64);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)prop.multiProcessorCount * per_cu));

  OWLMI_HIP(hipMemsetAsync(counters_, 0, 16 * sizeof(unsigned long long), s));
  OWLMI_HIP(hipEventRecord(ev_a_, s));
  switch (cap) {
    case 1: launch_wave<1>(a, blocks, s); break;
    case 2: launch_wave<2>(a, blocks, s); break;
    case 4: launch_wave<4>(a, blocks, s); break;
    case 5: launch_wave<5>(a, blocks, s); break;
    case 8: launch_wave<8>(a, blocks, s); break;
    case 10: launch_wave<10>(a, blocks, s); break;
    case 16: launch_wave<16>(a, blocks, s); break;
    case 24: launch_wave<24>(a, blocks, s); break;
    case 32: launch_wave<32>(a, blocks, s); break;
    default: launch_wave<64>(a, blocks, s); break;
  }
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipEventRecord(ev_b_, s));
  OWLMI_HIP(hipMemcpyAsync(h_counters_, counters_, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipStreamSynchronize(s));
  float ms = 0;
  OWLMI_HIP(hipEventElapsedTime(&ms, ev_a_, ev_b_));
  if (h_counters_[5] & 2ull) {
    // LDS node stack exhausted on some packet (pathologically deep tree): redo with the lane kernel
    solve_lane(sa, info, s);
    return;
  }
  if (h_counters_[5] & 1ull) throw RoundsExceeded{};
  if (info) {
    const int rounds = (int)h_counters_[1];
    info->rounds = rounds;
    float radius = sa.start_radius;
    for (int t = 1; t < rounds; t++) radius *= 2;
    info->final_radius = radius;
    info->node_tests = (int64_t)h_counters_[2];
    info->point_tests = (int64_t)h_counters_[3];
    info->total_intersections = (int64_t)h_counters_[4];
    info->solve_ms = ms;
    info->dominant_kernel_ms = ms;
    info->dominant_kernel_launches = 1;
    info->kernel_used = TKNN_KERNEL_WAVE;
    info->list_capacity = cap;
  }
}

}  // namespace owlmi
