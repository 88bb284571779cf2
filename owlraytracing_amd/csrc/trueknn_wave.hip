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
//     array.  A range is refined against the 64 individual query boxes (ballot), loaded with one
//     coalesced 16 B/lane access (lane l holds its l-th point) and broadcast point by point with
//     v_readlane, each lane testing the candidate against its own thresholds -- the candidate
//     loop touches no memory;
//   * survivors that could enter the lane's k-list are queued in a small per-lane LDS queue and
//     merged into the register-resident sorted list in bursts, so the long insertion sequence runs
//     with most lanes busy instead of once per candidate.
// Lanes whose box held >= k other points are finished (deviceCode.cu:118: numNeighbors reached
// 0) and write their row; the rest double the radius (hostCode.cpp:321) and go again.  Because a
// finished row is the k smallest (dist, index) among the candidates of its final box, recomputing
// the list from scratch at each level gives the same row as the reference's incremental insertion.
#include "knn_thresholds.h"
#include "trueknn_engine.h"

#include <algorithm>
#include <cstring>

namespace owlmi {

namespace {

constexpr int kWaveBlock = 256;          // 4 independent waves per workgroup
constexpr int kStackCap = 1024;          // node references per wave (LDS)
constexpr int kStackReserve = 160;       // slots kept for depth-first descent (tree depth <= 63 + 32 + 1)
constexpr int kQueueDepth = 8;           // pending k-list candidates per lane (LDS)
constexpr int kWaveLds = kStackCap * 4 + kQueueDepth * 64 * 8;  // 8 KiB per wave


struct WaveArgs {
  LbvhView bvh;
  LbvhView halo;      // second point set searched by every query (n == 0: none)
  int32_t *out_level; // n, caller order (may be null)
  const uint8_t *skip_done;  // per sorted slot, may be null: queries another kernel has finished already
  uint8_t *tie;              // per sorted slot: (1 + level) | 0x80 for rows finished with exact-distance ties (knn_flag_tie)
  int32_t *tie_list;
  int allow_unfinished;
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
  // [5] error flags (1: max_rounds reached, 2: stack overflow)  [6] sum over queries of levels run
  unsigned long long *counters;
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
  st.tau2 = knn_gate_from_worst(knn_key_dist(st.list.worst()));
}

template <int K>
__device__ __forceinline__ void emit_row(const WaveArgs &a, int32_t row, int level, const KList<K> &list,
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
  if (a.out_level) a.out_level[row] = level;
}

template <int K>
__global__ void __launch_bounds__(kWaveBlock) wave_packet_kernel(WaveArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  int32_t *stack = (int32_t *)(smem + wid * kWaveLds);
  uint64_t *queue = (uint64_t *)(smem + wid * kWaveLds + kStackCap * 4);
  const LbvhView &bvh = a.bvh;

  unsigned long long my_isect_sum = 0, my_levels = 0, my_unfinished = 0, wave_node_tests = 0, wave_point_tests = 0;
  int wave_levels = 0, wave_err = 0;

  for (;;) {
    int g = 0;
    if (lane == 0) g = (int)atomicAdd(&a.counters[0], 1ull);
    g = __builtin_amdgcn_readfirstlane(g);
    if (g >= a.ngroups) break;

    const int32_t slot = g * 64 + lane;
    bool active = slot < bvh.n;
    if (a.skip_done) {
      active = active && !a.skip_done[slot];
      if (__ballot(active) == 0ull) continue;
    }
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

      // One leaf range is always "pending" in registers (lane l holds its l-th point, loaded with
      // one coalesced 16 B/lane access); its points are broadcast with v_readlane, so the candidate
      // loop touches no memory.  The next range's load is issued before the pending one is
      // processed, and the node loads of the next pop are issued before the last range of the
      // previous pop is processed, so HBM/L2 latency hides behind the compare/insert work.
      LbvhPoint pend = {0.f, 0.f, 0.f, -1};
      int pend_count = 0;
      auto process_pending = [&]() {
        for (int j = 0; j < pend_count; j++) {
          const float px = bcast_f(pend.x, j), py = bcast_f(pend.y, j), pz = bcast_f(pend.z, j);
          const int32_t pid = __builtin_amdgcn_readlane(pend.id, j);
          const bool in = (lo_x <= px) & (px <= hi_x) & (lo_y <= py) & (py <= hi_y) & (lo_z <= pz) & (pz <= hi_z);
          const bool other = in & (pid != q.id);
          st.cnt += in ? 1u : 0u;
          st.others += other ? 1u : 0u;
          const float d2 = knn_dist2(px, py, pz, q.x, q.y, q.z);
          if (other & (d2 <= st.tau2)) {
            queue[st.qpos * 64 + lane] = ((uint64_t)__float_as_uint(d2) << 32) | (uint32_t)pid;
            st.qpos++;
          }
          if (__ballot(st.qpos == kQueueDepth) != 0ull) flush_queue<K>(st, queue, lane);
        }
        pend_count = 0;
      };

      for (int tree = 0; tree < 2 && !wave_err; tree++) {
      const LbvhView &tv = tree == 0 ? a.bvh : a.halo;
      if (tv.n <= 0) continue;
      int sp = 1;
      if (lane == 0) stack[0] = tv.root;
      wave_lds_sync();

      while (sp > 0) {
        // Pop width: up to 64 nodes while the stack has room for all their children; as it fills,
        // narrow down to single-node (depth-first) steps, which need at most one slot per tree
        // level still below -- kStackReserve covers the deepest possible radix tree.
        const int room = kStackCap - sp;
        int w = min(64, min(sp, max(1, room - kStackReserve)));
        if (room < 2) {  // not even one node's children fit: report, the host re-solves with the lane kernel
          wave_err |= 2;
          break;
        }
        int32_t ref = LBVH_END;
        if (lane < w) ref = stack[sp - 1 - lane];
        sp -= w;
        wave_node_tests += (unsigned)w;

        // issue my node's loads, then overlap them with the pending range
        LbvhNode nd = {{0.f, 0.f, 0.f}, 0, {0.f, 0.f, 0.f}, 0};
        LbvhPoint np = {0.f, 0.f, 0.f, -1};
        if (lane < w) {
          if (ref >= 0)
            nd = tv.nodes[ref];
          else
            np = tv.points[~ref];
        }
        if (pend_count) process_pending();

        // classify my node: box over centres + covered range
        float n_lo_x = 0, n_lo_y = 0, n_lo_z = 0, n_hi_x = 0, n_hi_y = 0, n_hi_z = 0;
        int32_t first = 0, count = 0, left = LBVH_END, right = LBVH_END;
        bool overlap = false;
        if (lane < w) {
          if (ref >= 0) {
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
            n_lo_x = n_hi_x = np.x;
            n_lo_y = n_hi_y = np.y;
            n_lo_z = n_hi_z = np.z;
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

        // leaf ranges: refine against the 64 individual query boxes, then stream the survivors
        unsigned long long rmask = __ballot(is_range);
        while (rmask) {
          const int src = __ffsll((long long)rmask) - 1;
          rmask &= rmask - 1;
          const float b_lo_x = bcast_f(n_lo_x, src), b_lo_y = bcast_f(n_lo_y, src), b_lo_z = bcast_f(n_lo_z, src);
          const float b_hi_x = bcast_f(n_hi_x, src), b_hi_y = bcast_f(n_hi_y, src), b_hi_z = bcast_f(n_hi_z, src);
          const bool mine = (b_lo_x <= hi_x) & (b_hi_x >= lo_x) & (b_lo_y <= hi_y) & (b_hi_y >= lo_y) &
                            (b_lo_z <= hi_z) & (b_hi_z >= lo_z);
          const unsigned long long takers = __ballot(mine);
          if (takers == 0ull) continue;
          const int r_first = __builtin_amdgcn_readlane(first, src);
          const int r_count = __builtin_amdgcn_readlane(count, src);
          wave_point_tests += (unsigned long long)r_count * (unsigned)__popcll(takers);
          LbvhPoint nxt = {0.f, 0.f, 0.f, -1};
          if (lane < r_count) nxt = tv.points[r_first + lane];
          if (pend_count) process_pending();
          pend = nxt;
          pend_count = r_count;
        }
      }
      }  // trees
      if (pend_count) process_pending();
      flush_queue<K>(st, queue, lane);

      bool finished = false;
      if (active) {
        my_levels++;
        isect += st.cnt;
        finished = st.others >= (uint32_t)a.k;
        if (finished) {
          emit_row<K>(a, bvh.prim_id[slot], level, st.list, isect);
          if (st.list.has_ties(a.k)) knn_flag_tie(a.tie, a.tie_list, a.counters, slot, level);  // redone by tie_fix_kernel in the reference's tie order
          my_isect_sum += (unsigned long long)isect;
        }
      }
      active = active && !finished;
      level++;
      if (__ballot(active) == 0ull || wave_err) break;
      if (level >= a.max_rounds) {
        if (!a.allow_unfinished) wave_err |= 1;
        break;
      }
      r = r * 2.0f;  // hostCode.cpp:321
    }
    wave_levels = max(wave_levels, level);
    if (active) my_unfinished++;
  }

  // once per wave lifetime
  const unsigned long long isum = wave_sum(my_isect_sum);
  const unsigned long long lsum = wave_sum(my_levels);
  const unsigned long long usum = wave_sum(my_unfinished);
  if (lane == 0) {
    atomicMax(&a.counters[1], (unsigned long long)wave_levels);
    atomicAdd(&a.counters[2], wave_node_tests);
    atomicAdd(&a.counters[3], wave_point_tests);
    atomicAdd(&a.counters[4], isum);
    atomicAdd(&a.counters[6], lsum);
    if (usum) atomicAdd(&a.counters[7], usum);
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

__global__ void thresholds_kernel(const float *__restrict__ q, const float *__restrict__ r, int64_t n,
                                  float *__restrict__ lo, float *__restrict__ hi) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  lo[i] = thr_lo(q[i], r[i]);
  hi[i] = thr_hi(q[i], r[i]);
}

}  // namespace

void debug_thresholds(const float *d_q, const float *d_r, int64_t n, float *d_lo, float *d_hi, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(thresholds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_q, d_r, n, d_lo, d_hi);
  OWLMI_HIP(hipGetLastError());
}

bool Engine::wave_kernel_available() { return true; }

void Engine::solve_wave(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s, bool only_unfinished) {
  const int64_t n = bvh_.size();
  const int cap = list_capacity_for(sa.k);
  WaveArgs a;
  a.bvh = bvh_.view();
  a.halo = halo_view();
  a.out_level = sa.d_levels;
  a.tie = tie_;
  a.tie_list = tie_list_;
  a.skip_done = only_unfinished ? done_ : nullptr;  // the team kernel's stragglers, solved from level 0
  a.allow_unfinished = sa.allow_unfinished ? 1 : 0;
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
  if (sa.d_levels && !only_unfinished) OWLMI_HIP(hipMemsetAsync(sa.d_levels, 0xff, (size_t)n * sizeof(int32_t), s));
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
  if ((h_counters_[5] & 2ull) || wave_force_redo_) {  // (TKNN_WAVE_FORCE_REDO: tests take this path without a pathological tree)
    // LDS node stack exhausted on some packet (pathologically deep tree): redo with the lane kernel
    if (only_unfinished) {
      continue_lane(sa, 0, info, s);  // done[] still names the stragglers; rows are simply rewritten
    } else {
      // the whole solve is redone: forget the rows this launch has flagged, or they are counted twice
      OWLMI_HIP(hipMemsetAsync(tie_, 0, (size_t)n, s));
      OWLMI_HIP(hipMemsetAsync(counters_ + kTieCounter, 0, 3 * sizeof(unsigned long long), s));
      solve_lane(sa, info, s);
    }
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
    info->total_active_rounds = (int64_t)h_counters_[6];
    info->solve_ms = ms;
    info->dominant_kernel_ms = ms;
    info->dominant_kernel_launches = 1;
    info->kernel_used = TKNN_KERNEL_WAVE;
    info->list_capacity = cap;
    info->unfinished = (int64_t)h_counters_[7];
  }
}

}  // namespace owlmi
