// halo_select.hip -- the send side of the halo exchange: which of this tile's
// points lie inside any of the (widened) Morton-cell boxes of each peer tile, as contiguous 16-byte
// wire rows per peer.  The reference has no multi-GPU path for this workload (owl/RayGen.cpp:150-200
// replicates everything); this replaces a per-peer, per-cell host loop of tensor operations that
// cost more than the solve itself.
//
// Two kernels over the tile's Morton-sorted points (the engine's own LBVH arrays):
//   1. one thread per 16-point leaf block: the block's box against all peer boxes -> a 64-bit mask
//      of the peers the block may have points for (almost all blocks: 0);
//   2. one thread per point of a block with a non-zero mask: the point against the boxes of those
//      peers -> its own peer mask; per workgroup the per-peer counts are gathered in LDS, one global
//      atomic per peer reserves the workgroup's range, and the rows are scattered.
// tknnHaloSelect runs them twice: counting (the caller sizes and offsets the send buffer), then
// writing.  The order of rows inside a peer's segment is unspecified (results of the solve do not
// depend on it: keys are (distance, global index)).
#include "trueknn_engine.h"

#include <cstring>

namespace owlmi {

namespace {

constexpr int kSelBlock = 256;
constexpr int kMaxPeers = 64;

struct SelectArgs {
  const LbvhPoint *points;  // n sorted points (+ sentinels up to whole blocks)
  const LbvhBox *block_box; // one box per 16-point block
  int32_t n;
  int32_t nblocks;
  const float *boxes;       // nboxes x {lo xyz, hi xyz}
  const int32_t *box_peer;  // nboxes
  int32_t nboxes;
  int32_t npeers;
  unsigned long long *block_mask;  // nblocks
  unsigned long long *cursor;      // npeers: rows handed out so far (write pass)
  int64_t *counts;                 // npeers (count pass)
  const int64_t *offsets;          // npeers: first row of each peer's segment (write pass)
  float4 *rows;                    // write pass
  uint8_t *boundary;               // count pass: per sorted slot, 1 if the point lies inside some peer's box
  const int64_t *caps;             // one-pass form (tknnHaloSelectFixed): rows a peer's segment holds; what does not fit is counted, not written
};

__global__ void __launch_bounds__(kSelBlock) block_mask_kernel(SelectArgs a) {
  const int32_t b = blockIdx.x * kSelBlock + threadIdx.x;
  if (b >= a.nblocks) return;
  const LbvhBox bx = a.block_box[b];
  unsigned long long m = 0;
  for (int j = 0; j < a.nboxes; j++) {  // uniform addresses: scalar loads
    const float *q = a.boxes + 6 * j;
    const bool ov = (bx.lo[0] <= q[3]) & (bx.hi[0] >= q[0]) & (bx.lo[1] <= q[4]) & (bx.hi[1] >= q[1]) &
                    (bx.lo[2] <= q[5]) & (bx.hi[2] >= q[2]);
    if (ov) m |= 1ull << a.box_peer[j];
  }
  a.block_mask[b] = m;
}

// FIXED: the one-pass form -- segments of a capacity both ends of a pair have agreed on (the message sizes of the pair's last
// exchange), so no count pass and no host round trip come before the rows are written; the cursors end as the exact counts
// (a peer whose rows did not fit is found out by them), and the boundary marks of the count pass are set here.
template <bool WRITE, bool FIXED = false>
__global__ void __launch_bounds__(kSelBlock) point_select_kernel(SelectArgs a) {
  __shared__ unsigned int cnt[kMaxPeers];
  __shared__ unsigned long long base[kMaxPeers];
  if (threadIdx.x < kMaxPeers) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int32_t t = blockIdx.x * kSelBlock + threadIdx.x;
  unsigned long long mine = 0;
  LbvhPoint p = {0.f, 0.f, 0.f, -1};
  if (t < a.n) {
    const unsigned long long bm = a.block_mask[t / LBVH_BLOCK];
    if (bm) {
      p = a.points[t];
      for (int j = 0; j < a.nboxes; j++) {
        const int peer = a.box_peer[j];
        if (!((bm >> peer) & 1ull)) continue;
        const float *q = a.boxes + 6 * j;
        // closed box; a NaN point fails every comparison
        const bool in = (p.x >= q[0]) & (p.x <= q[3]) & (p.y >= q[1]) & (p.y <= q[4]) & (p.z >= q[2]) & (p.z <= q[5]);
        if (in) mine |= 1ull << peer;
      }
    }
  }
  if ((!WRITE || FIXED) && a.boundary && t < a.n) a.boundary[t] = mine != 0ull;
  for (unsigned long long m = mine; m; m &= m - 1) atomicAdd(&cnt[__builtin_ctzll(m)], 1u);
  __syncthreads();
  if (threadIdx.x < a.npeers && cnt[threadIdx.x]) {
    if (WRITE) {
      base[threadIdx.x] = atomicAdd(&a.cursor[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
    } else {
      atomicAdd((unsigned long long *)&a.counts[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
    }
  }
  if (!WRITE) return;
  __syncthreads();
  if (threadIdx.x < kMaxPeers) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (unsigned long long m = mine; m; m &= m - 1) {
    const int peer = __builtin_ctzll(m);
    const unsigned int r = atomicAdd(&cnt[peer], 1u);
    const int64_t at = (int64_t)base[peer] + r;
    if (!FIXED || at < a.caps[peer]) a.rows[a.offsets[peer] + at] = make_float4(p.x, p.y, p.z, __int_as_float(p.id));
  }
}

}  // namespace

void Engine::halo_select(const float *d_boxes, const int32_t *d_box_peer, int32_t nboxes, int32_t npeers,
                         int64_t *d_counts, const int64_t *d_offsets, float *d_rows, hipStream_t s, const int64_t *d_caps) {
  if (npeers < 1 || npeers > kMaxPeers) throw ArgError{TKNN_E_ARG, "tknnHaloSelect: 1 <= npeers <= 64"};
  const int64_t n = bvh_.size();
  const LbvhWideView wv = bvh_.wide_view();
  const LbvhView v = bvh_.view();
  SelectArgs a;
  std::memset(&a, 0, sizeof a);
  a.points = v.points;
  a.block_box = wv.level[0];
  a.n = (int32_t)n;
  a.nblocks = wv.count[0];
  a.boxes = d_boxes;
  a.box_peer = d_box_peer;
  a.nboxes = nboxes;
  a.npeers = npeers;
  if ((int64_t)a.nblocks > halo_mask_cap_) {
    if (halo_mask_) (void)hipFree(halo_mask_);
    halo_mask_ = nullptr;
    OWLMI_HIP(hipMalloc((void **)&halo_mask_, (size_t)a.nblocks * sizeof(unsigned long long) + kMaxPeers * sizeof(unsigned long long)));
    halo_mask_cap_ = a.nblocks;
  }
  a.block_mask = halo_mask_;
  a.cursor = halo_mask_ + halo_mask_cap_;
  a.counts = d_counts;
  a.offsets = d_offsets;
  a.rows = reinterpret_cast<float4 *>(d_rows);
  const unsigned point_blocks = (unsigned)((n + kSelBlock - 1) / kSelBlock);
  a.boundary = boundary_;
  a.caps = d_caps;
  if (d_caps) {
    // one pass: rows into segments of the given capacities, exact counts out (tknnHaloSelectFixed)
    boundary_valid_ = true;
    OWLMI_HIP(hipMemsetAsync(a.cursor, 0, kMaxPeers * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(block_mask_kernel, dim3((a.nblocks + kSelBlock - 1) / kSelBlock), dim3(kSelBlock), 0, s, a);
    hipLaunchKernelGGL((point_select_kernel<true, true>), dim3(point_blocks), dim3(kSelBlock), 0, s, a);
    OWLMI_HIP(hipMemcpyAsync(d_counts, a.cursor, (size_t)npeers * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
  } else if (!d_rows) {
    boundary_valid_ = true;  // (the marks are those of THESE boxes: a point inside a peer's widened cell box is a query
                             // that peer's points can reach, and only such a query -- the halo relation is symmetric)
    OWLMI_HIP(hipMemsetAsync(d_counts, 0, (size_t)npeers * sizeof(int64_t), s));
    hipLaunchKernelGGL(block_mask_kernel, dim3((a.nblocks + kSelBlock - 1) / kSelBlock), dim3(kSelBlock), 0, s, a);
    hipLaunchKernelGGL(point_select_kernel<false>, dim3(point_blocks), dim3(kSelBlock), 0, s, a);
  } else {
    OWLMI_HIP(hipMemsetAsync(a.cursor, 0, kMaxPeers * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(block_mask_kernel, dim3((a.nblocks + kSelBlock - 1) / kSelBlock), dim3(kSelBlock), 0, s, a);
    hipLaunchKernelGGL(point_select_kernel<true>, dim3(point_blocks), dim3(kSelBlock), 0, s, a);
  }
  OWLMI_HIP(hipGetLastError());
}

}  // namespace owlmi
