// lbvh.hip -- LBVH construction for gfx950.
//
// Pipeline (all on the caller's stream, no host synchronisation, no inter-workgroup hand-offs):
//   scene bounds (2-stage reduction) -> 63-bit Morton codes -> radix sort (hipCUB) -> gather into
//   Morton order -> Karras radix tree (one thread per internal node) -> bounding boxes + ropes.
//
// Boxes are NOT fitted bottom-up with atomic "second arrival" counters: on an 8-XCD part that
// needs an agent-scope release/acquire per tree level per thread (the XCD L2s are not coherent
// inside a launch).  Instead every internal node knows the sorted range it covers, so its box is
// a range min/max query answered from a 64-ary table pyramid (level 0 = elements, level l = boxes
// of 64^l consecutive elements).  Each node reads <= 126 entries per level, every kernel only
// reads what an earlier launch wrote, and the result is deterministic.
#include "lbvh.h"

#include <hipcub/hipcub.hpp>

namespace owlmi {
namespace {

constexpr int kPartialBlocks = 1024;
constexpr int kBlock = 256;

struct Box6 {
  float lo[3], hi[3];
};

__device__ __forceinline__ Box6 empty_box() {
  Box6 b;
  b.lo[0] = b.lo[1] = b.lo[2] = INFINITY;
  b.hi[0] = b.hi[1] = b.hi[2] = -INFINITY;
  return b;
}
__device__ __forceinline__ void grow(Box6 &b, const float *lo, const float *hi) {
#pragma unroll
  for (int a = 0; a < 3; a++) {
    b.lo[a] = fminf(b.lo[a], lo[a]);
    b.hi[a] = fmaxf(b.hi[a], hi[a]);
  }
}
__device__ __forceinline__ Box6 wave_union(Box6 b) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      b.lo[a] = fminf(b.lo[a], __shfl_xor(b.lo[a], off));
      b.hi[a] = fmaxf(b.hi[a], __shfl_xor(b.hi[a], off));
    }
  return b;
}

// ---- scene bounds ---------------------------------------------------------------------------
// elem(i) gives the box of caller primitive i (a point is a degenerate box)
template <bool POINTS>
__global__ void __launch_bounds__(kBlock) scene_partial_kernel(const float *__restrict__ xyz,
                                                              const LbvhBox *__restrict__ boxes,
                                                              int64_t n, float *__restrict__ partials) {
  Box6 b = empty_box();
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (POINTS) {
      float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
      grow(b, p, p);
    } else {
      LbvhBox q = boxes[i];
      grow(b, q.lo, q.hi);
    }
  }
  b = wave_union(b);
  __shared__ Box6 sm[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    Box6 r = sm[0];
    for (int w = 1; w < kBlock / 64; w++) grow(r, sm[w].lo, sm[w].hi);
    for (int a = 0; a < 3; a++) {
      partials[blockIdx.x * 6 + a] = r.lo[a];
      partials[blockIdx.x * 6 + 3 + a] = r.hi[a];
    }
  }
}

__global__ void __launch_bounds__(kBlock) scene_final_kernel(const float *__restrict__ partials, int nblocks,
                                                            float *__restrict__ scene) {
  Box6 b = empty_box();
  for (int i = threadIdx.x; i < nblocks; i += kBlock) grow(b, partials + 6 * i, partials + 6 * i + 3);
  b = wave_union(b);
  __shared__ Box6 sm[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    Box6 r = sm[0];
    for (int w = 1; w < kBlock / 64; w++) grow(r, sm[w].lo, sm[w].hi);
    for (int a = 0; a < 3; a++) {
      scene[a] = r.lo[a];
      scene[3 + a] = r.hi[a];
    }
  }
}

// ---- Morton codes ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spread21(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

template <bool POINTS>
__global__ void __launch_bounds__(kBlock) morton_kernel(const float *__restrict__ xyz,
                                                       const LbvhBox *__restrict__ boxes, int64_t n,
                                                       const float *__restrict__ scene,
                                                       uint64_t *__restrict__ codes,
                                                       uint32_t *__restrict__ order) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float c[3];
  if (POINTS) {
    c[0] = xyz[3 * i];
    c[1] = xyz[3 * i + 1];
    c[2] = xyz[3 * i + 2];
  } else {
    LbvhBox q = boxes[i];
    for (int a = 0; a < 3; a++) c[a] = 0.5f * q.lo[a] + 0.5f * q.hi[a];
  }
  float ext = fmaxf(fmaxf(scene[3] - scene[0], scene[4] - scene[1]), scene[5] - scene[2]);
  // cubic cells: one scale for all axes; degenerate scenes (all points equal) map to cell 0
  float scale = (ext > 0.f && ext < INFINITY) ? 2097151.0f / ext : 0.f;
  uint64_t q[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    float t = (c[a] - scene[a]) * scale;
    t = fminf(fmaxf(t, 0.f), 2097151.0f);  // NaN -> 0 via fmaxf
    q[a] = (uint64_t)t;
  }
  uint64_t code = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  // a primitive with a NaN coordinate sorts after every real one (bit 63): such points are nobody's
  // candidates, and kernels that add up whole subtrees need to know where they are
  if (c[0] != c[0] || c[1] != c[1] || c[2] != c[2]) code = 1ull << 63;
  codes[i] = code;
  order[i] = (uint32_t)i;
}

template <bool POINTS>
__global__ void __launch_bounds__(kBlock) gather_kernel(const float *__restrict__ xyz,
                                                       const LbvhBox *__restrict__ boxes, int64_t n,
                                                       const uint32_t *__restrict__ order,
                                                       LbvhPoint *__restrict__ points,
                                                       LbvhBox *__restrict__ sorted_boxes,
                                                       int32_t *__restrict__ prim_id,
                                                       const int32_t *__restrict__ ids,
                                                       int32_t *__restrict__ nan_count,
                                                       int32_t *__restrict__ row_slot) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) {
    // sentinels fill the last leaf block and one whole block after it (the team kernel's "no
    // block" entry): NaN coordinates fail every box comparison
    if (POINTS && i < (n + LBVH_BLOCK - 1) / LBVH_BLOCK * LBVH_BLOCK + LBVH_BLOCK) {
      LbvhPoint s;
      s.x = s.y = s.z = __uint_as_float(0x7fc00000u);
      s.id = -1;
      points[i] = s;
    }
    return;
  }
  uint32_t src = order[i];
  prim_id[i] = (int32_t)src;
  // the inverse of prim_id (point trees): a result that lives by sorted slot reaches the caller's rows by a GATHER with
  // coalesced stores instead of a scatter of partial sectors (RT-DBSCAN's labels); lives in the sort's input buffer
  if (row_slot) row_slot[src] = (int32_t)i;
  if (POINTS) {
    LbvhPoint p;
    p.x = xyz[3 * (int64_t)src];
    p.y = xyz[3 * (int64_t)src + 1];
    p.z = xyz[3 * (int64_t)src + 2];
    // a point with any NaN coordinate is nobody's candidate (every closed-box comparison with NaN is
    // false) and finds none; store it as all-NaN so that kernels may test |c - q| per axis first
    // (morton_kernel sorts them last: they are the last *nan_count points of the sorted order)
    if (p.x != p.x || p.y != p.y || p.z != p.z) {
      p.x = p.y = p.z = __uint_as_float(0x7fc00000u);
      atomicAdd(nan_count, 1);
    }
    p.id = ids ? ids[src] : (int32_t)src;
    points[i] = p;
  } else {
    sorted_boxes[i] = boxes[src];
  }
}

// ---- Karras radix tree ----------------------------------------------------------------------
// common-prefix length of sorted keys i and j, ties broken by position so all keys are distinct
__device__ __forceinline__ int delta(const uint64_t *__restrict__ codes, int64_t n, int64_t i, uint64_t ci,
                                     int64_t j) {
  if (j < 0 || j >= n) return -1;
  uint64_t cj = codes[j];
  if (ci == cj) return 64 + __clz((unsigned)((uint32_t)i ^ (uint32_t)j));
  return __clzll((long long)(ci ^ cj));
}

__global__ void __launch_bounds__(kBlock) karras_kernel(const uint64_t *__restrict__ codes, int64_t n,
                                                       LbvhNode *__restrict__ nodes,
                                                       int32_t *__restrict__ split_owner) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n - 1) return;
  uint64_t ci = codes[i];
  int dl = delta(codes, n, i, ci, i - 1), dr = delta(codes, n, i, ci, i + 1);
  int64_t d = dr > dl ? 1 : -1;
  int dmin = dr > dl ? dl : dr;
  int64_t lmax = 2;
  while (delta(codes, n, i, ci, i + lmax * d) > dmin) lmax <<= 1;
  int64_t l = 0;
  for (int64_t t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(codes, n, i, ci, i + (l + t) * d) > dmin) l += t;
  int64_t j = i + l * d;
  int dnode = delta(codes, n, i, ci, j);
  int64_t s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(codes, n, i, ci, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
  nodes[i].split = (int32_t)gamma;
  nodes[i].other = (int32_t)j;
  split_owner[gamma] = (int32_t)i;
}

// ---- table pyramid ----------------------------------------------------------------------------
// out[j] = union of in-elements [64j, 64j+63]; one wave per output entry
template <int SRC>  // 0: LbvhPoint, 1: LbvhBox
__global__ void __launch_bounds__(kBlock) table_kernel(const void *__restrict__ in, int64_t n_in,
                                                      LbvhBox *__restrict__ out, int64_t n_out) {
  int64_t j = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (j >= n_out) return;
  int64_t i = j * 64 + (threadIdx.x & 63);
  Box6 b = empty_box();
  if (i < n_in) {
    if (SRC == 0) {
      LbvhPoint p = ((const LbvhPoint *)in)[i];
      float c[3] = {p.x, p.y, p.z};
      grow(b, c, c);
    } else {
      LbvhBox q = ((const LbvhBox *)in)[i];
      grow(b, q.lo, q.hi);
    }
  }
  b = wave_union(b);
  if ((threadIdx.x & 63) == 0) {
    LbvhBox o;
    for (int a = 0; a < 3; a++) {
      o.lo[a] = b.lo[a];
      o.hi[a] = b.hi[a];
    }
    out[j] = o;
  }
}

// level 0 of the wide pyramid: one thread per block of LBVH_BLOCK sorted points
__global__ void __launch_bounds__(kBlock) block_box_kernel(const LbvhPoint *__restrict__ pts, int64_t n,
                                                          LbvhBox *__restrict__ out, int64_t n_out) {
  int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_out) return;
  Box6 box = empty_box();
  for (int j = 0; j < LBVH_BLOCK; j++) {
    int64_t i = b * LBVH_BLOCK + j;
    if (i < n) {
      LbvhPoint p = pts[i];
      float c[3] = {p.x, p.y, p.z};
      grow(box, c, c);
    }
  }
  LbvhBox o;
  for (int a = 0; a < 3; a++) {
    o.lo[a] = box.lo[a];
    o.hi[a] = box.hi[a];
  }
  out[b] = o;
}

// Lbvh::block_paths_device(): one thread per block of LBVH_PATH_BLOCK sorted slots walks from the root to the deepest
// internal node that still holds the whole block, keeping the last four nodes it came through
__global__ void __launch_bounds__(kBlock) block_path_kernel(const LbvhNode *__restrict__ nodes, int64_t n, int32_t *__restrict__ paths,
                                                           int64_t n_blocks) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  const int32_t lo = (int32_t)(b * LBVH_PATH_BLOCK), hi = (int32_t)((b + 1) * LBVH_PATH_BLOCK - 1 < n - 1 ? (b + 1) * LBVH_PATH_BLOCK - 1 : n - 1);
  int32_t anc[4] = {0, 0, 0, 0}, node = 0;
  for (;;) {
    const LbvhNode nd = nodes[node];
    int32_t next;
    if (hi <= nd.split) {
      if (lbvh_first(node, nd.other) == nd.split) break;  // the child is a leaf (a block of one slot)
      next = nd.split;
    } else if (lo > nd.split) {
      if (lbvh_last(node, nd.other) == nd.split + 1) break;
      next = nd.split + 1;
    } else {
      break;  // the block lies on both sides
    }
    anc[0] = anc[1], anc[1] = anc[2], anc[2] = anc[3], anc[3] = node;
    node = next;
  }
  int32_t *out = paths + b * LBVH_PATH_WORDS;
  out[0] = anc[0], out[1] = anc[1], out[2] = anc[2], out[3] = anc[3], out[4] = node;
}

struct Pyramid {
  const LbvhPoint *points;
  const LbvhBox *boxes;
  const LbvhBox *table[4];
};

__device__ __forceinline__ void grow_elem(Box6 &b, const Pyramid &py, int level, int64_t idx) {
  if (level == 0) {
    if (py.points) {
      LbvhPoint p = py.points[idx];
      float c[3] = {p.x, p.y, p.z};
      grow(b, c, c);
    } else {
      LbvhBox q = py.boxes[idx];
      grow(b, q.lo, q.hi);
    }
  } else {
    LbvhBox q = py.table[level - 1][idx];
    grow(b, q.lo, q.hi);
  }
}

__device__ __forceinline__ int32_t rope_after(int64_t end, int64_t n, const LbvhNode *__restrict__ nodes,
                                              const int32_t *__restrict__ split_owner) {
  if (end >= n - 1) return LBVH_END;
  int32_t o = split_owner[end];
  int32_t last_o = lbvh_last(o, nodes[o].other);
  return last_o == (int32_t)end + 1 ? ~((int32_t)end + 1) : (int32_t)end + 1;
}

// one thread per sorted slot i: internal node i (i < n-1) gets its box and rope, leaf i its rope
__global__ void __launch_bounds__(kBlock) fit_kernel(LbvhNode *__restrict__ nodes, int64_t n, Pyramid py,
                                                    const int32_t *__restrict__ split_owner,
                                                    int32_t *__restrict__ rope_node,
                                                    int32_t *__restrict__ rope_leaf, int write_ropes) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (write_ropes) rope_leaf[i] = rope_after(i, n, nodes, split_owner);
  if (i >= n - 1) return;
  int32_t other = nodes[i].other;
  int64_t lo = lbvh_first((int32_t)i, other), hi = (int64_t)lbvh_last((int32_t)i, other) + 1;
  if (write_ropes) rope_node[i] = rope_after(hi - 1, n, nodes, split_owner);
  Box6 b = empty_box();
  for (int level = 0; level < 5; level++) {
    while (lo < hi && (lo & 63)) grow_elem(b, py, level, lo++);
    while (lo < hi && (hi & 63)) grow_elem(b, py, level, --hi);
    if (lo >= hi) break;
    if (level == 4) {  // cannot happen: 64^4 elements per top entry covers n < 2^31
      while (lo < hi) grow_elem(b, py, level, lo++);
      break;
    }
    lo >>= 6;
    hi >>= 6;
  }
  for (int a = 0; a < 3; a++) {
    nodes[i].lo[a] = b.lo[a];
    nodes[i].hi[a] = b.hi[a];
  }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

template <typename T>
void dev_alloc(T *&p, size_t count, size_t &total) {
  OWLMI_HIP(hipMalloc((void **)&p, count * sizeof(T)));
  total += count * sizeof(T);
}

}  // namespace

Lbvh::~Lbvh() { release(); }

void Lbvh::release() {
  for (auto &w : wide_) {
    if (w) (void)hipFree(w);
    w = nullptr;
  }
  void *ptrs[] = {scene_, partials_, codes_, codes_alt_, order_, order_alt_, sort_tmp_, nodes_,
                  split_owner_, rope_node_, rope_leaf_, block_paths_, points_, boxes_, prim_id_, table_[0],
                  table_[1], table_[2], table_[3]};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  scene_ = partials_ = nullptr;
  codes_ = codes_alt_ = nullptr;
  order_ = order_alt_ = nullptr;
  sort_tmp_ = nullptr;
  nodes_ = nullptr;
  split_owner_ = rope_node_ = rope_leaf_ = block_paths_ = nullptr;
  points_ = nullptr;
  boxes_ = nullptr;
  prim_id_ = nullptr;
  for (auto &t : table_) t = nullptr;
  cap_ = 0;
  bytes_ = 0;
  built_ = false;
}

void Lbvh::reserve(int64_t n) {
  if (n <= cap_) return;
  release();
  size_t total = 0;
  dev_alloc(scene_, 8, total);
  dev_alloc(partials_, (size_t)kPartialBlocks * 6, total);
  dev_alloc(codes_, (size_t)n, total);
  dev_alloc(codes_alt_, (size_t)n, total);
  dev_alloc(order_, (size_t)n, total);
  dev_alloc(order_alt_, (size_t)n, total);
  sort_tmp_bytes_ = 0;
  OWLMI_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp_bytes_, codes_, codes_alt_, order_,
                                              order_alt_, (int)n, 0, 64, (hipStream_t)0));
  OWLMI_HIP(hipMalloc(&sort_tmp_, sort_tmp_bytes_ ? sort_tmp_bytes_ : 16));
  total += sort_tmp_bytes_;
  dev_alloc(nodes_, (size_t)(n > 1 ? n - 1 : 1), total);
  dev_alloc(split_owner_, (size_t)n, total);
  dev_alloc(rope_node_, (size_t)n, total);
  dev_alloc(rope_leaf_, (size_t)n, total);
  dev_alloc(block_paths_, ((size_t)n / LBVH_PATH_BLOCK + 1) * LBVH_PATH_WORDS, total);
  dev_alloc(points_, (size_t)n + 2 * LBVH_BLOCK, total);  // + NaN sentinels up to a whole leaf block, + one all-NaN block
  dev_alloc(boxes_, (size_t)n, total);
  dev_alloc(prim_id_, (size_t)n, total);
  int64_t m = n;
  for (int l = 0; l < 4; l++) {
    m = (m + 63) / 64;
    dev_alloc(table_[l], (size_t)m, total);
  }
  m = (n + LBVH_BLOCK - 1) / LBVH_BLOCK;
  for (int l = 0; l < LBVH_WIDE_LEVELS; l++) {
    dev_alloc(wide_[l], (size_t)m, total);
    m = (m + 63) / 64;
  }
  cap_ = n;
  bytes_ = total;
}

void Lbvh::build_wide(hipStream_t stream) {
  int64_t m = (n_ + LBVH_BLOCK - 1) / LBVH_BLOCK;
  wide_n_[0] = m;
  hipLaunchKernelGGL(block_box_kernel, dim3(blocks_for(m)), dim3(kBlock), 0, stream, points_, n_, wide_[0], m);
  OWLMI_HIP(hipGetLastError());
  wide_levels_ = 1;
  while (wide_n_[wide_levels_ - 1] > 64 && wide_levels_ < LBVH_WIDE_LEVELS) {
    const int l = wide_levels_;
    const int64_t in_n = wide_n_[l - 1], out_n = (in_n + 63) / 64;
    wide_n_[l] = out_n;
    hipLaunchKernelGGL(table_kernel<1>, dim3(blocks_for(out_n * 64)), dim3(kBlock), 0, stream,
                       (const void *)wide_[l - 1], in_n, wide_[l], out_n);
    OWLMI_HIP(hipGetLastError());
    wide_levels_++;
  }
}

LbvhWideView Lbvh::wide_view() const {
  LbvhWideView w;
  for (int l = 0; l < LBVH_WIDE_LEVELS; l++) {
    w.level[l] = l < wide_levels_ ? wide_[l] : nullptr;
    w.count[l] = l < wide_levels_ ? (int32_t)wide_n_[l] : 0;
  }
  w.levels = point_mode_ ? wide_levels_ : 0;
  return w;
}

void Lbvh::sort_and_tree(hipStream_t stream) {
  OWLMI_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp_, sort_tmp_bytes_, codes_, codes_alt_, order_,
                                              order_alt_, (int)n_, 0, 64, stream));
  if (n_ > 1) {
    hipLaunchKernelGGL(karras_kernel, dim3(blocks_for(n_ - 1)), dim3(kBlock), 0, stream, codes_alt_, n_,
                       nodes_, split_owner_);
    OWLMI_HIP(hipGetLastError());
  }
}

void Lbvh::fit(hipStream_t stream) {
  // level-0 table from elements, then 64-ary up
  int64_t m = (n_ + 63) / 64;
  table_n_[0] = m;
  if (point_mode_)
    hipLaunchKernelGGL(table_kernel<0>, dim3(blocks_for(m * 64)), dim3(kBlock), 0, stream,
                       (const void *)points_, n_, table_[0], m);
  else
    hipLaunchKernelGGL(table_kernel<1>, dim3(blocks_for(m * 64)), dim3(kBlock), 0, stream,
                       (const void *)boxes_, n_, table_[0], m);
  OWLMI_HIP(hipGetLastError());
  for (int l = 1; l < 4; l++) {
    int64_t in_n = table_n_[l - 1];
    int64_t out_n = (in_n + 63) / 64;
    table_n_[l] = out_n;
    hipLaunchKernelGGL(table_kernel<1>, dim3(blocks_for(out_n * 64)), dim3(kBlock), 0, stream,
                       (const void *)table_[l - 1], in_n, table_[l], out_n);
    OWLMI_HIP(hipGetLastError());
  }
  Pyramid py;
  py.points = point_mode_ ? points_ : nullptr;
  py.boxes = point_mode_ ? nullptr : boxes_;
  for (int l = 0; l < 4; l++) py.table[l] = table_[l];
  hipLaunchKernelGGL(fit_kernel, dim3(blocks_for(n_)), dim3(kBlock), 0, stream, nodes_, n_, py,
                     split_owner_, rope_node_, rope_leaf_, 1);
  OWLMI_HIP(hipGetLastError());
}

void Lbvh::build_from_points(const float *d_xyz, int64_t n, hipStream_t stream, const int32_t *d_ids) {
  if (n <= 0 || n >= 0x7fffffffLL) throw HipError{"Lbvh: primitive count out of range"};
  built_ = false;  // until the last launch below has been issued without an error
  reserve(n);
  n_ = n;
  point_mode_ = true;
  int pb = (int)std::min<int64_t>(kPartialBlocks, (n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(scene_partial_kernel<true>, dim3(pb), dim3(kBlock), 0, stream, d_xyz,
                     (const LbvhBox *)nullptr, n, partials_);
  hipLaunchKernelGGL(scene_final_kernel, dim3(1), dim3(kBlock), 0, stream, partials_, pb, scene_);
  hipLaunchKernelGGL(morton_kernel<true>, dim3(blocks_for(n)), dim3(kBlock), 0, stream, d_xyz,
                     (const LbvhBox *)nullptr, n, scene_, codes_, order_);
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipMemsetAsync(nan_count(), 0, sizeof(int32_t), stream));
  sort_and_tree(stream);
  hipLaunchKernelGGL(gather_kernel<true>, dim3(blocks_for(n + 2 * LBVH_BLOCK)), dim3(kBlock), 0, stream, d_xyz,
                     (const LbvhBox *)nullptr, n, order_alt_, points_, (LbvhBox *)nullptr, prim_id_, d_ids, nan_count(), (int32_t *)order_);
  OWLMI_HIP(hipGetLastError());
  fit(stream);
  build_wide(stream);
  if (n > 1) {
    const int64_t n_blocks = (n + LBVH_PATH_BLOCK - 1) / LBVH_PATH_BLOCK;
    hipLaunchKernelGGL(block_path_kernel, dim3(blocks_for(n_blocks)), dim3(kBlock), 0, stream, nodes_, n, block_paths_, n_blocks);
    OWLMI_HIP(hipGetLastError());
  }
  built_ = true;
}

void Lbvh::build_from_boxes(const LbvhBox *d_boxes, int64_t n, hipStream_t stream) {
  if (n <= 0 || n >= 0x7fffffffLL) throw HipError{"Lbvh: primitive count out of range"};
  reserve(n);
  n_ = n;
  point_mode_ = false;
  int pb = (int)std::min<int64_t>(kPartialBlocks, (n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(scene_partial_kernel<false>, dim3(pb), dim3(kBlock), 0, stream,
                     (const float *)nullptr, d_boxes, n, partials_);
  hipLaunchKernelGGL(scene_final_kernel, dim3(1), dim3(kBlock), 0, stream, partials_, pb, scene_);
  hipLaunchKernelGGL(morton_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, stream,
                     (const float *)nullptr, d_boxes, n, scene_, codes_, order_);
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipMemsetAsync(nan_count(), 0, sizeof(int32_t), stream));
  sort_and_tree(stream);
  hipLaunchKernelGGL(gather_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, stream,
                     (const float *)nullptr, d_boxes, n, order_alt_, (LbvhPoint *)nullptr, boxes_, prim_id_, (const int32_t *)nullptr, nan_count(), (int32_t *)nullptr);
  OWLMI_HIP(hipGetLastError());
  fit(stream);
  built_ = true;
}

void Lbvh::refit_boxes(const LbvhBox *d_boxes, hipStream_t stream) {
  if (!built_ || point_mode_) throw HipError{"Lbvh::refit_boxes: no box tree to refit"};
  hipLaunchKernelGGL(gather_kernel<false>, dim3(blocks_for(n_)), dim3(kBlock), 0, stream,
                     (const float *)nullptr, d_boxes, n_, order_alt_, (LbvhPoint *)nullptr, boxes_, prim_id_, (const int32_t *)nullptr, nan_count(), (int32_t *)nullptr);
  OWLMI_HIP(hipGetLastError());
  fit(stream);
}

LbvhView Lbvh::view() const {
  LbvhView v;
  v.nodes = nodes_;
  v.rope_node = rope_node_;
  v.rope_leaf = rope_leaf_;
  v.points = point_mode_ ? points_ : nullptr;
  v.boxes = point_mode_ ? nullptr : boxes_;
  v.prim_id = prim_id_;
  v.n = (int32_t)n_;
  v.root = n_ > 1 ? 0 : ~0;
  v.nan_count = nan_count();
  return v;
}

void Lbvh::download(LbvhNode *nodes, int32_t *rope_node, int32_t *rope_leaf, int32_t *prim_id,
                    hipStream_t stream) const {
  if (!built_) throw HipError{"Lbvh::download: not built"};
  if (nodes && n_ > 1)
    OWLMI_HIP(hipMemcpyAsync(nodes, nodes_, (size_t)(n_ - 1) * sizeof(LbvhNode), hipMemcpyDeviceToHost, stream));
  if (rope_node && n_ > 1)
    OWLMI_HIP(hipMemcpyAsync(rope_node, rope_node_, (size_t)(n_ - 1) * 4, hipMemcpyDeviceToHost, stream));
  if (rope_leaf) OWLMI_HIP(hipMemcpyAsync(rope_leaf, rope_leaf_, (size_t)n_ * 4, hipMemcpyDeviceToHost, stream));
  if (prim_id) OWLMI_HIP(hipMemcpyAsync(prim_id, prim_id_, (size_t)n_ * 4, hipMemcpyDeviceToHost, stream));
  OWLMI_HIP(hipStreamSynchronize(stream));
}

void Lbvh::download_tables(int32_t *split_owner, int32_t *block_paths, hipStream_t stream) const {
  if (!built_) throw HipError{"Lbvh::download_tables: not built"};
  if (split_owner && n_ > 1)
    OWLMI_HIP(hipMemcpyAsync(split_owner, split_owner_, (size_t)(n_ - 1) * 4, hipMemcpyDeviceToHost, stream));
  if (block_paths && block_paths_device())
    OWLMI_HIP(hipMemcpyAsync(block_paths, block_paths_, (size_t)((n_ + LBVH_PATH_BLOCK - 1) / LBVH_PATH_BLOCK) * LBVH_PATH_WORDS * 4,
                             hipMemcpyDeviceToHost, stream));
  OWLMI_HIP(hipStreamSynchronize(stream));
}

}  // namespace owlmi
