// lbvh.h -- host API of the HIP LBVH builder (Morton sort + Karras radix tree + fence-free fit).
// Replaces what the reference delegates to optixAccelBuild (owl/UserGeomGroup.cpp:161-217 BUILD,
// :75-76/:199 UPDATE on refit).  Device layout: include/owl/lbvh_device.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "owl/lbvh_device.h"

#define LBVH_PATH_BLOCK 64
#define LBVH_PATH_WORDS 5

namespace owlmi {

struct HipError {
  std::string what;
};

#define OWLMI_HIP(call)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      throw ::owlmi::HipError{std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" + \
                              __FILE__ + ":" + std::to_string(__LINE__) + ")"};                \
  } while (0)

// One LBVH over n primitives.  Two flavours of leaf payload:
//   points: primitives are points (the TrueKNN / DBSCAN pattern: every box is centre +- rad with
//           one shared rad).  Node boxes bound the *centres*; the radius is added at test time,
//           so changing it needs no refit at all.
//   boxes : arbitrary per-primitive AABBs as written by a user bounds program; node boxes bound
//           the boxes; refit() recomputes them on the same topology (OptiX UPDATE semantics).
class Lbvh {
 public:
  Lbvh() = default;
  ~Lbvh();
  Lbvh(const Lbvh &) = delete;
  Lbvh &operator=(const Lbvh &) = delete;

  // xyz: device pointer to n packed fp32 triples (the reference's Sphere buffer, 12 B/point).
  // d_ids (optional): identity of each point as the caller wants it reported (e.g. a global index
  // when the buffer is one shard of a larger set); default is the position in d_xyz.
  void build_from_points(const float *d_xyz, int64_t n, hipStream_t stream, const int32_t *d_ids = nullptr);
  // boxes: device pointer to n {lo[3],hi[3]} (24 B, owl::box3f) in caller primitive order.
  void build_from_boxes(const LbvhBox *d_boxes, int64_t n, hipStream_t stream);
  // same topology, new boxes (only for build_from_boxes trees)
  void refit_boxes(const LbvhBox *d_boxes, hipStream_t stream);

  LbvhView view() const;
  LbvhWideView wide_view() const;  // point trees only
  const float *scene_device() const { return scene_; }  // 6 floats: lo xyz, hi xyz of the built set
  // device counter behind the scene box: points with a NaN coordinate (they sort last)
  int32_t *nan_count() const { return reinterpret_cast<int32_t *>(scene_ + 6); }
  // split_owner[s] = the internal node that splits its range after sorted position s (n - 1 entries): a node's parent in O(1)
  // -- internal node i is a left child iff it is its range's LAST position (parent = split_owner[i]), else a right child
  // (parent = split_owner[i - 1]); valid as long as the tree is
  const int32_t *split_owner_device() const { return split_owner_; }
  // point trees: per block of LBVH_PATH_BLOCK consecutive sorted slots, LBVH_PATH_WORDS nodes -- the deepest internal node whose
  // range holds the whole block (last word) and its four nearest ancestors, the farthest first (the root where the path is
  // shorter): a walk down to one of the block's slots can start there instead of at the root
  const int32_t *block_paths_device() const { return point_mode_ && n_ > 1 ? block_paths_ : nullptr; }
  // point trees: row_slot[row] = the sorted slot of the caller's row `row` (the inverse of prim_id), n entries
  const int32_t *row_slot_device() const { return point_mode_ && built_ ? reinterpret_cast<const int32_t *>(order_) : nullptr; }
  int64_t size() const { return n_; }
  bool built() const { return built_; }
  void clear() { built_ = false; }  // marks the tree unusable (a failed rebuild); memory stays reserved
  bool has_points() const { return points_ != nullptr && point_mode_; }
  size_t device_bytes() const { return bytes_; }

  // debug / test export (host copies)
  void download(LbvhNode *nodes, int32_t *rope_node, int32_t *rope_leaf, int32_t *prim_id,
                hipStream_t stream) const;
  // split_owner: n - 1 entries; block_paths: ceil(n / LBVH_PATH_BLOCK) * LBVH_PATH_WORDS (point trees with n > 1)
  void download_tables(int32_t *split_owner, int32_t *block_paths, hipStream_t stream) const;

 private:
  void reserve(int64_t n);
  void release();
  void sort_and_tree(hipStream_t stream);
  void fit(hipStream_t stream);

  int64_t n_ = 0, cap_ = 0;
  bool built_ = false, point_mode_ = false;
  size_t bytes_ = 0;
  float *scene_ = nullptr;     // 6 floats + partials
  float *partials_ = nullptr;  // kPartialBlocks*6
  uint64_t *codes_ = nullptr, *codes_alt_ = nullptr;
  uint32_t *order_ = nullptr, *order_alt_ = nullptr;
  void *sort_tmp_ = nullptr;
  size_t sort_tmp_bytes_ = 0;
  LbvhNode *nodes_ = nullptr;
  int32_t *split_owner_ = nullptr, *rope_node_ = nullptr, *rope_leaf_ = nullptr, *block_paths_ = nullptr;
  LbvhPoint *points_ = nullptr;
  LbvhBox *boxes_ = nullptr;  // sorted boxes (box mode)
  int32_t *prim_id_ = nullptr;
  LbvhBox *table_[4] = {nullptr, nullptr, nullptr, nullptr};
  int64_t table_n_[4] = {0, 0, 0, 0};
  LbvhBox *wide_[LBVH_WIDE_LEVELS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int64_t wide_n_[LBVH_WIDE_LEVELS] = {0, 0, 0, 0, 0, 0};
  int wide_levels_ = 0;
  void build_wide(hipStream_t stream);
};

}  // namespace owlmi
