// trueknn_engine.h -- host-side engine object behind include/owlknn.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "knn_device.h"
#include "lbvh.h"
#include "owlknn.h"

namespace owlmi {

struct RoundsExceeded {};
struct ArgError {
  int code;
  std::string what;
};

struct SolveArgs {
  int k = 0;
  float start_radius = 0;
  int max_rounds = 64;
  int32_t *d_idx = nullptr;
  float *d_dist = nullptr;
  int64_t *d_isect = nullptr;
  tknnNeigh *d_fb = nullptr;
  int32_t *d_levels = nullptr;
  bool allow_unfinished = false;
  const float *d_start_radii = nullptr;  // tknnSolveOptions.d_start_radii: per row, or null (one start radius for all)
  int phase = 0;  // tknnSolveOptions.phase: 0 every query, 1 interior queries in the own tree only, 2 boundary queries
};

// smallest register-list capacity instantiated for k, or -1
int list_capacity_for(int k);

// dbscan.hip: out[seg[i]] = min(out[seg[i]], val[i]) for seg[i] >= 0 (tknnSegmentMin)
void db_segment_min(const int32_t *d_seg, const int64_t *d_val, int64_t n, int64_t *d_out, hipStream_t s);

// test hook: the wave kernel's exact candidate thresholds for (q, r) pairs (trueknn_wave.hip)
void debug_thresholds(const float *d_q, const float *d_r, int64_t n, float *d_lo, float *d_hi, hipStream_t s);

class Engine {
 public:
  Engine();
  ~Engine();
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;

  void build(const float *d_xyz, const int32_t *d_ids, int64_t n, tknnBuildInfo *info, hipStream_t s);
  void set_halo(const float *d_xyz, const int32_t *d_ids, int64_t m, hipStream_t s);
  LbvhView halo_view() const;
  // halo_select.hip: my points inside any box of each peer, as 16-byte wire rows (count pass: d_rows == nullptr)
  // (d_caps: the one-pass form -- rows into segments of those capacities, exact counts out, what does not fit dropped)
  void halo_select(const float *d_boxes, const int32_t *d_box_peer, int32_t nboxes, int32_t npeers, int64_t *d_counts,
                   const int64_t *d_offsets, float *d_rows, hipStream_t s, const int64_t *d_caps = nullptr);
  double expected_box_population(float radius) const;
  void solve(const SolveArgs &sa, int kernel, tknnSolveInfo *info, hipStream_t s);
  // rewrites the rows whose k-th distance exceeds their final box half-width with exact kNN; returns how many
  int64_t repair_exact(int k, float start_radius, const int32_t *d_levels, int32_t *d_idx, float *d_dist, hipStream_t s);
  // dbscan.hip; with core_label (per row: the label the caller gave each core point, < 0 for others) only
  // the assignment runs: core points keep their label, others take the smallest among their core neighbours
  void dbscan(float eps, int min_pts, int32_t *d_labels, uint8_t *d_core, int32_t *d_counts, tknnDbscanInfo *info,
              hipStream_t s, const int32_t *core_label = nullptr);
  // "eps auto-grown" (BASELINE config 5; spec: oracle/dbscan_oracle.c dbref_dbscan_auto)
  int64_t dbscan_noise(float eps, int min_pts, uint8_t *d_noise, hipStream_t s);  // tknnDbscanNoise
  void db_read_stats(hipStream_t s);
  void dbscan_auto(float eps0, int min_pts, double max_noise, int max_rounds, int32_t *d_labels, uint8_t *d_core,
                   tknnDbscanAutoInfo *info, hipStream_t s);
  bool built() const { return bvh_.built(); }
  int device() const { return device_; }
  int64_t size() const { return bvh_.size(); }
  const Lbvh &tree() const { return bvh_; }

 private:
  void solve_lane(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s);
  // lane rounds over the queries another kernel left with done_[slot] == 0, each from next_level_[slot]
  void continue_lane(const SolveArgs &sa, int first_level, tknnSolveInfo *info, hipStream_t s);
  void lane_rounds(const SolveArgs &sa, int first_level, bool fresh, tknnSolveInfo *info, hipStream_t s);
  void solve_wave(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s, bool only_unfinished = false);  // trueknn_wave.hip
  static bool wave_kernel_available();                                        // trueknn_wave.hip
  // trueknn_team.hip; returns false if a packet needed more leaf blocks than the kernel can name
  bool solve_team(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s);
  static bool team_kernel_supports(int k);
  // trueknn_team.hip: 64 < k <= TKNN_MAX_K, the lists in memory (bigk_walk_kernel)
  static bool bigk_supports(int k);
  void solve_bigk(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s);
  // trueknn_team.hip: redo the rows flagged in tie_ with the reference's order of exact-distance ties
  void fix_ties(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s);
  void reset_stat_stripes(hipStream_t s);
  void fetch_stat_stripes(hipStream_t s);
  void fold_stat_stripes(bool with_min);
  void launch_tie_fix(const SolveArgs &sa, const int32_t *slots, int32_t nslots, int blocks, hipStream_t s,
                      const int32_t *d_slot_count = nullptr, int64_t expected_rows = 0);
  int first_step_estimate(const SolveArgs &sa) const;
  float scene_[6] = {0, 0, 0, 0, 0, 0};  // bounds of the built point set (host copy)

  int device_ = 0;
  Lbvh bvh_;
  Lbvh halo_;
  int64_t halo_n_ = 0;
  // A phase-1 solve (queries no halo point can reach) runs beside the halo exchange: tknnSetHalo may rebuild
  // halo_ on another host thread meanwhile, so that solve never looks at it.
  bool ignore_halo_ = false;
  int64_t halo_count() const { return ignore_halo_ ? 0 : halo_n_; }
  uint8_t *boundary_ = nullptr;  // per sorted slot: 1 if the last tknnHaloSelect count pass found the point inside a peer's box
  bool boundary_valid_ = false;
  uint8_t *done_ = nullptr;
  int64_t *isect_sorted_ = nullptr;
  int32_t *next_level_ = nullptr;
  int32_t *tie_list_ = nullptr;  // the first kTieListCap flagged slots, in the order the kernels met them
  uint8_t *tie_ = nullptr;  // per sorted slot: 0, or 1 + the level at which the query finished with exact-distance ties in reach of its row
  int64_t state_cap_ = 0;
  // counters_: kCounters words (knn_device.h) + kDbStripes x 8 words of RT-DBSCAN's work counters, striped over the workgroups
  // (dbscan.hip: a launch over 50 M points would otherwise queue 400 000 atomics on two addresses); h_counters_: 16 + the stripes
  static constexpr int kDbStripes = 32;
  // ... + the group-union kernel's per-XCD packet cursors, kDbCursorStride words apart (a cache line each: dbscan.hip)
  static constexpr int kDbCursorStride = 32;
  // ... + the packet kernel's end-of-wave statistics, striped over the workgroups: kStatStripes stripes of kStatStride words
  // (a cache line each; within a stripe the words are counters_'s [1] .. [9]), folded on the host (trueknn_team.hip)
  static constexpr int kStatStripes = 32, kStatStride = 16, kStatBase = kCounters + kDbStripes * 8 + 8 * kDbCursorStride;
  static constexpr int kCounterWords = kStatBase + kStatStripes * kStatStride;
  unsigned long long *counters_ = nullptr, *h_counters_ = nullptr;
  void *wave_ws_ = nullptr;
  size_t wave_ws_bytes_ = 0;
  int db_union_resident_ = 0;     // workgroups of db_group_union_kernel this engine's device holds at once (0: not asked yet)
  bool db_force_point_ = false;   // dbscan(): per-point unions (the fallback when a packet walk of the group unions ran out of stack)
  bool wave_force_redo_ = false;  // TKNN_WAVE_FORCE_REDO (tests): treat every wave-kernel solve as if its LDS stack had overflowed
  int wave_leaf_max_ = 16;  // subtrees of at most this many points are streamed as one range (TKNN_LEAF_MAX)
  int32_t *slot_list_ = nullptr;  // compact list of the sorted slots the team kernel handed over (+ its length)
  int64_t slot_list_cap_ = 0;
  unsigned long long *halo_mask_ = nullptr;  // per leaf block: peers it may have points for (+ 64 cursors)
  int64_t halo_mask_cap_ = 0;
  hipEvent_t ev_a_ = nullptr, ev_b_ = nullptr, ev_c_ = nullptr, ev_d_ = nullptr, ev_e_ = nullptr, ev_f_ = nullptr;
  // dbscan(): the walks of the points that are not core run beside the group unions on a stream of their own
  hipStream_t db_side_ = nullptr;
  hipEvent_t ev_side_a_ = nullptr, ev_side_b_ = nullptr;
  bool db_side_pending_ = false;  // a side launch was recorded in ev_side_b_ and no stream has been made to wait for it yet
  hipEvent_t ev_g_ = nullptr, ev_h_ = nullptr;  // dbscan(): around what runs between the two union launches
  // the packet kernel's solve launches the tie pass behind itself, before its one host round trip: set if
  // that launch has seen every flagged row (no tail ran, the list held them all)
  bool ties_early_ = false;
  int64_t early_tie_rows_ = 0, early_tie_left_ = 0;
  float early_tie_ms_ = 0;
};

}  // namespace owlmi
