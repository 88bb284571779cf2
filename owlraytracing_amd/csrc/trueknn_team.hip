// trueknn_team.hip -- the team TrueKNN kernel (TKNN_KERNEL_TEAM), k <= 64 (k > 16: two list registers per lane, k > 32: four).
//
// The wave-packet kernel (trueknn_wave.hip) broadcasts every candidate of a 64-query packet to all
// 64 lanes; on MI355X it is VALU-issue-bound with ~2.5 % useful lanes (profiles/r01_wave_v2_*),
// because the packet's candidate union is ~40x one query's own box content.  This kernel turns
// the work around: a wave is FOUR TEAMS OF 16 LANES; a team handles one query at a time and its
// lanes are the 16 CANDIDATES of one leaf block (LBVH_BLOCK consecutive Morton-sorted points,
// one coalesced 256-byte read) of that query's OWN block list.
//
// Per packet of 64 Morton-consecutive queries and per radius level (hostCode.cpp:285-340 rounds):
//   1. lanes = queries: LDS query records and conservative query boxes for the gather.
//   2. lanes = child boxes: the wave walks the 64-ary block pyramid (LbvhWideView) depth-first
//      with an LDS stack, one wide node per step, against the union box of the packet.
//   3. every surviving leaf block is tested against the 64 individual query boxes
//      (lanes = queries again, block box broadcast by v_readlane); blocks some query needs are
//      appended to the packet's block list and to the per-query lists of byte slots (no atomics).
//   4. COUNT pass, teams over the compacted list of active queries: count candidates in the
//      query's box (deviceCode.cu:74) and those other than the query itself (:103).
//   5. queries with >= k others are finished at this level (deviceCode.cu:118): SELECT pass,
//      teams over the compacted list of finishing queries: distances, gate, and a team-parallel
//      sorted insert -- lane j of the team holds the j-th best (dist,index) key; a new key is
//      broadcast to the team and every lane decides locally whether it keeps, takes the key, or
//      takes its left neighbour's entry (DPP row shift).  Rows are written straight from the
//      team: lane j stores neighbour j (coalesced 4*k bytes per array).
// Results are bit-identical to the other kernels: the candidate test is the literal one wherever a
// cheap bound cannot decide it (see the record layout below), same distance arithmetic, keys ordered
// by (dist, index).
#include "knn_thresholds.h"  // knn_gate_from_worst
#include "trueknn_engine.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// Diagnostic build only (make DIAG=1 -> libowl_mi355x_diag.so, never the shipped library): lets
// scripts/diag_team.py price the phases by switching them off; results are wrong when any bit is set.
#ifndef TKNN_DIAG_BUILD
#define TKNN_DIAG_BUILD 0
#endif

namespace owlmi {

namespace {

constexpr int kTeamBlock = 64;      // one wave per workgroup: LDS, not the block shape, limits residency
// leaf blocks one packet may need per level.  A per-query list names them by byte slots: 256 -- except with four list
// registers per lane (k > 48), where the lists of the last level outgrow that (10 M uniform points at k = 64: 36 % of the queries
// were handed to the team walk, two thirds of the solve): 512, the ninth bit of a slot in a word of its own
// (four words per query)
#ifndef TKNN_MAX_BLOCKS_3
#define TKNN_MAX_BLOCKS_3 256
#endif
#ifndef TKNN_MAX_BLOCKS_4
#define TKNN_MAX_BLOCKS_4 512
#endif
constexpr int max_blocks(int nreg) { return nreg == 4 ? TKNN_MAX_BLOCKS_4 : (nreg == 3 ? TKNN_MAX_BLOCKS_3 : 256); }
#ifndef TKNN_MAX_PER_QUERY
#define TKNN_MAX_PER_QUERY 72  // (k <= 16)
#endif
// leaf blocks one query may need per level, by the number of list registers per lane (k <= 16 / 32 / 48 / 64).  LDS is
// allocated in granules of 1 280 bytes, 128 of them per CU (lds_workgroups_per_cu below), so the sizes are chosen by the granule:
//   1: 72 slots (lists of the benchmark run to 60)             10 192 bytes = 8 granules, 16 waves per CU (the register limit)
//   2, 3: 88 slots                                             11 472 bytes = 9 granules, 14 waves
//         (round 4; 96 slots = 10 granules = 12 waves cost 10 % at k = 17 .. 44 on uniform, mixture and taxi-like sets of
//         10 M points; 72 would hand over a third of the queries of the uniform set at k = 32, 88 none)
//   4: 124 slots and 512 blocks per packet                     16 400 bytes = 13 granules, 9 waves
//         (round 4: with 256 blocks and 96 slots 36 % of the queries of 10 M uniform points were handed over at k = 64;
//         116 slots and 384 blocks = 12 granules = 10 waves: the packet kernel 8 % faster, the hand-overs eat it)
#ifndef TKNN_MAX_PER_QUERY_2
#define TKNN_MAX_PER_QUERY_2 88
#endif
#ifndef TKNN_MAX_PER_QUERY_3
#define TKNN_MAX_PER_QUERY_3 88
#endif
#ifndef TKNN_MAX_PER_QUERY_4
#define TKNN_MAX_PER_QUERY_4 124
#endif
constexpr int max_per_query(int nreg) {
  return nreg == 1 ? TKNN_MAX_PER_QUERY : (nreg == 4 ? TKNN_MAX_PER_QUERY_4 : (nreg == 3 ? TKNN_MAX_PER_QUERY_3 : TKNN_MAX_PER_QUERY_2));
}
// One-wave workgroups of `lds` bytes a CU holds at once: LDS is allocated in granules of 1 280 bytes (measured:
// scripts/microbench/lds_granule.hip, profiles/r04_lds_granule.txt -- the residency of a launch steps at every multiple of 1 280
// bytes, hipOccupancyMaxActiveBlocksPerMultiprocessor divides by the byte and says 13 where 12 fit)
inline int lds_workgroups_per_cu(size_t lds) {
  const size_t granule = 1280, total = 160 * 1024;  // gfx950
  return lds ? (int)((total / granule) / ((lds + granule - 1) / granule)) : 1 << 20;
}
// slots (in fours) a wave takes per turn at the tie pass's work cursor (TeamArgs::grab): so many that a wave comes `turns` times in
// all, at most `cap`; TKNN_GRAB overrides (measurements)
inline int grab_for(int64_t count, int blocks, int turns, int cap) {
  if (const char *g = getenv("TKNN_GRAB")) return std::max(1, atoi(g));
  return (int)std::max<int64_t>(1, std::min<int64_t>(cap, count / ((int64_t)std::max(1, blocks) * 4 * turns)));
}
// list registers per lane for k: 16 entries each.  Three (k = 33 .. 48; round 4) spare those k the four-register
// instantiation's fourth merge step and its 13 granules of LDS (10 M uniform points: 22.2 -> 15.0 ms at k = 33, 14.2 -> 12.7 at k = 32)
#ifndef TKNN_NREG3
#define TKNN_NREG3 1
#endif
inline int nreg_for(int k) { return k <= 16 ? 1 : (k <= 32 ? 2 : (k <= 48 && TKNN_NREG3 ? 3 : 4)); }
#ifndef TKNN_MERGE_AT
#define TKNN_MERGE_AT 12  // buffered candidates of some team at the end of a group of four blocks that trigger a merge
#endif
#ifndef TKNN_SORT_FIRST
#define TKNN_SORT_FIRST 1  // k <= 16: 1 = sort the query's own block on the spot instead of buffering its candidates
#endif
#ifndef TKNN_TEAM_WAVES
#define TKNN_TEAM_WAVES 4  // waves per SIMD the packet kernel's register allocation aims at
#endif
#ifndef TKNN_WALK_WAVES
// waves per SIMD the walks' register allocation aims at (0: the compiler's choice, 3 with 131 .. 147 registers).  Round 4: 4 --
// a few words of scratch outside the loops; 10 M points: the hand-over walk 12 .. 18 % faster (taxi-like set at k = 10: 4.4 -> 3.9 ms,
// uniform at k = 48: 5.3 -> 4.4 ms), the k > 64 walk 17 % (k = 65: 209 -> 174 ms); 5 and 6 spill inside the loops and lose
#define TKNN_WALK_WAVES 4
#endif
#if TKNN_WALK_WAVES
#define TKNN_WALK_ATTR __attribute__((amdgpu_waves_per_eu(TKNN_WALK_WAVES)))
#else
#define TKNN_WALK_ATTR
#endif
#ifndef TKNN_BIGK_WAVES
#define TKNN_BIGK_WAVES 4
#endif
#if TKNN_BIGK_WAVES
#define TKNN_BIGK_ATTR __attribute__((amdgpu_waves_per_eu(TKNN_BIGK_WAVES)))
#else
#define TKNN_BIGK_ATTR
#endif
constexpr int kWalkBlocksPerCu = 4 * (TKNN_WALK_WAVES > 4 ? TKNN_WALK_WAVES : 4);  // one-wave workgroups of the walks' launches
// the per-XCD packet counters: 256 bytes apart, behind the kCounters words (the words RT-DBSCAN stripes its statistics over: not in
// use during a solve).  Side by side in one cache line (round 3: counters[16 + x]) their atomics -- one per packet, 156 000 per launch
// of the benchmark -- queued at one place (see TeamArgs::grab for what a turn at one address costs)
constexpr int kXcdCounter = kCounters, kXcdCounterStride = 32;
// ... and the packet kernel's end-of-wave statistics: five atomics per wave on one cache line, 20 000 at the end of a launch of the
// benchmark, cost it 0.10 of 6.6 ms (measured by leaving them out).  Striped over the workgroups, a cache line per stripe (the layout
// is Engine::kStatBase .. in trueknn_engine.h: behind RT-DBSCAN's words), folded by the host into h_counters_[1 .. 9]
// (kStatStripes, kStatStride, kStatBase: knn_device.h)
constexpr int kTeamStack = 192;     // wide-pyramid stack entries per wave
constexpr int kQrecStride = 6;      // floats per LDS query record (layout below)
constexpr int kScanBudget = 16384;  // leaf blocks one packet-level may test against its queries before it is handed over
constexpr int kMaxStep = 2;          // radius levels one gather may serve (the count slots and the inner-box test assume <= 2)
// LDS per wave: query records | block list | per-query block lists | counts, query list / stack.
// The pyramid stack is live only during the gather, the counts and the query list only during the
// passes, so they share one region.  Every KB counts: LDS, not registers, limits residency.
constexpr int kLdsQrec = 64 * kQrecStride * 4;
constexpr int kLdsCnt = 64 * 2 * 4;  // per query: candidates at the inner level, at the outer level | self << 31
constexpr int kLdsList = 64 * 4 + 16;  // + the bucket-presence word of the list builder
constexpr int kLdsStack = kTeamStack * 4;
constexpr int kLdsShared = (kLdsCnt + kLdsList) > kLdsStack ? (kLdsCnt + kLdsList) : kLdsStack;
constexpr int kLdsSharedPadded = (kLdsShared + 15) & ~15;  // the entry lists are read 16 bytes at a time
// per team (every list size since round 3): candidates that passed the gate since the last merge into the team's sorted list, as
// 64-bit (dist, index) keys, and how many there are.  At most 16 when a block is tested, so 32 hold any block.
constexpr int kCandCapacity = 32;
constexpr int kCandCap = kCandCapacity;
// LDS per wave, by the number of list registers per lane: query records | block list | per-query block lists |
// counts, query list / pyramid stack | per-team entry lists | per-team candidate buffers
template <int NREG>
struct TeamLayout {
  static constexpr int kMaxPerQuery = max_per_query(NREG);
  static constexpr int kMaxBlocks = max_blocks(NREG);
  static constexpr int kLdsBlk = kMaxBlocks * 4;
  static constexpr int kLdsMask = 64 * kMaxPerQuery;  // per-query lists of block slots (bytes)
  static constexpr int kHiWords = kMaxBlocks > 256 ? (kMaxPerQuery + 31) / 32 : 0;  // per query: the slots' ninth bits
  static constexpr int kLdsHi = 64 * kHiWords * 4;
  // per team: the block entries of the query it serves, resolved and in visit order (+ 4: a pass
  // prefetches one group of four past the end of a list rounded up to whole groups)
  static constexpr int kEntStride = kMaxPerQuery + 4;
  static constexpr int kLdsEnt = 4 * kEntStride * 4;
  static constexpr int kLdsCand = 4 * kCandCap * 8;
  static constexpr int kOffMask = kLdsQrec + kLdsBlk;
  static constexpr int kOffHi = kOffMask + kLdsMask;
  static constexpr int kOffShared = kOffHi + kLdsHi;
  static constexpr int kOffEnt = kOffShared + kLdsSharedPadded;
  static constexpr int kOffCand = kOffEnt + kLdsEnt;
  static constexpr int kTeamLds = kOffCand + kLdsCand;
  static_assert(kEntStride % 4 == 0 && kOffEnt % 16 == 0, "entry lists must be 16-byte aligned");
  static_assert(kLdsMask % 4 == 0, "the ninth bits are words");
  static_assert(kMaxPerQuery <= 124 && kMaxPerQuery / 4 < 32, "list-length buckets of the pass lists are one bit each of a 32-bit word");
};

// LDS query record: [0..2] q, [3] id, [4] radius of the box the pass works in (outermost level of
// the step for COUNT, the finishing level for SELECT), [5] packed: #blocks | position of the query's
// own block in its list << 8.  (The margin of the fast box test and the query's row are recomputed /
// re-read by the passes: 512 bytes of LDS decide between 15 and 16 waves per CU.)
//
// The candidate test (deviceCode.cu:38-56: fl(c-r) <= q <= fl(c+r) per axis) is done in two tiers.
// With t = max |fl(c_a - q_a)| over the axes (the differences the distance needs anyway) and the
// margin M = 2^-21 (max_a |q_a| + 2r):  t <= r - M proves the literal test true, t > r + M (or t
// NaN) proves it false -- the rounding of fl(c-r), fl(c+r) and fl(c-q) is below 2^-23 (|q| + 2r),
// an eighth of M -- and only candidates inside that 2M band (about 1e-4 of them) run the literal
// test.  Three instructions instead of stored per-axis thresholds, and 2.5 KB less LDS per wave.
struct TeamArgs {
  LbvhView bvh, halo;
  LbvhWideView wide[2];
  float start_radius;
  // per-query radius schedule (tknnSolveOptions.d_start_radii): per ROW the radius query `row` starts
  // with, or null -- then every query starts with start_radius, as in the reference (hostCode.cpp:185,325)
  const float *start_radii;
  int k;
  int max_rounds;
  int allow_unfinished;
  int first_step;  // levels the first gather of every packet serves (density estimate, 1..kMaxStep)
  int first_ext;   // ... and whether that gather also lists the blocks of the level after them (see the level loop)
  float tie_span;  // sqrt(number of axes along which the points differ), rounded up: d <= tie_span * Chebyshev distance
  int diag;        // TKNN_DIAG_BUILD only: 1 skip inserts, 2 skip SELECT passes, 4 skip COUNT passes, 8 skip per-block query tests,
                   // 16 / 32 step and gather statistics (atomics: slow), 64 every block visit of a query reads its first listed block
  int32_t ngroups;
  int32_t *out_idx;
  float *out_dist;
  int64_t *out_isect;
  tknnNeigh *out_fb;
  int32_t *out_level;
  // continuation state for queries whose candidate lists outgrow the LDS lists (finished by the
  // lane kernel): per sorted slot, preset by the host to done=1
  uint8_t *done;
  int64_t *isect_sorted;
  int32_t *next_level;
  // per sorted slot: 1 + level (| 0x80: knn_flag_tie's `edge`) for rows that finished with bit-identical distances among entries
  // 0..k of the list (entry k: the best candidate left out); tie_fix_kernel redoes them in the reference's tie order
  uint8_t *tie;
  int32_t *tie_list;
  const uint8_t *skip;  // per sorted slot, or null: queries with skip[slot] == skip_is sit this solve out (tknnSolveOptions.phase)
  int32_t skip_is;
  const int32_t *slot_count;  // tie_fix_kernel, nslots == -2: length of `slots` as hipCUB's select wrote it (device side)
  // a wave takes 4 * grab consecutive slots per turn at the work cursor (>= 1).  One address, device scope: a turn costs some
  // 12 ns whoever asks -- at four rows a turn that was ALL of the tie pass on duplicate-heavy sets (10 M taxi-like points, k = 10 / 24:
  // 0.78 / 2.3 M rows in 2.5 / 7.2 ms; with longer turns 1.1 / 2.8 ms).  The walks keep 1: their queries differ too much in cost
  // (turns of up to 64 slots: the hand-over walk 4.0 -> 4.5 ms on that set, the k = 65 walk 173 -> 189 ms on 10 M uniform points).
  int grab;
  const int32_t *row_slot;    // tie_fix_kernel: the sorted slot of row i (Lbvh::row_slot_device), or null: no look at the written row first
  // [0] (unused here) [kXcdCounter + 32 x] per-XCD packet counters [1] max levels [2] node tests [3] point tests [4] sum isect
  // [5] error flags (1 max_rounds) [6] sum levels [7] unfinished [8] handed over [9] min hand-over level
  unsigned long long *counters;
};

__device__ __forceinline__ void t_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float t_bcast(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}
__device__ __forceinline__ float t_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ unsigned long long t_wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ int t_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// value of lane `src` (any lane of the wave, may differ per lane) -- LDS crossbar, no LDS memory
__device__ __forceinline__ uint32_t t_lane_read(uint32_t v, int src) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
}
// sum over the 16 lanes of my team (row), result in every lane of the team
__device__ __forceinline__ uint32_t t_team_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /*row_ror:8*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124 /*row_ror:4*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122 /*row_ror:2*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121 /*row_ror:1*/, 0xf, 0xf, false);
  return v;
}
// min / max over the 16 lanes of my team, result in every lane of the team
__device__ __forceinline__ float t_team_min(float v) {
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /*row_ror:8*/, 0xf, 0xf, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /*row_ror:4*/, 0xf, 0xf, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122 /*row_ror:2*/, 0xf, 0xf, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121 /*row_ror:1*/, 0xf, 0xf, false)));
  return v;
}
__device__ __forceinline__ float t_team_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /*row_ror:8*/, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /*row_ror:4*/, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122 /*row_ror:2*/, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121 /*row_ror:1*/, 0xf, 0xf, false)));
  return v;
}
// value of another lane of my row through DPP (quad permutes, mirrors) -- no LDS, no address register
template <int CTRL>
__device__ __forceinline__ uint32_t t_dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// value of lane (mine ^ 4): lanes 0-3 and 8-11 of a row read four lanes up, the others four lanes down -- two
// DPP moves with complementary bank masks (a ds_swizzle does it in one instruction, but through the LDS
// crossbar: 24 cycles of the LDS pipe and its latency, scripts/microbench/issue_rate.hip)
__device__ __forceinline__ uint32_t t_xor4(uint32_t v) {
  const int up = __builtin_amdgcn_update_dpp(0, (int)v, 0x104 /*row_shl:4*/, 0xf, 0x5, false);
  return (uint32_t)__builtin_amdgcn_update_dpp(up, (int)v, 0x114 /*row_shr:4*/, 0xf, 0xa, false);
}

// my left neighbour's value inside the team (lane 0 of a team gets 0: bound_ctrl)
__device__ __forceinline__ uint32_t t_team_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}

// acc + (my bit of the wave mask) as ONE add-with-carry on the compare mask (the compiler's select + add is two)
__device__ __forceinline__ uint32_t t_count(uint32_t acc, unsigned long long mask) {
  uint32_t out;
  asm("v_addc_co_u32_e64 %0, vcc, 0, %1, %2" : "=v"(out) : "v"(acc), "s"(mask) : "vcc");
  return out;
}

// the same, and `keep` stays live (in its register) up to here at no cost: the COUNT pass never reads a
// block's id word, and the allocator would reuse the fourth register of a load's destination tuple as a
// temporary while the load is in flight -- a write-after-write hazard the compiler covers with
// s_waitcnt vmcnt(0), which drains the whole ring
__device__ __forceinline__ uint32_t t_count_keep(uint32_t acc, unsigned long long mask, int32_t keep) {
  uint32_t out;
  asm("v_addc_co_u32_e64 %0, vcc, 0, %1, %2" : "=v"(out) : "v"(acc), "s"(mask), "v"(keep) : "vcc");
  return out;
}

// squared distance from the three differences: knn_dist2's expression ((x*x) + (y*y)) + (z*z),
// x and y squared in one packed instruction
typedef float t_point4 __attribute__((ext_vector_type(4)));
// knn_dist2's expression ((x*x) + (y*y)) + (z*z), every operation rounded on its own.  Plain instructions: a
// packed v_pk_mul_f32 issues in 6.3 cycles against 2.4 for each of the two multiplies it replaces
// (scripts/microbench/issue_rate.hip); the file is compiled with -fno-slp-vectorize for the same reason.
__device__ __forceinline__ float t_dist2(float dx, float dy, float dz) {
#pragma clang fp contract(off)
  return ((dx * dx) + (dy * dy)) + (dz * dz);
}

// one leaf block = LBVH_BLOCK sorted points, 16 bytes each; lane tl of a team reads point tl.
// The sorted arrays are padded with NaN sentinels to whole blocks and followed by one all-NaN block
// (lbvh.hip), so there is no bounds test and "no block" is an ordinary entry.
// Without a halo tree an entry is resolved to the block's BYTE offset when the entry registers are
// filled, and the load is "uniform base + 32-bit lane offset" (saddr form): one add per block
// instead of a mask, a 64-bit shift and a 64-bit add.  (Blocks < 2^24, i.e. n < 2^28: solve_team checks.)
template <bool HALO>
__device__ __forceinline__ LbvhPoint load_block_point(const LbvhPoint *own, const LbvhPoint *halo, int32_t entry,
                                                      const LbvhPoint *own_base, uint32_t lane_bytes) {
  if (!HALO) return *(const LbvhPoint *)((const char *)own_base + ((uint32_t)entry + lane_bytes));
  const LbvhPoint *base = entry < 0 ? halo : own;
  return base[(int64_t)(entry & 0x7fffffff) * LBVH_BLOCK];
}
template <bool HALO>
__device__ __forceinline__ int32_t resolve_entry(int32_t e) {
  return HALO ? e : (int32_t)((uint32_t)e * (uint32_t)(LBVH_BLOCK * sizeof(LbvhPoint)));
}

// ---- a team's sorted list and its candidate buffer ---------------------------------------------------------------------
// One compare-exchange with the lane whose key is (pd, pi): the lower lane keeps the smaller key.
__device__ __forceinline__ void t_exchange(uint32_t &kd, uint32_t &ki, uint32_t pd, uint32_t pi, bool upper) {
  const uint64_t mine_k = ((uint64_t)kd << 32) | ki, other = ((uint64_t)pd << 32) | pi;
  const bool take = (other < mine_k) != upper;  // lower lane: the smaller key; upper lane: the larger (equal: either)
  kd = take ? pd : kd;
  ki = take ? pi : ki;
}
// 16 keys of a team, one per lane, into ascending order: a bitonic network written so that every exchange
// keeps the smaller key in the lower lane -- mirror within 2, 4, 8, 16 lanes followed by xor 4 / 2 / 1
// steps; ten exchanges of DPP moves (quad permutes, mirrors, row shifts) and a 64-bit compare each.
__device__ __forceinline__ void t_sort16(uint32_t &kd, uint32_t &ki, int tl) {
  const bool up1 = (tl & 1) != 0, up2 = (tl & 2) != 0, up4 = (tl & 4) != 0, up8 = (tl & 8) != 0;
  t_exchange(kd, ki, t_dpp<0xb1>(kd), t_dpp<0xb1>(ki), up1);    // pairs
  t_exchange(kd, ki, t_dpp<0x1b>(kd), t_dpp<0x1b>(ki), up2);    // mirror within 4
  t_exchange(kd, ki, t_dpp<0xb1>(kd), t_dpp<0xb1>(ki), up1);
  t_exchange(kd, ki, t_dpp<0x141>(kd), t_dpp<0x141>(ki), up4);  // mirror within 8 (row_half_mirror)
  t_exchange(kd, ki, t_dpp<0x4e>(kd), t_dpp<0x4e>(ki), up2);    // xor 2
  t_exchange(kd, ki, t_dpp<0xb1>(kd), t_dpp<0xb1>(ki), up1);
  t_exchange(kd, ki, t_dpp<0x140>(kd), t_dpp<0x140>(ki), up8);  // mirror within 16 (row_mirror)
  t_exchange(kd, ki, t_xor4(kd), t_xor4(ki), up4);
  t_exchange(kd, ki, t_dpp<0x4e>(kd), t_dpp<0x4e>(ki), up2);
  t_exchange(kd, ki, t_dpp<0xb1>(kd), t_dpp<0xb1>(ki), up1);
}
// four half-cleaners: a bitonic sequence of sixteen keys, one per lane of the team, into ascending order
__device__ __forceinline__ void t_clean16(uint32_t &kd, uint32_t &ki, int tl) {
  const bool up1 = (tl & 1) != 0, up2 = (tl & 2) != 0, up4 = (tl & 4) != 0, up8 = (tl & 8) != 0;
  t_exchange(kd, ki, t_dpp<0x128>(kd), t_dpp<0x128>(ki), up8);  // xor 8 (row_ror:8)
  t_exchange(kd, ki, t_xor4(kd), t_xor4(ki), up4);
  t_exchange(kd, ki, t_dpp<0x4e>(kd), t_dpp<0x4e>(ki), up2);
  t_exchange(kd, ki, t_dpp<0xb1>(kd), t_dpp<0xb1>(ki), up1);
}
// ---- the same network on ONE 32-bit word per lane (round 4) ------------------------------------------------------------
// A 64-bit exchange is two DPP moves, a 64-bit compare, a mask xor and two selects -- six slow instructions and their wait
// states, ten times per sort: the sorting networks were a fifth of the packet kernel's vector time.  On one word the
// exchange is a DPP move and ONE v_med3_u32: med3(a, b, 0) = min(a, b) for the lower lane of a pair, med3(a, b, ~0) =
// max(a, b) for the upper one (the third operand is a constant of the lane).  The word is the key's order WITHOUT its last
// four bits, which carry the lane the key came from; the caller fetches the exact key from there afterwards and checks that
// the dropped bits could not have mattered (t_sorted_row).
__device__ __forceinline__ uint32_t t_med3_u32(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t out;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(out) : "v"(a), "v"(b), "v"(c));
  return out;
}
// partner lane ^ 4, lower lane keeps the smaller word: the two halves as one masked min and one masked max (bank = four
// consecutive lanes of a row; row_shl:4 reads four lanes up, row_shr:4 four lanes down).  s_nop: a DPP operand written by the
// instruction before it needs two wait states, and the compiler does not look into inline assembly for that.
__device__ __forceinline__ uint32_t t_exchange_xor4_u32(uint32_t v) {
  uint32_t out;
  asm("s_nop 1\n\t"
      "v_min_u32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_max_u32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(out)
      : "v"(v));
  return out;
}
__device__ __forceinline__ void t_sort16_u32(uint32_t &v, int tl) {
  const uint32_t c1 = 0u - ((uint32_t)tl & 1u), c2 = 0u - (((uint32_t)tl >> 1) & 1u), c4 = 0u - (((uint32_t)tl >> 2) & 1u),
                 c8 = 0u - (((uint32_t)tl >> 3) & 1u);
  v = t_med3_u32(v, t_dpp<0xb1>(v), c1);   // pairs
  v = t_med3_u32(v, t_dpp<0x1b>(v), c2);   // mirror within 4
  v = t_med3_u32(v, t_dpp<0xb1>(v), c1);
  v = t_med3_u32(v, t_dpp<0x141>(v), c4);  // mirror within 8
  v = t_med3_u32(v, t_dpp<0x4e>(v), c2);   // xor 2
  v = t_med3_u32(v, t_dpp<0xb1>(v), c1);
  v = t_med3_u32(v, t_dpp<0x140>(v), c8);  // mirror within 16
  v = t_exchange_xor4_u32(v);
  v = t_med3_u32(v, t_dpp<0x4e>(v), c2);
  v = t_med3_u32(v, t_dpp<0xb1>(v), c1);
}
// Sixteen (squared distance, index) keys, one per lane of a team (`have`: my lane has one), as the sorted row of
// (IEEE distance, index) keys the lists take -- KNN_EMPTY_KEY past the last.  `fetch(lane)` returns the key lane `lane` of
// my team came with.  The sort runs on (bits of d2 without their last four | lane); it is the order of the full keys unless
// two neighbours of the outcome are closer than that can tell -- their truncated words equal or one apart, which covers
// d2 values less than sixteen ulps apart: exact duplicates, the ties of quantised data, and the pairs of different squares
// whose ROUNDED roots coincide (the pre-image of one rounded root spans three floats), for which the index decides -- or a
// square overflowed to infinity (its root is beyond the empty key, which the word order does not know).  Then the exact
// network runs instead (uniform 10 M points: once in 10^4 rows).
template <typename Fetch>
__device__ __forceinline__ void t_sorted_row(uint32_t d2_bits, uint32_t id, bool have, int tl, Fetch fetch, uint32_t &kd, uint32_t &ki) {
  uint32_t w = have ? ((d2_bits & ~15u) | (uint32_t)tl) : (0xfffffff0u | (uint32_t)tl);
  t_sort16_u32(w, tl);
  const uint32_t before = t_team_shr1(w);
  const bool real = w < 0xfffffff0u;
  const bool unsure = real & (((tl > 0) & ((w >> 4) - (before >> 4) <= 1u)) | (w >= 0x7f7ffff0u));
  if (__builtin_expect(__ballot(unsure) == 0ull, 1)) {
    const unsigned long long key = fetch((int)(w & 15u));
    const float dist = knn_sqrt(__uint_as_float((uint32_t)(key >> 32)));
    kd = real ? __float_as_uint(dist) : 0x7f7fffffu;
    ki = real ? (uint32_t)key : 0u;
  } else {
    const float dist = knn_sqrt(__uint_as_float(d2_bits));
    kd = have ? __float_as_uint(dist) : 0x7f7fffffu;
    ki = have ? id : 0u;
    t_sort16(kd, ki, tl);
  }
}

// Candidates that pass a team's gate are not inserted one lock-step round each: they wait in the team's LDS buffer as
// (squared distance, index) and are MERGED into the sorted list sixteen at a time.  `buf`: my team's buffer, `fill`: how
// many wait (the same in the team's lanes).  Per row of sixteen: the IEEE root (sixteen instructions, once per row and not
// per block step), the sorting network, then the row meets the list one register (sixteen sorted entries, all of them
// below the next register's) at a time: mirrored, lane j against the row's key 15 - j, the sixteen smallest of both stay
// in the register and the sixteen largest travel on as the row for the next register -- both come out as bitonic
// sequences, four half-cleaners each.  A register no key of the row gets into is left as it is.  What comes out of the
// last register has fallen out of the list (`left_out`, tracked if `full`: its smallest distance, team minimum taken by
// the caller).  Some 80 vector instructions per row for k <= 16, 250 for k <= 64, whatever the number of candidates.
template <int NREG>
__device__ __forceinline__ void t_merge_rows(uint32_t (&bd)[NREG], uint32_t (&bi)[NREG], uint32_t &left_out, bool full,
                                             const unsigned long long *buf, uint32_t fill, int tl) {
#pragma unroll
  for (int row = 0; row < kCandCapacity / 16; row++) {
    if (row > 0 && __ballot(fill > 16u * (uint32_t)row) == 0ull) break;
    const uint32_t at = 16u * (uint32_t)row + (uint32_t)tl;
    const bool have = at < fill;
    const unsigned long long key = have ? buf[at] : 0ull;
    uint32_t kd, ki;  // the row, sorted: (IEEE distance, index), KNN_EMPTY_KEY past the end
    t_sorted_row((uint32_t)(key >> 32), (uint32_t)key, have, tl, [&](int from) { return buf[16 * row + from]; }, kd, ki);
#pragma unroll
    for (int j = 0; j < NREG; j++) {
      const uint32_t od = t_dpp<0x140>(kd), oi = t_dpp<0x140>(ki);  // the row's key 15 - tl
      const uint64_t mine_k = ((uint64_t)bd[j] << 32) | bi[j], other = ((uint64_t)od << 32) | oi;
      const bool take = other < mine_k;
      if (NREG > 1 && __ballot(take) == 0ull) {  // every key of the row is larger than this whole register: on to the next
        if (full && j == NREG - 1) left_out = min(left_out, kd);
        continue;
      }
      const uint32_t hd = take ? bd[j] : od, hi_i = take ? bi[j] : oi;  // the larger of the pair: travels on (or falls out)
      bd[j] = take ? od : bd[j];
      bi[j] = take ? oi : bi[j];
      t_clean16(bd[j], bi[j], tl);
      if (j == NREG - 1) {
        if (full) left_out = min(left_out, hd);
      } else {
        kd = hd, ki = hi_i;
        t_clean16(kd, ki, tl);
      }
    }
  }
}

struct TeamLds {
  float *qrec;        // [64][kQrecStride]
  int32_t *blk;       // [kMaxBlocks] block entries of the packet (bit 31: halo tree)
  uint8_t *qblk;      // [64][kMaxPerQuery] per-query slots into blk[]
  uint32_t *qhi;      // [64][kHiWords] (k > 32) bit p of a query's words: the ninth bit of the slot at position p of its list
  uint32_t *qcnt;     // [64][2] out: candidates in the inner box, in the outer box | self << 31
  int32_t *qlist;     // compact list of the queries this pass serves
  int32_t *ent;       // [4][kEntStride] per team: resolved block entries of its query, in visit order
  unsigned long long *cand;  // [4][kCandCap] per team: keys waiting to be merged into its list
};

// Can two candidates of one query at the same fp32 distance d have become candidates in DIFFERENT
// rounds?  A candidate's Chebyshev distance t obeys d / sqrt(3) <= t <= d (sqrt(2) for points in a
// plane z = const, as the reference's 2-D inputs are: `span`), and round l takes it iff
// t <= r_l (up to the rounding margin M of the box test, see team_kernel).  Going up the radii: if
// d is safely below r_l, every candidate at distance d passes round l -- and none passed an earlier
// round, or the loop would have stopped there; if not, but d / sqrt(3) can be below r_l, some may
// pass and others not.  Exact duplicates (d = 0) and the other ties of quantised data mostly are of
// the first kind and need no second look.  qmax = max |q|, r_last = the radius the query finished with.
__device__ __forceinline__ bool tie_may_straddle(float d, float r0, float r_last, float qmax, float span) {
  for (float r = r0;; r = r * 2.0f) {
    const float mg = (qmax + 2.0f * r) * 4.76837158203125e-07f;  // 2^-21
    if (d <= (r - mg) * 0.99999f) return false;
    if (d <= (r + mg) * span) return true;  // t >= d / span > r + M otherwise: certainly not a candidate of round l
    if (!(r < r_last)) return true;  // (a listed candidate passes the last round's test: not reached)
  }
}

// One pass of the four teams over a compact list of queries.  SELECT = false: count candidates
// (deviceCode.cu:74,103).  SELECT = true: also keep the k best (dist,index) keys, lane j of the
// team holding the j-th, and write the row if the query turns out finished (>= k others).
// NREG: list registers per lane -- the team's sorted list holds 16 * NREG keys (k <= 16: 1, k <= 32: 2, k <= 64: 4)
// TWO: a COUNT pass that also counts the inner box of a two-level step (a template flag: as a run-time value the
// compiler kept "m > 1" as a lane mask and re-derived a branch from it for every block)
template <bool SELECT, bool HALO, int NREG, bool FULL, bool TWO>
__device__ __forceinline__ void team_pass(const TeamArgs &a, const TeamLds &L, int n_list, int32_t first_slot,
                                          const LbvhPoint *own_pts, const LbvhPoint *halo_pts, int lane) {
  const int team = lane >> 4, tl = lane & 15;
  constexpr int kMaxPerQuery = TeamLayout<NREG>::kMaxPerQuery, kEntStride = TeamLayout<NREG>::kEntStride;
  static_assert(!(SELECT && TWO), "only a COUNT pass serves two levels");
  n_list = __builtin_amdgcn_readfirstlane(n_list);  // wave-uniform by construction; says so to the compiler (scalar branches on it)
  [[maybe_unused]] unsigned long long tp[4] = {0, 0, 0, 0};  // TKNN_DIAG_BUILD, flag 128: setup | first group | other groups | epilogue
  for (int r0 = 0; r0 < n_list; r0 += 4) {
    [[maybe_unused]] unsigned long long tp_mark = 0;
    if (TKNN_DIAG_BUILD && (a.diag & 128)) tp_mark = __builtin_amdgcn_s_memtime();
#define TP_LAP(i)                                                     \
  do {                                                                \
    if (TKNN_DIAG_BUILD && (a.diag & 128)) {                          \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
      tp[i] += now_ - tp_mark;                                        \
      tp_mark = now_;                                                 \
    }                                                                 \
  } while (0)
    const bool on = r0 + team < n_list;
    int qi = 0;
    if (on) qi = L.qlist[r0 + team];
    const float *rec = L.qrec + qi * kQrecStride;
    const float t_qx = rec[0], t_qy = rec[1], t_qz = rec[2];
    const int32_t t_qid = __float_as_int(rec[3]);
    const float t_r = rec[4];
    const float t_mg = (fmaxf(fmaxf(fabsf(t_qx), fabsf(t_qy)), fabsf(t_qz)) + 2.0f * t_r) * 4.76837158203125e-07f;  // 2^-21, see above
    const float in_below = t_r - t_mg, in_upto = t_r + t_mg;          // certainly / possibly a candidate
    // the inner level of a two-level COUNT step: half my outer radius (exact: radii double in fp32).  Not a pass
    // argument: with a per-query schedule every query has radii of its own.
    const float r_inner = t_r * 0.5f;
    const float i0_below = r_inner - t_mg, i0_upto = r_inner + t_mg;
    uint32_t cnt_i0 = 0;  // inner level of a two-level COUNT step (m = levels served by this gather, wave-uniform)
    const int packed = __float_as_int(rec[5]);
    // the query's output row, read at set-up: a load in flight over the whole round instead of a round trip to memory
    // between the last block and the row's stores (-1.4 % on the benchmark)
    [[maybe_unused]] int32_t out_row = 0;
    if (SELECT) out_row = a.bvh.prim_id[first_slot + qi];
    const int my_n = on ? (packed & 0xff) : 0;  // leaf blocks of my team's query
    // wave-uniform trip count: longest list of the 4 teams -- from v_readlane values only, so that the
    // compiler keeps it (and the loop tests below) in scalar registers
    const int steps = max(max((int)__builtin_amdgcn_readlane(my_n, 0), (int)__builtin_amdgcn_readlane(my_n, 16)),
                          max((int)__builtin_amdgcn_readlane(my_n, 32), (int)__builtin_amdgcn_readlane(my_n, 48)));
    if (TKNN_DIAG_BUILD && (a.diag & 16) && lane == 0) atomicAdd(&a.counters[15], (unsigned long long)steps);
    // my query's block entries, spread over the team's lanes: lane tl holds entries tl, tl+16, ...
    // (clamped to the last one; teams without a query read block 0 of the own tree and ignore it)
    const uint8_t *mine = L.qblk + qi * kMaxPerQuery;
    const int last = my_n - 1;
    // positions past the end of my list (and teams without a query) name the all-NaN block after
    // the own tree's last block: its points fail every test, so the loop needs no "am I still in
    // my list" check
    const int32_t nan_block = a.wide[0].count[0];
    // SELECT visits blocks outward from the query's own block, alternating sides of its Morton-ordered
    // list: near blocks first tightens the k-th-distance gate early (about 30 % fewer inserts than
    // list order on uniform data) and changes nothing else -- the result does not depend on order.
    // COUNT takes the list as it is.
    // A list that serves the level AFTER the step it was gathered for (see team_kernel) is "the step's blocks in Morton
    // order, then the rest": outward order inside the first part (bits 16..23 of the record; 0: the whole list), the
    // rest as listed -- farther than every block of the first part anyway.
    const int own_pos = (packed >> 8) & 0xff;
    const int n_near = (packed >> 16) & 0xff;
    const int last_near = n_near ? n_near - 1 : last;
    const int both = min(own_pos, last_near - own_pos);
    const bool right_longer = last_near - own_pos > own_pos;
    auto list_pos = [&](int it) -> int {
      if (!SELECT) return it;
      const int half = (it + 1) >> 1;
      const int alt = (it & 1) ? half : -half;  // it = 0: own block
      const int far = it - both;
      const int off = it <= 2 * both ? alt : (right_longer ? far : -far);
      return it > last_near ? it : own_pos + off;
    };
    auto list_entry = [&](int pos) -> int32_t {
      int lp = list_pos(pos);
      if (n_near && lp >= n_near) lp = kMaxPerQuery - 1 - (lp - n_near);  // the back part, filled downwards
      const int at = min(max(lp, 0), kMaxPerQuery - 1);
      int slot_at = mine[at];
      if (TeamLayout<NREG>::kHiWords) slot_at |= (int)((L.qhi[qi * TeamLayout<NREG>::kHiWords + (at >> 5)] >> (at & 31)) & 1u) << 8;
      int32_t e = L.blk[slot_at];
      if (TKNN_DIAG_BUILD && (a.diag & 64)) e = L.blk[mine[0]];  // every visit reads one and the same block: what do cache misses cost?
      return resolve_entry<HALO>(pos <= last ? e : nan_block);
    };
    // The team's entries go to LDS once, resolved (byte offsets) and in visit order, lane tl writing
    // positions tl, tl + 16, ...; the loop below then fetches four of them with ONE 16-byte LDS read per
    // team (same address in the team's 16 lanes) -- no cross-lane read, no select, no address
    // arithmetic per block.  Positions up to the longest list rounded up to whole groups of four, plus
    // one group of prefetch, name the all-NaN block.
    const int steps4 = (steps + 3) & ~3;
    int32_t *my_ent = L.ent + team * kEntStride;
    for (int base = 0; base < steps4 + 4; base += 16)
      if (base + tl < kEntStride) my_ent[base + tl] = list_entry(base + tl);
    t_wave_sync();
    uint32_t cnt = 0;
    uint32_t bd[NREG], bi[NREG];  // register j of lane t holds list entry 16 j + t (indices are compile-time: stays in VGPRs)
#pragma unroll
    for (int j = 0; j < NREG; j++) {
      bd[j] = 0x7f7fffffu;  // KNN_EMPTY_KEY = {FLT_MAX, 0}
      bi[j] = 0u;
    }
    // k == list size: the smallest distance (bits) among the keys that found no room, tracked in the lane
    // of the last entry -- equal to the k-th distance iff a candidate tied with the row's last stayed out
    // (FULL is a template parameter: as a run-time flag the three instructions it adds to an insert
    // were if-converted into every k's loop, +4 % on the benchmark)
    constexpr bool full = FULL;
    uint32_t left_out = 0xffffffffu;
    // the k-th best of my team, whose distance gates further candidates
    auto kth_dist = [&]() -> float {
      uint32_t reg = bd[0];
#pragma unroll
      for (int j = 1; j < NREG; j++) reg = ((a.k - 1) >> 4) == j ? bd[j] : reg;
      return __uint_as_float(t_lane_read(reg, (team << 4) + ((a.k - 1) & 15)));
    };
    float tau2 = INFINITY;
    // Candidates that pass the gate wait in the team's LDS buffer and are merged into the sorted list sixteen at a time
    // (t_merge_rows).  Between merges the gate is the last merge's k-th distance: looser than it could be, never wrong.
    bool dirty = false;  // wave-uniform: some team has buffered candidates
    // how many wait in my team's buffer (the same in its sixteen lanes).  In a register, advanced by population counts of the
    // candidate mask: the slot of a candidate used to come from an LDS atomic -- a round trip through the LDS pipe in the
    // middle of every block step that had a candidate, in a kernel that is bound by the latency of such chains
    uint32_t fill_n = 0;
    auto merge_buffer = [&]() {
      t_wave_sync();
      t_merge_rows<NREG>(bd, bi, left_out, full, L.cand + team * kCandCap, fill_n, tl);
      t_wave_sync();  // (the rows are read: the next candidates may overwrite them)
      fill_n = 0;
      dirty = false;
      tau2 = knn_gate_from_worst(kth_dist());
    };
    // one block's test.  The candidate test in two tiers (see the record layout above): outside the
    // band |t - r| <= M the comparison t <= r IS the literal test; inside it (about 1e-6 of the points,
    // every point of a face-aligned lattice) the literal test runs.  The band check is one subtraction,
    // one compare and a branch that is almost never taken -- lane predicates cost a compare (twice a
    // plain VALU instruction on gfx950) and every wave-mask operation a slot of the CU's one scalar
    // pipe, which this kernel fills as much as the vector pipes (scripts/microbench/issue_rate.hip).
    auto in_box = [&](const LbvhPoint &p, float t, float radius, float below, float upto) -> unsigned long long {
      // certainly in | possibly in: two compares against the band's edges (no subtraction); they differ for a point in a
      // thousand blocks, and only then does the literal test run
      unsigned long long in_m = __ballot(t <= below);
      const unsigned long long maybe = __ballot(t <= upto);
      if (__builtin_expect(maybe != in_m, 0)) in_m |= maybe & __ballot(knn_in_box(p.x, p.y, p.z, radius, t_qx, t_qy, t_qz));
      return in_m;
    };
    auto process = [&](const LbvhPoint &p, bool own_block = false) {
      const float dx = p.x - t_qx, dy = p.y - t_qy, dz = p.z - t_qz;
      // NaN only if all three are (sentinels; lbvh.hip turns a point with any NaN coordinate into one)
      const float t = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
      const unsigned long long in_m = in_box(p, t, t_r, in_below, in_upto);
      cnt = t_count_keep(cnt, in_m, p.id);
      if (TWO) cnt_i0 = t_count(cnt_i0, in_box(p, t, r_inner, i0_below, i0_upto));
      if (SELECT) {
        const float d2 = t_dist2(dx, dy, dz);
        // candidates that pass the gate.  The query itself sits in its own block, which SELECT visits
        // first and settles in the sorted first step, so no block seen here can hold it (ids are unique).
        unsigned long long pm = in_m & __ballot(d2 <= tau2);
        if (own_block) pm &= __ballot(p.id != t_qid);  // deviceCode.cu:103: a query is no neighbour of itself (ids are unique: only its own block holds it)
        if (TKNN_DIAG_BUILD && (a.diag & 1)) pm = 0;
        if (pm) {
          if (TKNN_DIAG_BUILD && (a.diag & 16) && lane == 0) atomicAdd(&a.counters[26], (unsigned long long)__popcll(pm));
          // my candidate into the team's buffer; a full row of sixteen is merged at once (a block
          // adds at most sixteen to at most fifteen: the buffer holds 32)
          const uint32_t mine16 = (uint32_t)(pm >> (team << 4)) & 0xffffu;  // my team's lanes with a candidate
          if ((mine16 >> tl) & 1u) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (uint32_t)p.id;  // (the root: merge_buffer)
            L.cand[team * kCandCap + fill_n + __popc(mine16 & ((1u << tl) - 1u))] = key;
          }
          fill_n += __popc(mine16);
          dirty = true;
          if (__ballot(fill_n >= 16u) != 0ull) {
            if (TKNN_DIAG_BUILD && (a.diag & 16) && lane == 0) atomicAdd(&a.counters[27], 1ull);
            merge_buffer();
          }
        }
      }
    };
    // A ring of four block buffers: while one block is tested the next three are in flight, and the
    // NEXT group's four entries are read from LDS a whole group ahead.  Loads are unconditional and
    // consumed in strict rotation (the compiler's s_waitcnt vmcnt(N) then counts exactly); whole groups
    // only: positions past a list's end are the all-NaN block, whose points fail every test.
    const LbvhPoint *own_base = a.bvh.points;  // wave-uniform
    const uint32_t lane_bytes = (uint32_t)tl * (uint32_t)sizeof(LbvhPoint);
    typedef int32_t t_int4 __attribute__((ext_vector_type(4)));
    const t_int4 *my_groups = (const t_int4 *)my_ent;
    // A block in flight is ONE 128-bit value (x, y, z, id bits): carried through the loop as a vector, it stays
    // in the register tuple the load writes.  As four scalars the compiler narrows the COUNT pass's loads
    // to three words and copies them into other registers at the loop's back edge -- which waits for
    // every load in flight.
    auto fetch = [&](int32_t entry) -> t_point4 {
      const LbvhPoint p = load_block_point<HALO>(own_pts, halo_pts, entry, own_base, lane_bytes);
      return t_point4{p.x, p.y, p.z, __int_as_float(p.id)};
    };
    auto unpack = [](const t_point4 &v) -> LbvhPoint { return LbvhPoint{v.x, v.y, v.z, __float_as_int(v.w)}; };
    auto sort_first = [&](const LbvhPoint &p) {
      // The first block (the query's own) meets an empty list: every candidate in it would be inserted,
      // one lock-step round each.  Instead the team SORTS its 16 keys (non-candidates = the empty key)
      // with a bitonic network written so that every exchange keeps the smaller key in the lower lane:
      // mirror within 2, 4, 8, 16 lanes followed by xor 4 / 2 / 1 steps -- ten exchanges of two
      // cross-lane moves (DPP quad permutes and mirrors, one ds_swizzle) and a 64-bit compare each.
      const float dx = p.x - t_qx, dy = p.y - t_qy, dz = p.z - t_qz;
      const float t = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
      bool in = t <= t_r;
      if (__ballot(fabsf(t - t_r) <= t_mg) != 0ull) in = (t <= in_below) || ((t <= in_upto) && knn_in_box(p.x, p.y, p.z, t_r, t_qx, t_qy, t_qz));
      cnt = t_count(cnt, __ballot(in));
      bool cand = in && (p.id != t_qid);
      if (TKNN_DIAG_BUILD && (a.diag & 1)) cand = false;
      // (the squares go through the team's candidate buffer, empty at this point, so that the sorted lanes can fetch them)
      const uint32_t d2_bits = __float_as_uint(t_dist2(dx, dy, dz));
      unsigned long long *my_buf = L.cand + team * kCandCap;
      my_buf[tl] = ((unsigned long long)d2_bits << 32) | (uint32_t)p.id;
      t_wave_sync();
      t_sorted_row(d2_bits, (uint32_t)p.id, cand, tl, [&](int from) { return my_buf[from]; }, bd[0], bi[0]);
      t_wave_sync();  // (read: the first candidates may overwrite them)
      tau2 = knn_gate_from_worst(kth_dist());  // k > 16: the second register is still empty, the gate stays open
    };
    TP_LAP(0);
    t_point4 b0, b1, b2, b3;
    {
      // issued in ring order (the scheduler would reorder four independent loads, and the loop's
      // s_waitcnt vmcnt(N) is the minimum over the orders it can be entered with)
      const t_int4 g = my_groups[0];
      b0 = fetch(g.x);
      __builtin_amdgcn_sched_barrier(0);
      b1 = fetch(g.y);
      __builtin_amdgcn_sched_barrier(0);
      b2 = fetch(g.z);
      __builtin_amdgcn_sched_barrier(0);
      b3 = fetch(g.w);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int it = 0; it < steps; it += 4) {
      const t_int4 g = my_groups[(it >> 2) + 1];  // entries it+4 .. it+7
      if (SELECT && it == 0) {
        // the query's own block meets an empty list.  k <= 16: its candidates simply are the first into the team's
        // buffer, sorted by the first merge; larger k: sorted on the spot (sort_first), the other blocks insert one by one
        if (NREG == 1 && TKNN_SORT_FIRST == 0)
          process(unpack(b0), true);
        else
          sort_first(unpack(b0));
      } else {
        process(unpack(b0));
      }
      b0 = fetch(g.x);
      process(unpack(b1));
      b1 = fetch(g.y);
      process(unpack(b2));
      b2 = fetch(g.z);
      process(unpack(b3));
      b3 = fetch(g.w);
      // a tighter gate for the next group as soon as a handful of candidates wait (the first groups, whose
      // blocks lie next to the query, bring most of them)
      if (SELECT && dirty && __ballot(fill_n >= (uint32_t)TKNN_MERGE_AT) != 0ull) {
        if (TKNN_DIAG_BUILD && (a.diag & 16) && lane == 0) atomicAdd(&a.counters[27], 1ull);
        merge_buffer();
      }
      TP_LAP(it == 0 ? 1 : 2);
    }
    if (SELECT && dirty) {
      if (TKNN_DIAG_BUILD && (a.diag & 16) && lane == 0) atomicAdd(&a.counters[27], 1ull);
      merge_buffer();
    }
    if (SELECT && full) {
      // the smallest key left out by any merge, where the tie test below looks for it: lane 15
      uint32_t v = left_out;
      v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x128 /*row_ror:8*/, 0xf, 0xf, false));
      v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x124 /*row_ror:4*/, 0xf, 0xf, false));
      v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x122 /*row_ror:2*/, 0xf, 0xf, false));
      v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x121 /*row_ror:1*/, 0xf, 0xf, false));
      left_out = v;
    }
    cnt = t_team_sum(cnt);
    const uint32_t self = cnt ? 1u : 0u;  // a query lies in its own box, and ids are unique
    const uint32_t others = cnt - self;
    if (TWO) cnt_i0 = t_team_sum(cnt_i0);
    uint32_t tied = 0u;
    if (SELECT) {
      // entry j against entry j - 1, for j = 1..k (KList::has_ties is the per-lane form of this)
      bool tie[NREG];
      bool any = false;
#pragma unroll
      for (int j = 0; j < NREG; j++) {
        uint32_t before = t_team_shr1(bd[j]);
        if (j > 0) before |= t_dpp<0x121>(bd[j - 1]) & (tl == 0 ? 0xffffffffu : 0u);
        tie[j] = ((j > 0) | (tl >= 1)) & (16 * j + tl <= a.k) & (bd[j] == before);
        if (full && j == NREG - 1) tie[j] |= (tl == 15) & (left_out == bd[j]);
        any |= tie[j];
      }
      if (__ballot(any) != 0ull) {  // rare: does any of them span two rounds?
        const float qmax = fmaxf(fmaxf(fabsf(t_qx), fabsf(t_qy)), fabsf(t_qz));
        any = false;
        bool edge = false;
        const float q_r0 = a.start_radii ? a.start_radii[a.bvh.prim_id[first_slot + qi]] : a.start_radius;
#pragma unroll
        for (int j = 0; j < NREG; j++) {
          const bool t = tie[j] && tie_may_straddle(__uint_as_float(bd[j]), q_r0, t_r, qmax, a.tie_span);
          any |= t;
          // ... with a candidate that is not written: entry k of a list with room to spare, the best one left out of a full list
          edge |= t && (16 * j + tl == a.k || (full && j == NREG - 1 && tl == 15 && left_out == bd[j]));
        }
        // (said to knn_flag_tie through the flag byte itself, here in the rare branch: a second bit through the passes' count
        // words and the level loop moved the packet kernel's register allocation -- 0.6 % of the benchmark)
        if (on && edge) knn_note_tie_edge(a.tie, first_slot + qi);
      }
      tied = ((uint32_t)(__ballot(any) >> (team * 16)) & 0xffffu) ? 1u : 0u;
    }
    if (on && tl == 0) {
      // counts innermost level first; a one-level pass fills slot 0
      uint32_t *out = L.qcnt + qi * 2;
      if (!TWO) {
        out[0] = cnt;
        out[1] = (self << 31) | (tied << 30);
      } else {
        out[0] = cnt_i0;
        out[1] = cnt | (self << 31);
      }
    }
    if (SELECT) {
      // finished at this level (deviceCode.cu:118: k insertions happened): lane j < k stores neighbour j.
      // The query's own lane adds the intersection count and the level afterwards (team_kernel).
      if (on && others >= (uint32_t)a.k) {
#pragma unroll
        for (int reg = 0; reg < NREG; reg++) {
          const int j = tl + 16 * reg;  // my entry of this register
          if (j >= a.k) continue;
          const int64_t o = (int64_t)out_row * a.k + j;
          const int32_t prim = knn_key_prim(((uint64_t)bd[reg] << 32) | bi[reg]);
          const float d = __uint_as_float(bd[reg]);
          if (a.out_idx) a.out_idx[o] = prim;
          if (a.out_dist) a.out_dist[o] = d;
          if (a.out_fb) {
            // slot 0 of the row: everything but `intersections`, which only the query's lane writes
            int2 *rec8 = (int2 *)(a.out_fb + o);
            rec8[0] = make_int2(prim, __float_as_int(d));
            rec8[1] = make_int2(j == 0 ? 0 : a.k, 0);
            if (j != 0) rec8[2] = make_int2(0, 0);
          }
        }
      }
    }
    TP_LAP(3);
  }
#undef TP_LAP
  if (TKNN_DIAG_BUILD && (a.diag & 128) && lane == 0)
    for (int i = 0; i < 4; i++) atomicAdd(&a.counters[(SELECT ? 36 : 28) + i], tp[i]);  // (32..34: the tie pass)
}

template <bool HALO, int NREG, bool FULL>
__global__ void __launch_bounds__(kTeamBlock) __attribute__((amdgpu_waves_per_eu(TKNN_TEAM_WAVES))) team_kernel(TeamArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;  // 0 with one wave per workgroup
  const int tl = lane & 15;
  using Lay = TeamLayout<NREG>;
  constexpr int kMaxPerQuery = Lay::kMaxPerQuery;
  unsigned char *base = smem + wid * Lay::kTeamLds;
  float *qrec = (float *)base;
  int32_t *blk = (int32_t *)(base + kLdsQrec);
  uint8_t *qblk = (uint8_t *)(base + Lay::kOffMask);  // [query][kMaxPerQuery] slots into blk[]
  uint32_t *qhi = (uint32_t *)(base + Lay::kOffHi);    // [query][kHiWords] their ninth bits (k > 32)
  constexpr int kMaxBlocks = Lay::kMaxBlocks;
  uint32_t *qcnt = (uint32_t *)(base + Lay::kOffShared);
  int32_t *qlist = (int32_t *)(base + Lay::kOffShared + kLdsCnt);
  int32_t *stack = (int32_t *)(base + Lay::kOffShared);  // shares the counts / query list region
  int32_t *ent = (int32_t *)(base + Lay::kOffEnt);
  unsigned long long *cand = (unsigned long long *)(base + Lay::kOffCand);

  const LbvhPoint *own_pts = a.bvh.points + tl, *halo_pts = a.halo.points ? a.halo.points + tl : a.bvh.points + tl;
  unsigned long long my_isect_sum = 0, wave_node_tests = 0, wave_point_tests = 0;
  uint32_t my_levels = 0, my_unfinished = 0;  // (per lane, a few dozen packets each: 32 bits, a register less apiece -- the kernel sits at its 128)
  int wave_levels = 0, wave_err = 0;
  uint32_t my_handed = 0;
  [[maybe_unused]] unsigned long long diag_scanned = 0, diag_listed = 0;  // TKNN_DIAG_BUILD only
  int wave_min_handover = 0x7fffffff;
#if TKNN_DIAG_BUILD
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, t_mark = __builtin_amdgcn_s_memtime();
#define PHASE_END(i)                                         \
  do {                                                       \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    ph[i] += now_ - t_mark;                                  \
    t_mark = now_;                                           \
  } while (0)
#else
#define PHASE_END(i) \
  do {               \
  } while (0)
#endif

  // Packets are Morton-consecutive, so neighbouring packets read the same leaf blocks.  Each XCD has
  // its own L2: packets are dealt to the XCDs in chunks of up to 1024 consecutive packets, a wave
  // pulls from the chunks of the XCD it runs on (HW_REG_XCC_ID) and steals from the others when
  // those are used up.  Placement changes speed only.  (Measured at C2: 15.0 ms with one global
  // counter, 14.4 ms with chunks of 1024-4096; whole eighths of the range are slower, 21.9 ms.)
  const int xcc = (int)__builtin_amdgcn_s_getreg(6164 /* hwreg(HW_REG_XCC_ID, 0, 4) */) & 7;
  const int chunk = max(1, min(1024, a.ngroups / 128));
  uint32_t seg_empty = 0;  // XCDs whose chunks are used up (wave-uniform)
  for (;;) {
    int g = -1;
    for (int t = 0; t < 8 && g < 0; t++) {
      const int x = (xcc + t) & 7;
      if (seg_empty & (1u << x)) continue;
      int v = 0;
      if (lane == 0) v = (int)atomicAdd(&a.counters[kXcdCounter + kXcdCounterStride * x], 1ull);
      v = __builtin_amdgcn_readfirstlane(v);
      const int cand = ((v / chunk) * 8 + x) * chunk + (v % chunk);  // v-th packet of XCD x
      if (cand < a.ngroups)
        g = cand;
      else
        seg_empty |= 1u << x;
    }
    if (g < 0 || wave_err) break;

    const int32_t slot = g * 64 + lane;
    bool active = slot < a.bvh.n;
    if (a.skip) {
      if (active && (int32_t)a.skip[slot] == a.skip_is) active = false;
      if (__ballot(active) == 0ull) continue;  // a packet without a query of this phase
    }
    LbvhPoint q = {0.f, 0.f, 0.f, -1};
    if (active) q = a.bvh.points[slot];
    const int32_t row = active ? a.bvh.prim_id[slot] : 0;
    float r = a.start_radius;
    if (a.start_radii && active) r = a.start_radii[row];  // a radius schedule of my own (levels stay common: level t = r0_q * 2^t)
    int level = 0;
    int64_t isect = 0;
    uint32_t prev_others = 0;  // others in my box at the previous level
    int step = a.first_step;   // levels the next gather serves
    TeamLds L;
    L.qrec = qrec;
    L.blk = blk;
    L.qblk = qblk;
    L.qhi = qhi;
    L.qcnt = qcnt;
    L.qlist = qlist;
    L.ent = ent;
    L.cand = cand;

    // One gather may serve MORE than the step it is made for: when the queries are unlikely all to finish inside the step
    // (levels level .. level+m-1), the walk is made at the radius of the level after it ("extension"), and every block a
    // query needs is listed either at the front of its list (needed inside the step) or at its back (needed at the
    // extension level only).  The step's passes work on the front parts; the next turn of the level loop then finds its
    // lists in LDS (`reuse`) and runs its passes without a walk of its own -- at BASELINE config 2 (levels 0 and 1 share a
    // step, 59 % of the queries go on to level 2) one pyramid walk per packet instead of two.  A packet whose extended
    // lists do not fit walks again for the step alone.
    bool reuse = false;          // wave-uniform: this level's lists are in LDS already
    int ext_next = a.first_ext;  // wave-uniform: extend the next gather
    bool ext_ok = a.first_ext >= 0;  // ... no more for this packet once its extended lists did not fit (its boxes only grow); < 0: never
    for (;;) {  // radius levels
      PHASE_END(4);
      // ---- 1. query records, conservative query boxes ---------------------------------------
      // This step serves levels level .. level+m-1 with ONE gather at the outermost radius.
      // (level, step and r are the same in every lane; readfirstlane says so to the compiler, which then
      // keeps them and everything derived from them in scalar registers and branches on them without exec masks)
      level = __builtin_amdgcn_readfirstlane(level);
      step = __builtin_amdgcn_readfirstlane(step);
      if (!a.start_radii) r = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r)));
      int m = step < 1 ? 1 : (step > kMaxStep ? kMaxStep : step);
      if (reuse) m = 1;  // the extension level of the previous gather
      if (level + m > a.max_rounds) m = a.max_rounds - level;
      const float r_in0 = r;                        // radius of the inner level of a two-level step
      float r_out = r;
      for (int j = 1; j < m; j++) r_out = r_out * 2.0f;
      {
        float *rec = qrec + lane * kQrecStride;
        rec[0] = q.x;
        rec[1] = q.y;
        rec[2] = q.z;
        rec[3] = __int_as_float(q.id);
        rec[4] = r_out;
      }
      int my_nblk = 0;  // lane = query: how many blocks of the packet's list this turn's passes visit for me
      int n_in = 0, n_out = 0;  // ... of my list: needed inside the step (its front part) / at the extension level only (its back)
      int my_own_pos = 0;       // ... and where my own block (the one holding me) sits in the front part
      bool too_big = false;  // this packet-level does not fit the LDS lists
      bool ext = !reuse && ext_next > 0 && level + m < a.max_rounds;
      ext = __builtin_amdgcn_readfirstlane((int)ext) != 0;
      if (reuse) {
        // front part + back part (team_pass maps positions past the front part to the back of the list); their lengths
        // wait in my record -- three more registers live across the passes would spill
        const int pk = __float_as_int(qrec[lane * kQrecStride + 5]);
        n_in = pk & 0xff;
        my_own_pos = (pk >> 8) & 0xff;
        n_out = (pk >> 24) & 0xff;
        my_nblk = n_in + n_out;
        reuse = false;
      } else
      for (;;) {  // the walk; a second time, for the step alone, if the extended lists do not fit
      const float r_walk = ext ? r_out * 2.0f : r_out;
      // conservative query boxes for the gather: every literal candidate of q at r_walk lies inside
      float lo_x = INFINITY, lo_y = INFINITY, lo_z = INFINITY, hi_x = -INFINITY, hi_y = -INFINITY, hi_z = -INFINITY;
      // ... and, per query, how far a block's box may be from the query POINT (largest per-axis gap) to be needed at the
      // walk's radius / inside the step: radius plus the margin of the candidate test (-inf: a query that needs nothing)
      float reach_walk = -INFINITY, reach_step = -INFINITY;
      {
        const float qabs = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z));
        const float mg = (qabs + 2.0f * r_walk) * 4.76837158203125e-07f;  // 2^-21
        if (active && q.x == q.x) {  // (a NaN query -- lbvh.hip makes every coordinate NaN -- keeps the empty box: it needs no block)
          lo_x = (q.x - r_walk) - 2.0f * mg;
          lo_y = (q.y - r_walk) - 2.0f * mg;
          lo_z = (q.z - r_walk) - 2.0f * mg;
          hi_x = (q.x + r_walk) + 2.0f * mg;
          hi_y = (q.y + r_walk) + 2.0f * mg;
          hi_z = (q.z + r_walk) + 2.0f * mg;
          reach_walk = r_walk + 2.0f * mg;
          reach_step = ext ? r_out + 2.0f * ((qabs + 2.0f * r_out) * 4.76837158203125e-07f) : reach_walk;
        }
      }
      // The pyramid is culled against FOUR boxes, one per 16 Morton-consecutive queries (= one leaf
      // block of queries), not against the packet's one union box: where the Z-curve jumps, or
      // among the outliers of a clustered set, the union covers space none of the queries needs.
      float s_lo_x[4], s_lo_y[4], s_lo_z[4], s_hi_x[4], s_hi_y[4], s_hi_z[4];
      {
        const float r_lo_x = t_team_min(lo_x), r_lo_y = t_team_min(lo_y), r_lo_z = t_team_min(lo_z);
        const float r_hi_x = t_team_max(hi_x), r_hi_y = t_team_max(hi_y), r_hi_z = t_team_max(hi_z);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          s_lo_x[j] = t_bcast(r_lo_x, 16 * j);
          s_lo_y[j] = t_bcast(r_lo_y, 16 * j);
          s_lo_z[j] = t_bcast(r_lo_z, 16 * j);
          s_hi_x[j] = t_bcast(r_hi_x, 16 * j);
          s_hi_y[j] = t_bcast(r_hi_y, 16 * j);
          s_hi_z[j] = t_bcast(r_hi_z, 16 * j);
        }
      }
      // Box against box as arithmetic, not as six compares and five mask operations: the largest separation along
      // any axis, max_a max(b.lo_a - hi_a, lo_a - b.hi_a), is <= 0 exactly when the closed boxes meet (block boxes
      // are never NaN, empty ones are (+inf, -inf): lbvh.hip).  Plain subtractions and maxima issue at twice the
      // rate of compares on gfx950 and leave the CU's scalar pipe alone (scripts/microbench/issue_rate.hip).
      auto separation = [](float b_lo_x, float b_lo_y, float b_lo_z, float b_hi_x, float b_hi_y, float b_hi_z, float q_lo_x, float q_lo_y,
                           float q_lo_z, float q_hi_x, float q_hi_y, float q_hi_z) -> float {
        const float sx = fmaxf(b_lo_x - q_hi_x, q_lo_x - b_hi_x), sy = fmaxf(b_lo_y - q_hi_y, q_lo_y - b_hi_y),
                    sz = fmaxf(b_lo_z - q_hi_z, q_lo_z - b_hi_z);
        return fmaxf(fmaxf(sx, sy), sz);
      };
      auto overlaps_packet = [&](const LbvhBox &bx) -> bool {
        float best = INFINITY;
#pragma unroll
        for (int j = 0; j < 4; j++)
          best = fminf(best, separation(bx.lo[0], bx.lo[1], bx.lo[2], bx.hi[0], bx.hi[1], bx.hi[2], s_lo_x[j], s_lo_y[j], s_lo_z[j], s_hi_x[j],
                                        s_hi_y[j], s_hi_z[j]));
        return best <= 0.f;
      };
      // The 64 child boxes of the wide node in hand, lane = child, go to LDS (the passes' entry lists are dead
      // during the gather) so that a surviving child's box reaches all 64 query lanes by three broadcast reads
      // instead of six v_readlane.
      float *node_boxes = (float *)ent;
      static_assert(64 * 6 * 4 <= Lay::kLdsEnt + Lay::kLdsCand, "the gather's child boxes borrow the entry lists' LDS");
      auto stash_boxes = [&](const LbvhBox &bx) {
        float2 *dst = (float2 *)(node_boxes + 6 * lane);
        dst[0] = make_float2(bx.lo[0], bx.lo[1]);
        dst[1] = make_float2(bx.lo[2], bx.hi[0]);
        dst[2] = make_float2(bx.hi[1], bx.hi[2]);
        t_wave_sync();
      };
      // the largest per-axis gap between the child box in lane `src` and MY query point (<= 0: I am inside): the child is
      // needed at radius R iff gap <= R (+ the margin: reach_*).  One distance serves the walk's radius and the step's.
      auto gap_to_me = [&](int src) -> float {
        // (src is wave-uniform: its byte offset is a scalar product -- left to itself the compiler folds the product and
        // the address add into one v_mad_u64_u32, a quarter-rate instruction, for every block tested)
        int boff;
        asm("s_mul_i32 %0, %1, 24" : "=s"(boff) : "s"(src));
        const float2 *b = (const float2 *)((const char *)node_boxes + boff);
        const float2 b0 = b[0], b1 = b[1], b2 = b[2];
        return separation(b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, q.x, q.y, q.z, q.x, q.y, q.z);
      };

      PHASE_END(0);
      // ---- 2+3. gather the packet's block list and the per-query masks ----------------------
      my_own_pos = 0;
      int cur = 0;  // lane = query: blocks in the front part of my list | blocks in its back part << 16
      uint8_t *my_list = qblk + lane * kMaxPerQuery;
      if (Lay::kHiWords) {
#pragma unroll
        for (int w = 0; w < Lay::kHiWords; w++) qhi[lane * Lay::kHiWords + w] = 0u;  // (my own words: no lane but mine writes them)
      }
      int nb = 0, scanned = 0;
      for (int tree = 0; tree < 2 && !too_big; tree++) {
        const LbvhWideView &wv = a.wide[tree];
        const int32_t tree_n = tree == 0 ? a.bvh.n : a.halo.n;
        if (tree_n <= 0 || wv.levels <= 0) continue;
        // Depth-first over the inner levels of the pyramid.  Wide nodes whose children are leaf blocks
        // are not expanded one by one but collected (the list grows down from the top of the stack
        // array) and drained together: in ascending Morton order, the next node's child boxes in
        // flight while this node's blocks are tested against the 64 queries.
        int sp = 0, nleaf = 0;
        if (lane == 0) stack[wv.levels > 1 ? 0 : kTeamStack - 1] = (wv.levels << 26) | 0;  // virtual root above the top level
        if (wv.levels > 1)
          sp = 1;
        else
          nleaf = 1;
        t_wave_sync();
        const int32_t last_block = wv.count[0] - 1;
        while ((sp > 0 || nleaf > 0) && !too_big) {
          if (sp > 0 && sp + nleaf + 63 <= kTeamStack) {  // room for the 64 children of one more node
            const int32_t e = __builtin_amdgcn_readfirstlane(stack[sp - 1]);  // same address in every lane
            sp--;
            const int lvl = (e >> 26) - 1;    // level of the children (>= 1 here)
            const int32_t first_child = (e & 0x3ffffff) * 64;
            const int32_t c = first_child + lane;
            const bool valid = lvl == wv.levels - 1 ? (lane < wv.count[lvl] && first_child == 0) : (c < wv.count[lvl]);
            LbvhBox bx = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            if (valid) bx = wv.level[lvl][c];
            wave_node_tests += 64;
            const bool ov = valid & overlaps_packet(bx);
            const unsigned long long om = __ballot(ov);
            const int cnt = __popcll(om);
            if (lvl > 1) {
              if (ov) stack[sp + t_rank(om)] = (lvl << 26) | c;
              sp += cnt;
            } else {
              // children of `c` are leaf blocks.  Keep a survivor only if some single QUERY box reaches
              // it (lanes = queries, the survivor's box by v_readlane): a dozen instructions that save
              // its 64 blocks from being tested one by one where only the group box, not a query,
              // overlapped.  Nodes are popped in descending Morton order and taken here from the
              // highest lane down, so filling the list downwards keeps it ascending in memory.
              unsigned long long rest = om;
              if (rest) stash_boxes(bx);
              while (rest) {
                const int src = 63 - __builtin_clzll(rest);
                rest &= ~(1ull << src);
                if (__ballot(gap_to_me(src) <= reach_walk) == 0ull) continue;
                if (lane == 0) stack[kTeamStack - 1 - nleaf] = (lvl << 26) | (first_child + src);
                nleaf++;
              }
            }
            t_wave_sync();
            continue;
          }
          if (nleaf == 0) {  // inner nodes alone fill the stack
            too_big = true;
            break;
          }
          auto leaf_boxes = [&](int i) -> LbvhBox {  // i-th collected node; index and children clamped
            const int32_t e = __builtin_amdgcn_readfirstlane(stack[kTeamStack - nleaf + (i < nleaf ? i : nleaf - 1)]);
            return wv.level[0][min((e & 0x3ffffff) * 64 + lane, last_block)];
          };
          LbvhBox bx_next = leaf_boxes(0);
          for (int i = 0; i < nleaf && !too_big; i++) {
            const int32_t e = __builtin_amdgcn_readfirstlane(stack[kTeamStack - nleaf + i]);
            const int32_t first_child = (e & 0x3ffffff) * 64;
            const LbvhBox bx = bx_next;
            bx_next = leaf_boxes(i + 1);
            const bool valid = first_child + lane <= last_block;
            wave_node_tests += 64;
            const bool ov = valid & overlaps_packet(bx);
            unsigned long long om = __ballot(ov);
            // A packet of far-apart queries (outliers of a clustered set) has a union box that covers
            // whole clusters none of its queries needs: testing those blocks one by one would keep
            // this wave busy long after the others have finished.  The lane kernel walks per query.
            scanned += __popcll(om);
            if (TKNN_DIAG_BUILD && (a.diag & 32)) diag_scanned += __popcll(om);
            if (scanned > kScanBudget) {
              too_big = true;
              break;
            }
            // leaf blocks: which of my 64 queries need block c?  (lanes = queries, box from LDS)
            if (om) stash_boxes(bx);
            while (om) {
              const int src = __ffsll((long long)om) - 1;
              om &= om - 1;
              const float gap = gap_to_me(src);
              bool need = gap <= reach_walk;
              if (TKNN_DIAG_BUILD && (a.diag & 8)) need = false;
              if (__ballot(need) == 0ull) continue;
              if (nb >= kMaxBlocks) {
                too_big = true;  // more blocks than a byte slot can name
                break;
              }
              const int32_t block = first_child + src;
              blk[nb] = (int32_t)((uint32_t)block | ((uint32_t)tree << 31));  // (every lane stores the same word: no exec mask to set up)
              // my own block is one of the packet's own four (a scalar test; its position: how many I have listed before it)
              if (tree == 0 && (uint32_t)(block - g * 4) < 4u && block == (slot >> 4)) my_own_pos = cur & 0xffff;
              if (need) {
                // inside the step: the front of my list, upwards; at the extension level only: its back, downwards.  A list
                // that outgrows its slots keeps writing inside them (clamped): the packet walks again or is handed over.
                const bool inner = gap <= reach_step;
                const int at = inner ? (cur & 0xffff) : kMaxPerQuery - 1 - (cur >> 16);
                const int at_c = min(max(at, 0), kMaxPerQuery - 1);  // (v_med3_i32)
                my_list[at_c] = (uint8_t)nb;
                if (Lay::kHiWords && nb >= 256) qhi[lane * Lay::kHiWords + (at_c >> 5)] |= 1u << (at_c & 31);
                cur += inner ? 1 : 0x10000;
              }
              nb++;
            }
          }
          nleaf = 0;
          t_wave_sync();
        }
        t_wave_sync();  // the stack array is rewritten by the next tree / shared with the passes
      }
      if (TKNN_DIAG_BUILD && (a.diag & 32)) diag_listed += nb;
      n_in = cur & 0xffff, n_out = cur >> 16;
      if (ext) {
        // extended lists that do not fit (a list longer than its slots, more blocks than the packet's list names, the
        // scan budget): once more, for the step alone
        if (too_big || __ballot(n_in + n_out > kMaxPerQuery) != 0ull) {
          ext = false;
          ext_ok = false;
          too_big = false;
          t_wave_sync();
          continue;
        }
        reuse = __ballot(active) != 0ull;
      }
      my_nblk = n_in;
      if (!ext) n_out = 0;
      break;
      }  // the walk
      // The packet's block list or scan budget is exhausted, or a query needs more blocks than its own
      // list holds: the packet's unfinished queries are handed over (team walk / lane rounds / wave
      // kernel, see solve_team), from this level on.  With four list registers (k > 32: boxes of
      // hundreds of candidates at the last level) only the queries whose own lists overflow go and
      // the rest carry on -- 0.49 M instead of 2.3 M of 10 M uniform points at k = 50.  (For k <= 32
      // that finer hand-over changed little on the clustered sets and cost 1 % on the benchmark: the
      // loop no longer ends at this point, which moved the register allocation.)
      if (NREG < 3) {
        if (__ballot(my_nblk > kMaxPerQuery) != 0ull) too_big = true;
        if (too_big) {
          if (active) {
            a.done[slot] = 0;
            a.isect_sorted[slot] = isect;
            a.next_level[slot] = level;
            my_handed++;
          }
          wave_min_handover = min(wave_min_handover, level);
          active = false;
          break;
        }
      } else if (too_big || __ballot(my_nblk > kMaxPerQuery) != 0ull) {
        const bool hand_over = active && (too_big || my_nblk > kMaxPerQuery);
        if (hand_over) {
          a.done[slot] = 0;
          a.isect_sorted[slot] = isect;
          a.next_level[slot] = level;
          my_handed++;
          active = false;
        }
        wave_min_handover = min(wave_min_handover, level);
        if (__ballot(active) == 0ull) break;
      }
      // read back by the teams: list length | position of my own block | length of the front part if a back part follows
      // (bits 24..31, which the teams ignore: the back part's length while the step's passes run on the front part)
      const int my_packed = my_nblk | (my_own_pos << 8) | (my_nblk > n_in ? n_in << 16 : n_out << 24);
      qrec[lane * kQrecStride + 5] = __int_as_float(my_packed);

      PHASE_END(1);
      // ---- 4. passes ---------------------------------------------------------------------------
      // One-level step: the box grows 8x per level, so a query with a few others at the previous
      // level will almost surely reach k now and goes straight to the fused count+select pass; the
      // rest are counted first and selected only if they turn out finished.  Multi-level step:
      // count all m nested boxes in one pass, pick the first level with >= k others, select there.
      // A row is written only when its query really has >= k others in the chosen box, so neither
      // speculation nor level grouping can change a result.
      const bool speculate = m == 1 && active && level > 0 && prev_others * 8u >= (uint32_t)(a.k + a.k / 2);
      const bool count_first = active && !speculate;
      // The four teams of a pass step through their lists in lockstep, so a group of four queries
      // costs the longest of its lists: the pass lists are ordered by list length (buckets of 4
      // blocks), which lifts the teams' occupancy from ~77 % to ~95 % on uniform data.
      auto build_qlist = [&](bool sel) -> int {
        const int bucket = my_nblk >> 2;  // <= kMaxPerQuery / 4 < 32
        if (lane == 0) qlist[64] = 0;
        t_wave_sync();
        if (sel) __hip_atomic_fetch_or((uint32_t *)&qlist[64], 1u << bucket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        t_wave_sync();
        uint32_t present = (uint32_t)__builtin_amdgcn_readfirstlane(qlist[64]);
        int base = 0, pos = 0;
        while (present) {
          const int b = __ffs((int)present) - 1;
          present &= present - 1u;
          const bool mine = sel && bucket == b;
          const unsigned long long bm = __ballot(mine);
          if (mine) pos = base + t_rank(bm);
          base += __popcll(bm);
        }
        if (sel) qlist[pos] = lane;
        t_wave_sync();
        return base;
      };
      {
        const int n_count = build_qlist(count_first);
        if (!(TKNN_DIAG_BUILD && (a.diag & 4))) {
          if (m > 1)
            team_pass<false, HALO, NREG, false, true>(a, L, n_count, g * 64, own_pts, halo_pts, lane);
          else
            team_pass<false, HALO, NREG, false, false>(a, L, n_count, g * 64, own_pts, halo_pts, lane);
        }
        t_wave_sync();
      }
      PHASE_END(2);
      // first level of the step at which I have >= k others (deviceCode.cu:118), -1 if none
      uint32_t c0 = 0, c1 = 0, selfc = 0;
      int fin_at = -1;
      if (count_first) {
        c0 = qcnt[lane * 2 + 0];
        const uint32_t w1 = qcnt[lane * 2 + 1];
        c1 = m > 1 ? (w1 & 0x3fffffffu) : 0u;
        selfc = w1 >> 31;
        if (c0 - selfc >= (uint32_t)a.k)
          fin_at = 0;
        else if (m > 1 && c1 - selfc >= (uint32_t)a.k)
          fin_at = 1;
      }
      const bool select_now = speculate || fin_at >= 0;
      if (m > 1 && fin_at == 0) qrec[lane * kQrecStride + 4] = r_in0;  // finishing inside the step: SELECT works in the inner box
      {
        t_wave_sync();
        const int n_select = build_qlist(select_now);
        if (!(TKNN_DIAG_BUILD && (a.diag & 2))) team_pass<true, HALO, NREG, FULL, false>(a, L, n_select, g * 64, own_pts, halo_pts, lane);
        t_wave_sync();
      }
      PHASE_END(3);
      bool finished = fin_at >= 0;
      uint32_t last_others = 0;
      if (speculate) {
        c0 = qcnt[lane * 2 + 0];
        selfc = qcnt[lane * 2 + 1] >> 31;
        finished = c0 - selfc >= (uint32_t)a.k;
        fin_at = finished ? 0 : -1;
      }
      int levels_run = 0;
      if (active) {
        levels_run = finished ? fin_at + 1 : m;
        isect += c0;
        if (levels_run > 1) isect += c1;
        my_levels += (uint32_t)levels_run;
        last_others = (m > 1 ? c1 : c0) - selfc;
        prev_others = last_others;
      }
      wave_levels = max(wave_levels, (int)t_wave_max((float)(active ? level + levels_run : 0)));
      {
        // exact box tests executed for my query: LBVH_BLOCK per listed block and pass
        const unsigned long long passes = (count_first ? 1ull : 0ull) + (select_now ? 1ull : 0ull);
        wave_point_tests += t_wave_sum(active ? (unsigned long long)my_nblk * LBVH_BLOCK * passes : 0ull);
      }
      if (finished) {
        // the team wrote the row (SELECT pass); the counter and the level come from the query's lane
        my_isect_sum += (unsigned long long)isect;
        if (a.out_isect) a.out_isect[row] = isect;
        if (a.out_fb) a.out_fb[(int64_t)row * a.k].intersections = isect;
        const int fin_level = level + (fin_at > 0 ? fin_at : 0);
        if (a.out_level) a.out_level[row] = fin_level;
        if ((qcnt[lane * 2 + 1] >> 30) & 1u) knn_flag_tie(a.tie, a.tie_list, a.counters, slot, fin_level, 2);  // SELECT saw exact-distance ties
      }
      active = active && !finished;
      level += m;
      t_wave_sync();
      if (__ballot(active) == 0ull) break;
      if (level >= a.max_rounds) {
        if (!a.allow_unfinished) wave_err |= 1;
        break;
      }
      for (int j = 0; j < m; j++) r = r * 2.0f;  // hostCode.cpp:321
      // how many levels the next gather should serve: the box grows 8x per level; group levels as
      // long as even the best-off unfinished query is unlikely to collect k others
      {
        const float pmax = t_wave_max(active ? (float)last_others : 0.f);
        float expect = pmax * 8.f;
        step = 1;
        while (step < kMaxStep && expect < 0.5f * (float)a.k) {
          step++;
          expect *= 8.f;
        }
        // ... and list the blocks of the level after the step, too, unless even the best-off query will almost
        // surely finish inside it (a packet walks again as long as ONE of its 64 queries is unfinished)
        ext_next = ext_ok && expect < 2.5f * (float)a.k ? 1 : 0;
      }
    }
    if (active) my_unfinished++;
  }

  const unsigned long long isum = t_wave_sum(my_isect_sum), lsum = t_wave_sum((unsigned long long)my_levels), usum = t_wave_sum((unsigned long long)my_unfinished),
                           hsum = t_wave_sum((unsigned long long)my_handed);
  if (lane == 0) {
    unsigned long long *st = a.counters + kStatBase + (blockIdx.x & (kStatStripes - 1)) * kStatStride;  // my stripe: [1] .. [9] as in counters
    atomicMax(&st[1], (unsigned long long)wave_levels);
    atomicAdd(&st[2], wave_node_tests);
    atomicAdd(&st[3], wave_point_tests);
    atomicAdd(&st[4], isum);
    atomicAdd(&st[6], lsum);
    if (usum) atomicAdd(&st[7], usum);
    if (hsum) {
      atomicAdd(&st[8], hsum);
      atomicMin(&st[9], (unsigned long long)wave_min_handover);
    }
    if (wave_err) atomicOr(&st[5], (unsigned long long)wave_err);
#if TKNN_DIAG_BUILD
    for (int i = 0; i < 5; i++) atomicAdd(&a.counters[10 + i], ph[i]);
    if (a.diag & 32) {
      atomicAdd(&a.counters[24], diag_scanned);
      atomicAdd(&a.counters[25], diag_listed);
    }
#endif
  }
}


// ---- team walk: the stragglers, one query per team, no packet lists --------------------------------
// Queries the packet kernel hands over (outliers whose boxes have grown over whole clusters, dense
// duplicates) have candidate sets far beyond its LDS lists.  Here a team walks the pyramid for ONE
// query: its 16 lanes test 16 child boxes of a wide node at a time and push the survivors on the
// team's LDS stack; a leaf block is 16 points = 16 lanes, tested, counted and selected exactly like
// in the passes above.  A child box that lies inside the part of the query's box where the
// candidate test is certain, and beyond the list's gate, is COUNTED (its points are consecutive
// sorted slots: the count is arithmetic) instead of walked -- a box over a cluster of 100 000 points
// costs a few hundred steps.  Levels loop inside the kernel (hostCode.cpp:285-340 per query).
constexpr int kWalkStack = 384;  // stack entries per team: up to 63 siblings wait on each of <= 6 levels

struct NotDone {
  __host__ __device__ bool operator()(uint8_t d) const { return d == 0; }
};

struct WalkLevel {  // per tree and pyramid level, in LDS: lanes of different teams are at different levels
  const LbvhBox *boxes;
  int32_t count;
  int32_t pad_;
};

template <bool HALO, int NREG>
__global__ void __launch_bounds__(kTeamBlock) TKNN_WALK_ATTR team_walk_kernel(TeamArgs a, const int32_t *slots, int32_t nslots) {
  __shared__ int32_t stack_mem[4 * kWalkStack];
  __shared__ WalkLevel levels[2][LBVH_WIDE_LEVELS];
  __shared__ unsigned long long cand_mem[4 * kCandCapacity];  // per team: candidates waiting to be merged into its list (t_merge_rows)
  const int lane = threadIdx.x & 63, team = lane >> 4, tl = lane & 15;
  int32_t *stack = stack_mem + team * kWalkStack;
  if (lane < 2 * LBVH_WIDE_LEVELS) {
    const int t = lane / LBVH_WIDE_LEVELS, l = lane % LBVH_WIDE_LEVELS;
    levels[t][l].boxes = a.wide[t].level[l];
    levels[t][l].count = a.wide[t].count[l];
  }
  t_wave_sync();
  unsigned long long isect_sum = 0, levels_sum = 0, node_tests = 0, point_tests = 0;
  unsigned int unfinished = 0, failed = 0;
  int max_level = 0;
  int turn_next = 0, turn_left = 0;  // wave-uniform: the slots of my turn at the cursor that are still to do
  for (;;) {
    if (turn_left == 0) {
      int got = 0;
      if (lane == 0) got = (int)atomicAdd(&a.counters[0], 4ull * (unsigned long long)max(a.grab, 1));
      turn_next = __builtin_amdgcn_readfirstlane(got);
      turn_left = max(a.grab, 1);
    }
    const int base = turn_next;
    turn_next += 4;
    turn_left--;
    if (base >= nslots) break;
    const bool has_q = base + team < nslots;
    const int32_t slot = has_q ? (slots ? slots[base + team] : base + team) : 0;
    const LbvhPoint q = a.bvh.points[slot];
    const int32_t row = a.bvh.prim_id[slot];
    int level = has_q ? a.next_level[slot] : 0;
    int64_t isect = has_q ? a.isect_sorted[slot] : 0;
    const float q_r0 = a.start_radii ? a.start_radii[row] : a.start_radius;  // per-query schedule, if asked for
    float r = q_r0;
    for (int i = 0; i < level; i++) r = r * 2.0f;
    bool active = has_q;
    while (__ballot(active) != 0ull) {  // one radius level for every team that is still at work
      const float mg = (fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z)) + 2.0f * r) * 4.76837158203125e-07f;  // 2^-21
      const float in_below = r - mg, in_upto = r + mg;
      // boxes that can hold a candidate meet [q - r - 2M, q + r + 2M]; every point of a box inside
      // [q - r + 2M, q + r - 2M] certainly is one
      const float rl = r + 2.0f * mg, rs = r - 2.0f * mg;
      uint32_t part = 0;  // my lane's share of the candidate count of this level
      uint32_t bd[NREG], bi[NREG];  // register j of lane t holds list entry 16 j + t (indices are compile-time: stays in VGPRs)
#pragma unroll
      for (int j = 0; j < NREG; j++) {
        bd[j] = 0x7f7fffffu;  // KNN_EMPTY_KEY = {FLT_MAX, 0}
        bi[j] = 0u;
      }
      float tau2 = INFINITY;
      bool overflow = false;
      const bool full = a.k == 16 * NREG;  // see team_pass
      uint32_t left_out = 0xffffffffu;
      auto kth_dist = [&]() -> float {
        uint32_t reg = bd[0];
#pragma unroll
        for (int j = 1; j < NREG; j++) reg = ((a.k - 1) >> 4) == j ? bd[j] : reg;
        return __uint_as_float(t_lane_read(reg, (team << 4) + ((a.k - 1) & 15)));
      };
      // candidates wait in the team's buffer and are merged sixteen at a time, as in the passes (one insert per lock-step
      // round was most of this kernel's time at k = 64: some 200 inserts per query)
      unsigned long long *my_cand = cand_mem + team * kCandCapacity;
      uint32_t fill_n = 0;
      auto merge_buffer = [&]() {
        t_wave_sync();
        t_merge_rows<NREG>(bd, bi, left_out, full, my_cand, fill_n, tl);
        t_wave_sync();
        fill_n = 0;
        tau2 = knn_gate_from_worst(kth_dist());
      };
      for (int tree = 0; tree < (HALO ? 2 : 1); tree++) {
        const LbvhWideView &wv = a.wide[tree];
        const LbvhView &tv = tree == 0 ? a.bvh : a.halo;
        if (tv.n <= 0 || wv.levels <= 0) continue;
        const int32_t clean_end = tv.n - (tv.nan_count ? *tv.nan_count : 0);  // NaN points sort last
        int sp = 0;
        if (active) {
          if (tl == 0) stack[0] = (wv.levels << 26) | 0;  // virtual root above the top level
          sp = 1;
        }
        t_wave_sync();
        while (__ballot(sp > 0) != 0ull) {
          const bool work = sp > 0;
          const int32_t e = work ? stack[sp - 1] : (1 << 26);
          if (work) sp--;
          const int lvl = (e >> 26) - 1;  // level of the children
          const int32_t first_child = (e & 0x3ffffff) * 64;
          const WalkLevel wl = levels[tree][lvl];
          // the virtual root has the top level's few boxes as its children
          const int32_t nchild = lvl == wv.levels - 1 ? (first_child == 0 ? wl.count : 0) : wl.count;
          LbvhBox bx4[4];  // the node's 64 child boxes, all four loads in flight at once
#pragma unroll
          for (int chunk = 0; chunk < 4; chunk++) {
            const int32_t c = first_child + 16 * chunk + tl;
            bx4[chunk] = LbvhBox{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            if (work && c < nchild) bx4[chunk] = wl.boxes[c];
          }
#pragma unroll
          for (int chunk = 0; chunk < 4; chunk++) {
            const int32_t c = first_child + 16 * chunk + tl;
            const bool valid = work && c < nchild;
            const LbvhBox bx = bx4[chunk];
            const bool ov = valid & (bx.lo[0] <= q.x + rl) & (bx.hi[0] >= q.x - rl) & (bx.lo[1] <= q.y + rl) &
                            (bx.hi[1] >= q.y - rl) & (bx.lo[2] <= q.z + rl) & (bx.hi[2] >= q.z - rl);
            node_tests += valid ? 1u : 0u;
            // inside the certain part of my box and beyond the gate: count, do not walk
            bool counted = false;
            if (ov) {
              const bool inside = (bx.lo[0] >= q.x - rs) & (bx.hi[0] <= q.x + rs) & (bx.lo[1] >= q.y - rs) &
                                  (bx.hi[1] <= q.y + rs) & (bx.lo[2] >= q.z - rs) & (bx.hi[2] <= q.z + rs);
              if (inside) {
                const float gx = fmaxf(fmaxf(bx.lo[0] - q.x, q.x - bx.hi[0]), 0.f), gy = fmaxf(fmaxf(bx.lo[1] - q.y, q.y - bx.hi[1]), 0.f),
                            gz = fmaxf(fmaxf(bx.lo[2] - q.z, q.z - bx.hi[2]), 0.f);
                const float m2 = (gx * gx + gy * gy) + gz * gz;
                const int64_t span = (int64_t)LBVH_BLOCK << (6 * lvl);  // points under one child of this level
                const int64_t first = (int64_t)c * span;
                if (m2 * 0.999995f > tau2 && first + span <= (int64_t)clean_end) {
                  part += (uint32_t)span;
                  counted = true;
                }
              }
            }
            const bool keep = ov && !counted;
            const uint32_t keep_mine = (uint32_t)(__ballot(keep) >> (team * 16)) & 0xffffu;
            if (lvl > 0) {
              if (sp + __popc(keep_mine) > kWalkStack) {
                overflow = true;
              } else {
                if (keep) stack[sp + __popc(keep_mine & ((1u << tl) - 1u))] = (lvl << 26) | c;
                sp += __popc(keep_mine);
              }
            } else {
              // children are leaf blocks: lanes become the 16 points of one block at a time
              uint32_t todo = keep_mine;
              while (__ballot(todo != 0u) != 0ull) {
                const bool has_b = todo != 0u;
                const int32_t b = first_child + 16 * chunk + (has_b ? __ffs((int)todo) - 1 : 0);
                todo &= todo - 1u;
                LbvhPoint p = {__uint_as_float(0x7fc00000u), 0.f, 0.f, -1};
                if (has_b) p = tv.points[(int64_t)b * LBVH_BLOCK + tl];
                point_tests += has_b ? 1u : 0u;
                const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                const float t = has_b ? fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) : __uint_as_float(0x7fc00000u);
                unsigned long long in_m = __ballot(t <= in_below);
                const unsigned long long maybe_m = __ballot(t <= in_upto) & ~in_m;
                if (maybe_m) in_m |= maybe_m & __ballot(knn_in_box(p.x, p.y, p.z, r, q.x, q.y, q.z));
                part = t_count(part, in_m);
                const float d2 = t_dist2(dx, dy, dz);
                unsigned long long pm = in_m & __ballot(p.id != q.id) & __ballot(d2 <= tau2);
                if (TKNN_DIAG_BUILD && (a.diag & 1)) pm = 0;
                if (pm) {
                  const uint32_t mine16 = (uint32_t)(pm >> (team << 4)) & 0xffffu;  // my team's lanes with a candidate
                  if ((mine16 >> tl) & 1u) my_cand[fill_n + __popc(mine16 & ((1u << tl) - 1u))] = ((unsigned long long)__float_as_uint(d2) << 32) | (uint32_t)p.id;
                  fill_n += __popc(mine16);
                  if (__ballot(fill_n >= 16u) != 0ull) merge_buffer();
                }
              }
              // a tighter gate for what comes next as soon as a handful of candidates wait
              if (__ballot(fill_n >= (uint32_t)TKNN_MERGE_AT) != 0ull) merge_buffer();
            }
          }
          t_wave_sync();
        }
      }
      if (__ballot(fill_n > 0u) != 0ull) merge_buffer();
      if (full) {
        // the smallest key left out by any merge, where the tie test below looks for it: lane 15
        uint32_t v = left_out;
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x128 /*row_ror:8*/, 0xf, 0xf, false));
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x124 /*row_ror:4*/, 0xf, 0xf, false));
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x122 /*row_ror:2*/, 0xf, 0xf, false));
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x121 /*row_ror:1*/, 0xf, 0xf, false));
        left_out = v;
      }
      // ---- the level's outcome, per team ----
      const uint32_t cnt = t_team_sum(part);
      const uint32_t others = cnt ? cnt - 1u : 0u;  // a query lies in its own box
      const bool fin = active && !overflow && others >= (uint32_t)a.k;
      if (active && overflow) {  // stack exhausted: leave the query to the lane rounds, state untouched
        failed += tl == 0 ? 1u : 0u;
        active = false;
      } else if (active) {
        isect += cnt;
        levels_sum += tl == 0 ? 1ull : 0ull;
        if (fin) {
#pragma unroll
          for (int reg = 0; reg < NREG; reg++) {
            const int j = tl + 16 * reg;
            if (j >= a.k) continue;
            const int64_t o = (int64_t)row * a.k + j;
            const int32_t prim = knn_key_prim(((uint64_t)bd[reg] << 32) | bi[reg]);
            const float d = __uint_as_float(bd[reg]);
            if (a.out_idx) a.out_idx[o] = prim;
            if (a.out_dist) a.out_dist[o] = d;
            if (a.out_fb) {
              tknnNeigh ev;
              ev.ind = prim;
              ev.dist = d;
              ev.numNeighbors = j == 0 ? 0 : a.k;
              ev.pad_ = 0;
              ev.intersections = j == 0 ? isect : 0;
              a.out_fb[o] = ev;
            }
          }
          bool tie = false, edge = false;
          const float qmax = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z));
#pragma unroll
          for (int reg = 0; reg < NREG; reg++) {
            uint32_t before = t_team_shr1(bd[reg]);
            if (reg > 0) before |= t_dpp<0x121>(bd[reg - 1]) & (tl == 0 ? 0xffffffffu : 0u);
            bool t = ((reg > 0) | (tl >= 1)) & (16 * reg + tl <= a.k) & (bd[reg] == before);
            const bool out_t = reg == NREG - 1 && full && tl == 15 && left_out == bd[reg];
            t |= out_t;
            t = t && tie_may_straddle(__uint_as_float(bd[reg]), q_r0, r, qmax, a.tie_span);
            tie |= t;
            edge |= t && (16 * reg + tl == a.k || out_t);  // (a tie with a candidate that is not written: knn_flag_tie)
          }
          const bool tied = ((uint32_t)(__ballot(tie) >> (team * 16)) & 0xffffu) != 0u;
          const bool tied_edge = ((uint32_t)(__ballot(edge) >> (team * 16)) & 0xffffu) != 0u;
          if (tl == 0) {
            if (a.out_isect) a.out_isect[row] = isect;
            if (a.out_level) a.out_level[row] = level;
            if (tied) knn_flag_tie(a.tie, a.tie_list, a.counters, slot, level, tied_edge ? 1 : 0);
            a.done[slot] = 1;
            isect_sum += (unsigned long long)isect;
          }
          max_level = max(max_level, level + 1);
          active = false;
        } else {
          level++;
          r = r * 2.0f;  // hostCode.cpp:321
          if (level >= a.max_rounds) {
            // out of rounds: the caller decides (allow_unfinished); the state says where it stopped
            if (tl == 0) {
              a.isect_sorted[slot] = isect;
              a.next_level[slot] = level;
              unfinished++;
            }
            max_level = max(max_level, level);
            active = false;
          }
        }
      }
    }
  }
  const unsigned long long isum = t_wave_sum(isect_sum), lsum = t_wave_sum(levels_sum), nt = t_wave_sum(node_tests),
                           pt = t_wave_sum(point_tests) * LBVH_BLOCK / 16, usum = t_wave_sum((unsigned long long)unfinished),
                           fsum = t_wave_sum((unsigned long long)failed);
  const int ml = (int)t_wave_max((float)max_level);
  if (lane == 0) {
    unsigned long long *st = a.counters + kStatBase + (blockIdx.x & (kStatStripes - 1)) * kStatStride;  // (my stripe: see kStatBase)
    atomicMax(&st[1], (unsigned long long)ml);
    atomicAdd(&st[2], nt);
    atomicAdd(&st[3], pt);
    atomicAdd(&st[4], isum);
    atomicAdd(&st[6], lsum);
    if (usum) atomicAdd(&st[7], usum);
    if (fsum) atomicAdd(&st[8], fsum);
  }
}


// Start-of-solve state of the packet kernel in one launch: done[] = 1 (a query is "done" unless handed
// over), tie[] = 0, counters = 0 except [9] = ~0 (min hand-over level), levels[] = -1 if asked for.
__global__ void __launch_bounds__(256) team_prep_kernel(uint8_t *done, uint8_t *tie, int64_t n, unsigned long long *counters, int32_t *levels) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  if (tid < kCounters + 8 * kXcdCounterStride) counters[tid] = tid == 9 ? ~0ull : 0ull;  // (+ the per-XCD packet counters behind them)
  if (tid < kStatStripes * kStatStride) counters[kStatBase + tid] = (tid & (kStatStride - 1)) == 9 ? ~0ull : 0ull;  // the statistics' stripes
  // hipMalloc'd arrays are 256-byte aligned: whole 16-byte words, then the last few bytes
  const int64_t words = n / 16;
  uint4 *d16 = reinterpret_cast<uint4 *>(done), *t16 = reinterpret_cast<uint4 *>(tie);
  const uint4 ones = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u), zeros = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = tid; i < words; i += nth) {
    d16[i] = ones;
    t16[i] = zeros;
  }
  for (int64_t i = words * 16 + tid; i < n; i += nth) {
    done[i] = 1;
    tie[i] = 0;
  }
  if (levels)
    for (int64_t i = tid; i < n; i += nth) levels[i] = -1;
}

// ---- exact-distance ties in the reference's order ---------------------------------------------------
// The reference's per-query lists persist over the rounds (deviceCode.cu:77-85 skips what is listed
// already, :116,:125 insert with a strict '<'): of two candidates at bit-identical fp32 distances the
// one that became a candidate in an EARLIER round stays ahead, whatever its index; inside one round
// the canonical order is by index (oracle/trueknn_oracle.c, decision 2).  The kernels above list by
// (dist, index) only -- the age would cost every insert a third key word -- and flag the rows where
// that can matter (a.tie).  Here a team redoes one flagged row with the full key (dist, first level,
// index): the first level of a candidate is the first radius of the doubling sequence whose box test
// it passes (the test is monotone in r).  Only neighbours within the row's k-th distance can be
// part of the answer, and that distance is already known (it does not depend on the order of ties):
// the walk prunes with it from the start, so a row costs a few wide nodes and leaf blocks.
constexpr int kFixStack = 384;  // as the walk: 6 KB of LDS per wave leaves room for 16 waves per CU; the ball pruning keeps stacks far below

struct HasTie {
  __host__ __device__ bool operator()(uint8_t t) const { return (t & 0x7f) != 0; }  // (bit 7 alone: team_pass's note, no flag)
};

template <bool HALO, int NREG>
__global__ void __launch_bounds__(kTeamBlock) tie_fix_kernel(TeamArgs a, const int32_t *slots, int32_t nslots) {
  // nslots < 0: `slots` is the kernels' own list (knn_flag_tie), as long as the device-side count says --
  // launched without the host knowing whether anything was flagged; nothing was: every wave leaves at once
  // nslots == -2: `slots` is a compacted list whose length hipCUB's select wrote to a.slot_count.  The host's
  // count of knn_flag_tie calls is only an upper bound of it (a wave-kernel solve that gives up on its
  // LDS stack is redone by the lane kernel, which flags the same rows a second time).
  if (nslots == -2)
    nslots = *a.slot_count;
  else if (nslots < 0)
    nslots = (int32_t)min(a.counters[kTieCounter], (unsigned long long)kTieListCap);
  if (nslots <= 0) return;
  __shared__ int32_t stack_mem[4 * kFixStack];
  __shared__ WalkLevel levels[2][LBVH_WIDE_LEVELS];
  const int lane = threadIdx.x & 63, team = lane >> 4, tl = lane & 15;
  int32_t *stack = stack_mem + team * kFixStack;
  if (lane < 2 * LBVH_WIDE_LEVELS) {
    const int t = lane / LBVH_WIDE_LEVELS, l = lane % LBVH_WIDE_LEVELS;
    levels[t][l].boxes = a.wide[t].level[l];
    levels[t][l].count = a.wide[t].count[l];
  }
  t_wave_sync();
  unsigned int failed = 0, stood = 0;
  int turn_next = 0, turn_left = 0;  // (as in team_walk_kernel)
  for (;;) {
    if (turn_left == 0) {
      int got = 0;
      if (lane == 0) got = (int)atomicAdd(&a.counters[kTieCounter + 1], 4ull * (unsigned long long)max(a.grab, 1));
      turn_next = __builtin_amdgcn_readfirstlane(got);
      turn_left = max(a.grab, 1);
    }
    const int base = turn_next;
    turn_next += 4;
    turn_left--;
    if (base >= nslots) break;
    bool active = base + team < nslots;
    const int32_t slot = active ? slots[base + team] : 0;
    const LbvhPoint q = a.bvh.points[slot];
    const int32_t row = a.bvh.prim_id[slot];
    const uint32_t tie_word = active ? (uint32_t)a.tie[slot] : 0u;
    active = active && (tie_word & 0x7fu) != 0u;  // a listed slot that is not flagged (any more) keeps its row
    const int level = active ? (int)(tie_word & 0x7fu) - 1 : 0;
    const float q_r0 = a.start_radii ? a.start_radii[row] : a.start_radius;
    float r = q_r0;
    for (int i = 0; i < level; i++) r = r * 2.0f;
    const float mg = (fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z)) + 2.0f * r) * 4.76837158203125e-07f;  // 2^-21, as in team_walk_kernel
    const float in_below = r - mg, in_upto = r + mg;
    const float rl = r + 2.0f * mg;
    // key word between distance and index: the level at which the candidate was first one
    auto first_level = [&](const LbvhPoint &p) -> uint32_t {
      float rr = q_r0;
      for (int l = 0; l < level; l++) {
        if (knn_in_box(p.x, p.y, p.z, rr, q.x, q.y, q.z)) return (uint32_t)l;
        rr = rr * 2.0f;
      }
      return (uint32_t)level;
    };
    uint32_t bd[NREG], bl[NREG], bi[NREG];
#pragma unroll
    for (int j = 0; j < NREG; j++) {
      bd[j] = 0x7f7fffffu;  // KNN_EMPTY_KEY = {FLT_MAX, 0}
      bl[j] = 0u;
      bi[j] = 0u;
    }
    auto kth_dist = [&]() -> float {
      uint32_t reg = bd[0];
#pragma unroll
      for (int j = 1; j < NREG; j++) reg = ((a.k - 1) >> 4) == j ? bd[j] : reg;
      return __uint_as_float(t_lane_read(reg, (team << 4) + ((a.k - 1) & 15)));
    };
    // the row's k-th distance, if the caller asked for distances (else the gate closes as the list fills)
    float tau2 = INFINITY;
    if (active) {
      const int64_t last = (int64_t)row * a.k + (a.k - 1);
      if (a.out_dist)
        tau2 = knn_gate_from_worst(a.out_dist[last]);
      else if (a.out_fb)
        tau2 = knn_gate_from_worst(a.out_fb[last].dist);
    }
    // Every tie of the row between two WRITTEN entries (no `edge`): the row is in (distance, index) order already, which is the
    // full key's order unless two tied neighbours became candidates at different levels.  Look that up from the row -- two
    // points per tied pair -- before walking for it: the duplicates of a data set (taxi pick-ups at one street corner) tie in
    // every row that holds both, always at one level, and are most of what is flagged on such sets (10 M taxi-like points with
    // 5 % duplicates, k = 10: 0.78 M rows flagged, 2.5 of the solve's 13.1 ms in this pass before this check).
    if (!HALO && a.row_slot && __ballot(active && !(tie_word & 0x80u)) != 0ull) {
      bool differs = false;
      const bool look = active && !(tie_word & 0x80u);
      uint32_t rd[NREG], ri[NREG];
#pragma unroll
      for (int reg = 0; reg < NREG; reg++) {
        const int j = tl + 16 * reg;
        rd[reg] = 0xffffffffu;  // (no entry: never equal to a distance)
        ri[reg] = 0u;
        if (look && j < a.k) {
          const int64_t o = (int64_t)row * a.k + j;
          if (a.out_dist && a.out_idx) {
            rd[reg] = __float_as_uint(a.out_dist[o]);
            ri[reg] = (uint32_t)a.out_idx[o];
          } else if (a.out_fb) {
            rd[reg] = __float_as_uint(a.out_fb[o].dist);
            ri[reg] = (uint32_t)a.out_fb[o].ind;
          } else {
            differs = true;  // (indices without distances: nothing to compare)
          }
        }
      }
#pragma unroll
      for (int reg = 0; reg < NREG; reg++) {
        uint32_t pd = t_team_shr1(rd[reg]), pi = t_team_shr1(ri[reg]);
        if (reg > 0) {
          const uint32_t lane0 = tl == 0 ? 0xffffffffu : 0u;
          pd = (pd & ~lane0) | (t_dpp<0x121>(rd[reg - 1]) & lane0);
          pi = (pi & ~lane0) | (t_dpp<0x121>(ri[reg - 1]) & lane0);
        }
        const int j = tl + 16 * reg;
        if (look && j >= 1 && j < a.k && rd[reg] == pd) {
          if (ri[reg] >= (uint32_t)a.bvh.n || pi >= (uint32_t)a.bvh.n) {
            differs = true;
          } else {
            const LbvhPoint pa = a.bvh.points[a.row_slot[ri[reg]]], pb = a.bvh.points[a.row_slot[pi]];
            differs |= first_level(pa) != first_level(pb);
          }
        }
      }
      const bool team_differs = ((uint32_t)(__ballot(differs) >> (team * 16)) & 0xffffu) != 0u;
      if (look && !team_differs) {  // the row stands
        active = false;
        stood += tl == 0 ? 1u : 0u;
      }
    }
    bool overflow = false;
    for (int tree = 0; tree < (HALO ? 2 : 1); tree++) {
      const LbvhWideView &wv = a.wide[tree];
      const LbvhView &tv = tree == 0 ? a.bvh : a.halo;
      if (tv.n <= 0 || wv.levels <= 0) continue;
      int sp = 0;
      if (active) {
        if (tl == 0) stack[0] = (wv.levels << 26) | 0;  // virtual root above the top level
        sp = 1;
      }
      t_wave_sync();
      while (__ballot(sp > 0) != 0ull) {
        const bool work = sp > 0;
        const int32_t e = work ? stack[sp - 1] : (1 << 26);
        if (work) sp--;
        const int lvl = (e >> 26) - 1;  // level of the children
        const int32_t first_child = (e & 0x3ffffff) * 64;
        const WalkLevel wl = levels[tree][lvl];
        const int32_t nchild = lvl == wv.levels - 1 ? (first_child == 0 ? wl.count : 0) : wl.count;
        LbvhBox bx4[4];  // the node's 64 child boxes, all four loads in flight at once
#pragma unroll
        for (int chunk = 0; chunk < 4; chunk++) {
          const int32_t c = first_child + 16 * chunk + tl;
          bx4[chunk] = LbvhBox{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
          if (work && c < nchild) bx4[chunk] = wl.boxes[c];
        }
#pragma unroll
        for (int chunk = 0; chunk < 4; chunk++) {
          const int32_t c = first_child + 16 * chunk + tl;
          const bool valid = work && c < nchild;
          const LbvhBox bx = bx4[chunk];
          const bool ov = valid & (bx.lo[0] <= q.x + rl) & (bx.hi[0] >= q.x - rl) & (bx.lo[1] <= q.y + rl) &
                          (bx.hi[1] >= q.y - rl) & (bx.lo[2] <= q.z + rl) & (bx.hi[2] >= q.z - rl);
          // beyond the gate: nothing in the box can be listed (0.999995: roundings of m2 and of the
          // points' distance arithmetic, as in team_walk_kernel)
          const float gx = fmaxf(fmaxf(bx.lo[0] - q.x, q.x - bx.hi[0]), 0.f), gy = fmaxf(fmaxf(bx.lo[1] - q.y, q.y - bx.hi[1]), 0.f),
                      gz = fmaxf(fmaxf(bx.lo[2] - q.z, q.z - bx.hi[2]), 0.f);
          const float m2 = (gx * gx + gy * gy) + gz * gz;
          const bool keep = ov && !(m2 * 0.999995f > tau2);
          const uint32_t keep_mine = (uint32_t)(__ballot(keep) >> (team * 16)) & 0xffffu;
          if (lvl > 0) {
            if (sp + __popc(keep_mine) > kFixStack) {
              overflow = true;
            } else {
              if (keep) stack[sp + __popc(keep_mine & ((1u << tl) - 1u))] = (lvl << 26) | c;
              sp += __popc(keep_mine);
            }
          } else {
            uint32_t todo = keep_mine;
            while (__ballot(todo != 0u) != 0ull) {
              const bool has_b = todo != 0u;
              const int32_t b = first_child + 16 * chunk + (has_b ? __ffs((int)todo) - 1 : 0);
              todo &= todo - 1u;
              LbvhPoint p = {__uint_as_float(0x7fc00000u), 0.f, 0.f, -1};
              if (has_b) p = tv.points[(int64_t)b * LBVH_BLOCK + tl];
              const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
              const float t = has_b ? fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) : __uint_as_float(0x7fc00000u);
              unsigned long long in_m = __ballot(t <= in_below);
              const unsigned long long maybe_m = __ballot(t <= in_upto) & ~in_m;
              if (maybe_m) in_m |= maybe_m & __ballot(knn_in_box(p.x, p.y, p.z, r, q.x, q.y, q.z));
              const float d2 = t_dist2(dx, dy, dz);
              unsigned long long pm = in_m & __ballot(p.id != q.id) & __ballot(d2 <= tau2);
              if (pm) {
                const uint32_t key_d = __float_as_uint(knn_sqrt(d2));
                const uint32_t key_l = ((pm >> lane) & 1ull) ? first_level(p) : 0u;
                const uint32_t key_i = (uint32_t)p.id;
                do {
                  const uint32_t pending_mine = (uint32_t)(pm >> (team * 16)) & 0xffffu;
                  const bool has = pending_mine != 0u;
                  const int src = (team << 4) + (has ? __ffs((int)pending_mine) - 1 : 0);
                  const uint32_t cd = t_lane_read(key_d, src), cl = t_lane_read(key_l, src), ci = t_lane_read(key_i, src);
                  const uint64_t chi = ((uint64_t)cd << 32) | cl;
                  const uint32_t lane0 = tl == 0 ? 0xffffffffu : 0u;
                  uint32_t nd_[NREG], nl_[NREG], ni_[NREG];
#pragma unroll
                  for (int j = 0; j < NREG; j++) {
                    uint32_t pd = t_team_shr1(bd[j]), pl = t_team_shr1(bl[j]), pi = t_team_shr1(bi[j]);
                    if (j > 0) {
                      pd |= t_dpp<0x121>(bd[j - 1]) & lane0;
                      pl |= t_dpp<0x121>(bl[j - 1]) & lane0;
                      pi |= t_dpp<0x121>(bi[j - 1]) & lane0;
                    }
                    const uint64_t cur_hi = ((uint64_t)bd[j] << 32) | bl[j], prev_hi = ((uint64_t)pd << 32) | pl;
                    const bool below_cur = (chi < cur_hi) | ((chi == cur_hi) & (ci < bi[j]));
                    const bool below_prev = (chi < prev_hi) | ((chi == prev_hi) & (ci < pi));
                    const bool take_prev = has & ((j > 0) | (tl != 0)) & below_prev;  // entry 0 has no entry before it
                    const bool take_c = has & below_cur;
                    nd_[j] = take_prev ? pd : (take_c ? cd : bd[j]);
                    nl_[j] = take_prev ? pl : (take_c ? cl : bl[j]);
                    ni_[j] = take_prev ? pi : (take_c ? ci : bi[j]);
                  }
#pragma unroll
                  for (int j = 0; j < NREG; j++) {
                    bd[j] = nd_[j];
                    bl[j] = nl_[j];
                    bi[j] = ni_[j];
                  }
                  pm &= ~__ballot(lane == src);
                } while (pm);
                tau2 = fminf(tau2, knn_gate_from_worst(kth_dist()));
              }
            }
          }
        }
        t_wave_sync();
      }
    }
    if (active && overflow) {
      failed += tl == 0 ? 1u : 0u;  // the row keeps its (dist, index) order; reported in tknnSolveInfo.tie_rows_left
    } else if (active) {
#pragma unroll
      for (int reg = 0; reg < NREG; reg++) {
        const int j = tl + 16 * reg;
        if (j >= a.k) continue;
        const int64_t o = (int64_t)row * a.k + j;
        const int32_t prim = knn_key_prim(((uint64_t)bd[reg] << 32) | bi[reg]);
        const float d = __uint_as_float(bd[reg]);
        if (a.out_idx) a.out_idx[o] = prim;
        if (a.out_dist) a.out_dist[o] = d;
        if (a.out_fb) {
          a.out_fb[o].ind = prim;
          a.out_fb[o].dist = d;
        }
      }
    }
  }
  const unsigned long long fsum = t_wave_sum((unsigned long long)failed);
  if (lane == 0 && fsum) atomicAdd(&a.counters[kTieCounter + 2], fsum);
  const unsigned long long ssum = t_wave_sum((unsigned long long)stood);
  if (lane == 0 && ssum) atomicAdd(&a.counters[kTieCounter + 3], ssum);
}


// ---- k > 64: the list in memory -----------------------------------------------------------------------------------------
// The reference keeps every query's k-list in global memory and takes any k from its command line (hostCode.cpp:111,
// deviceCode.cu:77-134).  The kernels above hold up to 64 entries in registers.  Larger lists live in memory, sixteen keys
// to a CHUNK (lane j of the team reads and writes entry 16 c + j of chunk c: one coalesced 256-byte access, and always the
// lane's own words), per team that is resident on the device, not per query: a query's list is built anew at every radius
// level, as in team_walk_kernel, whose walk of the pyramid this kernel shares -- one query per team, its sixteen lanes the
// child boxes of a wide node or the points of a leaf block.  Candidates wait in the team's LDS buffer and are merged a sorted
// row of sixteen at a time (t_merge_rows with the registers in memory): the row goes to the first chunk whose largest distance
// is not below the row's smallest (the chunks' maxima sit in LDS), meets it mirrored -- the sixteen smallest of both stay, the
// sixteen largest travel on to the next chunk -- until what travels is empty.
// Keys carry THREE words, (distance, level at which the neighbour first was a candidate, index): the reference's order of
// bit-identical distances (section 1 of DESIGN.md; tie_fix_kernel's key), so rows come out final and no tie pass follows.
struct BigKey {
  uint32_t d, l, i, pad;
};
constexpr int kBigMaxChunks = TKNN_MAX_K / 16;
static_assert(TKNN_MAX_K % 16 == 0 && kBigMaxChunks <= 64, "chunk maxima: 64 words of LDS per team");

__device__ __forceinline__ bool t_less3(uint32_t ad, uint32_t al, uint32_t ai, uint32_t bd, uint32_t bl, uint32_t bi) {
  const uint64_t ah = ((uint64_t)ad << 32) | al, bh = ((uint64_t)bd << 32) | bl;
  return (ah < bh) | ((ah == bh) & (ai < bi));
}
// one compare-exchange of three-word keys with the lane whose key is (pd, pl, pi): the lower lane keeps the smaller key
__device__ __forceinline__ void t_exchange3(uint32_t &kd, uint32_t &kl, uint32_t &ki, uint32_t pd, uint32_t pl, uint32_t pi, bool upper) {
  const bool take = t_less3(pd, pl, pi, kd, kl, ki) != upper;
  kd = take ? pd : kd;
  kl = take ? pl : kl;
  ki = take ? pi : ki;
}
#define T_EX3(CTRL, UP) t_exchange3(kd, kl, ki, t_dpp<CTRL>(kd), t_dpp<CTRL>(kl), t_dpp<CTRL>(ki), UP)
__device__ __forceinline__ void t_sort16_3(uint32_t &kd, uint32_t &kl, uint32_t &ki, int tl) {
  const bool up1 = (tl & 1) != 0, up2 = (tl & 2) != 0, up4 = (tl & 4) != 0, up8 = (tl & 8) != 0;
  T_EX3(0xb1, up1);
  T_EX3(0x1b, up2);
  T_EX3(0xb1, up1);
  T_EX3(0x141, up4);
  T_EX3(0x4e, up2);
  T_EX3(0xb1, up1);
  T_EX3(0x140, up8);
  t_exchange3(kd, kl, ki, t_xor4(kd), t_xor4(kl), t_xor4(ki), up4);
  T_EX3(0x4e, up2);
  T_EX3(0xb1, up1);
}
__device__ __forceinline__ void t_clean16_3(uint32_t &kd, uint32_t &kl, uint32_t &ki, int tl) {
  const bool up1 = (tl & 1) != 0, up2 = (tl & 2) != 0, up4 = (tl & 4) != 0, up8 = (tl & 8) != 0;
  T_EX3(0x128, up8);
  t_exchange3(kd, kl, ki, t_xor4(kd), t_xor4(kl), t_xor4(ki), up4);
  T_EX3(0x4e, up2);
  T_EX3(0xb1, up1);
}
#undef T_EX3

template <bool HALO>
__global__ void __launch_bounds__(kTeamBlock) TKNN_BIGK_ATTR bigk_walk_kernel(TeamArgs a, BigKey *lists, int chunks) {
  __shared__ int32_t stack_mem[4 * kWalkStack];
  __shared__ WalkLevel levels[2][LBVH_WIDE_LEVELS];
  __shared__ BigKey cand_mem[4 * kCandCapacity];     // per team: candidates waiting to be merged, (squared distance, first level, index)
  __shared__ uint32_t cmax_mem[4 * kBigMaxChunks];   // per team and chunk of its list: the largest distance in it (bits)
  const int lane = threadIdx.x & 63, team = lane >> 4, tl = lane & 15;
  int32_t *stack = stack_mem + team * kWalkStack;
  BigKey *my_cand = cand_mem + team * kCandCapacity;
  uint32_t *my_cmax = cmax_mem + team * kBigMaxChunks;
  BigKey *my_list = lists + ((size_t)blockIdx.x * 4 + (size_t)team) * (size_t)chunks * 16;
  if (lane < 2 * LBVH_WIDE_LEVELS) {
    const int t = lane / LBVH_WIDE_LEVELS, l = lane % LBVH_WIDE_LEVELS;
    levels[t][l].boxes = a.wide[t].level[l];
    levels[t][l].count = a.wide[t].count[l];
  }
  t_wave_sync();
  const int32_t n = a.bvh.n;
  const int kc = (a.k - 1) >> 4;  // the chunk of the k-th entry
  unsigned long long isect_sum = 0, levels_sum = 0, node_tests = 0, point_tests = 0;
  unsigned int unfinished = 0, failed = 0;
  int max_level = 0;
  int turn_next = 0, turn_left = 0;  // (as in team_walk_kernel)
  for (;;) {
    if (turn_left == 0) {
      int got = 0;
      if (lane == 0) got = (int)atomicAdd(&a.counters[0], 4ull * (unsigned long long)max(a.grab, 1));
      turn_next = __builtin_amdgcn_readfirstlane(got);
      turn_left = max(a.grab, 1);
    }
    const int base = turn_next;
    turn_next += 4;
    turn_left--;
    if (base >= n) break;
    const int32_t slot = min(base + team, n - 1);
    bool has_q = base + team < n;
    if (has_q && a.skip && (int32_t)a.skip[slot] == a.skip_is) has_q = false;  // (tknnSolveOptions.phase: not a query of this call)
    const LbvhPoint q = a.bvh.points[slot];
    const int32_t row = a.bvh.prim_id[slot];
    int level = 0;
    int64_t isect = 0;
    const float q_r0 = a.start_radii ? a.start_radii[row] : a.start_radius;
    float r = q_r0;
    bool active = has_q;
    // A level only COUNTS (deviceCode.cu:74) as long as the query is unlikely to finish at it -- its box grows eightfold per
    // level, so: fewer than k / 5 others at the level before, and in the first levels the scene's mean density (a.first_step)
    // -- and keeps no list; a level that counts k others after all is walked once more, with the list.  (Every level with its
    // list: 10 M uniform points at k = 65 took three times the k = 64 solve.)
    uint32_t prev_others = 0;
    bool again = false;  // this level has counted k others without a list
    while (__ballot(active) != 0ull) {  // one radius level for every team that is still at work
      const bool select_on = again || level >= a.first_step || prev_others * 5u >= (uint32_t)a.k;
      const unsigned long long select_m = __ballot(select_on);
      const float mg = (fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z)) + 2.0f * r) * 4.76837158203125e-07f;  // 2^-21
      const float in_below = r - mg, in_upto = r + mg;
      const float rl = r + 2.0f * mg, rs = r - 2.0f * mg;
      uint32_t part = 0;       // my lane's share of the candidate count of this level
      uint32_t n_list = 0;     // keys in my team's list (the same in its lanes)
      uint32_t kth_bits = 0x7f7fffffu;  // distance of the list's k-th entry (FLT_MAX: not that many yet)
      // (a squared distance that has overflowed is no neighbour: the reference's strict `<` against its initial FLT_MAX,
      // hostCode.cpp:41, deviceCode.cu:116)
      float tau2 = 3.402823466e+38f;
      bool overflow = false;
      uint32_t fill_n = 0;
      // key word between distance and index: the level at which the candidate was first one (the box test is monotone in r)
      auto first_level = [&](const LbvhPoint &p) -> uint32_t {
        float rr = q_r0;
        for (int l = 0; l < level; l++) {
          if (knn_in_box(p.x, p.y, p.z, rr, q.x, q.y, q.z)) return (uint32_t)l;
          rr = rr * 2.0f;
        }
        return (uint32_t)level;
      };
      auto merge_buffer = [&]() {
        t_wave_sync();
#pragma unroll 1
        for (int rw = 0; rw < kCandCapacity / 16; rw++) {
          if (rw > 0 && __ballot(fill_n > 16u * (uint32_t)rw) == 0ull) break;
          const uint32_t at = 16u * (uint32_t)rw + (uint32_t)tl;
          const bool have = at < fill_n;
          BigKey c = {0u, 0u, 0u, 0u};
          if (have) c = my_cand[at];
          const float dist = knn_sqrt(__uint_as_float(c.d));
          uint32_t kd = have ? __float_as_uint(dist) : 0x7f7fffffu, kl = have ? c.l : 0u, ki = have ? c.i : 0u;  // the empty key past the end
          t_sort16_3(kd, kl, ki, tl);
          const uint32_t in_row = fill_n > 16u * (uint32_t)rw ? min(fill_n - 16u * (uint32_t)rw, 16u) : 0u;
          const uint32_t row_min = t_lane_read(kd, team << 4);
          const int nch = (int)((n_list + 15u) >> 4);  // chunks that hold keys
          // the first chunk a key of the row can get into: the chunks whose largest distance is below the row's smallest stay
          uint32_t below = 0;
          for (int j = tl; j < nch; j += 16) below += my_cmax[j] < row_min ? 1u : 0u;
          const int j0 = (int)t_team_sum(below);
          const int jend = min(nch, chunks - 1);  // ... and the last: the first chunk without keys, if the list has room for one
          bool live = in_row > 0u;                // what travels still holds keys
          int j_first = live ? j0 : 0x7fffffff;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) j_first = min(j_first, __shfl_xor(j_first, off));
          for (int j = j_first; __ballot(live && j <= jend) != 0ull; j++) {
            const bool on = live && j >= j0 && j <= jend;
            BigKey e = {0x7f7fffffu, 0u, 0u, 0u};
            if (on && j < nch) e = my_list[(size_t)j * 16 + tl];
            const uint32_t od = t_dpp<0x140>(kd), ol = t_dpp<0x140>(kl), oi = t_dpp<0x140>(ki);  // the row's key 15 - tl
            const bool take = t_less3(od, ol, oi, e.d, e.l, e.i);
            uint32_t hd = take ? e.d : od, hl = take ? e.l : ol, hi_i = take ? e.i : oi;  // the larger of the pair: travels on
            uint32_t ld = take ? od : e.d, ll = take ? ol : e.l, li = take ? oi : e.i;    // the smaller: stays in this chunk
            t_clean16_3(ld, ll, li, tl);
            t_clean16_3(hd, hl, hi_i, tl);
            const uint32_t kth_here = t_lane_read(ld, (team << 4) + ((a.k - 1) & 15));
            if (on) {
              my_list[(size_t)j * 16 + tl] = BigKey{ld, ll, li, 0u};
              if (tl == 15) my_cmax[j] = ld;
              if (j == kc) kth_bits = kth_here;
              kd = hd, kl = hl, ki = hi_i;
            }
            const bool more = ((uint32_t)(__ballot(kd != 0x7f7fffffu) >> (team * 16)) & 0xffffu) != 0u;
            if (on) live = more && j < nch;  // (a chunk that held no key takes all that travels)
          }
          n_list = min((uint32_t)chunks * 16u, n_list + in_row);
          t_wave_sync();  // (the chunk maxima are read by the next row's lanes)
        }
        fill_n = 0;
        tau2 = fminf(3.402823466e+38f, knn_gate_from_worst(__uint_as_float(kth_bits)));
      };
      for (int tree = 0; tree < (HALO ? 2 : 1); tree++) {
        const LbvhWideView &wv = a.wide[tree];
        const LbvhView &tv = tree == 0 ? a.bvh : a.halo;
        if (tv.n <= 0 || wv.levels <= 0) continue;
        const int32_t clean_end = tv.n - (tv.nan_count ? *tv.nan_count : 0);  // NaN points sort last
        int sp = 0;
        if (active) {
          if (tl == 0) stack[0] = (wv.levels << 26) | 0;  // virtual root above the top level
          sp = 1;
        }
        t_wave_sync();
        while (__ballot(sp > 0) != 0ull) {
          const bool work = sp > 0;
          const int32_t e = work ? stack[sp - 1] : (1 << 26);
          if (work) sp--;
          const int lvl = (e >> 26) - 1;  // level of the children
          const int32_t first_child = (e & 0x3ffffff) * 64;
          const WalkLevel wl = levels[tree][lvl];
          const int32_t nchild = lvl == wv.levels - 1 ? (first_child == 0 ? wl.count : 0) : wl.count;
          LbvhBox bx4[4];
#pragma unroll
          for (int chunk = 0; chunk < 4; chunk++) {
            const int32_t c = first_child + 16 * chunk + tl;
            bx4[chunk] = LbvhBox{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            if (work && c < nchild) bx4[chunk] = wl.boxes[c];
          }
#pragma unroll 1
          for (int chunk = 0; chunk < 4; chunk++) {  // (not unrolled: the merge below is large; the box by selects, not by an index)
            const int32_t c = first_child + 16 * chunk + tl;
            const bool valid = work && c < nchild;
            LbvhBox bx = bx4[0];
#pragma unroll
            for (int u = 1; u < 4; u++)
              if (chunk == u) bx = bx4[u];
            const bool ov = valid & (bx.lo[0] <= q.x + rl) & (bx.hi[0] >= q.x - rl) & (bx.lo[1] <= q.y + rl) &
                            (bx.hi[1] >= q.y - rl) & (bx.lo[2] <= q.z + rl) & (bx.hi[2] >= q.z - rl);
            node_tests += valid ? 1u : 0u;
            bool counted = false;  // inside the certain part of my box and beyond the gate: count, do not walk (team_walk_kernel)
            if (ov) {
              const bool inside = (bx.lo[0] >= q.x - rs) & (bx.hi[0] <= q.x + rs) & (bx.lo[1] >= q.y - rs) &
                                  (bx.hi[1] <= q.y + rs) & (bx.lo[2] >= q.z - rs) & (bx.hi[2] <= q.z + rs);
              if (inside) {
                const float gx = fmaxf(fmaxf(bx.lo[0] - q.x, q.x - bx.hi[0]), 0.f), gy = fmaxf(fmaxf(bx.lo[1] - q.y, q.y - bx.hi[1]), 0.f),
                            gz = fmaxf(fmaxf(bx.lo[2] - q.z, q.z - bx.hi[2]), 0.f);
                const float m2 = (gx * gx + gy * gy) + gz * gz;
                const int64_t span = (int64_t)LBVH_BLOCK << (6 * lvl);  // points under one child of this level
                const int64_t first = (int64_t)c * span;
                if (m2 * 0.999995f > tau2 && first + span <= (int64_t)clean_end) {
                  part += (uint32_t)span;
                  counted = true;
                }
              }
            }
            const bool keep = ov && !counted;
            const uint32_t keep_mine = (uint32_t)(__ballot(keep) >> (team * 16)) & 0xffffu;
            if (lvl > 0) {
              if (sp + __popc(keep_mine) > kWalkStack) {
                overflow = true;
              } else {
                if (keep) stack[sp + __popc(keep_mine & ((1u << tl) - 1u))] = (lvl << 26) | c;
                sp += __popc(keep_mine);
              }
            } else {
              uint32_t todo = keep_mine;
              while (__ballot(todo != 0u) != 0ull) {
                const bool has_b = todo != 0u;
                const int32_t b = first_child + 16 * chunk + (has_b ? __ffs((int)todo) - 1 : 0);
                todo &= todo - 1u;
                LbvhPoint p = {__uint_as_float(0x7fc00000u), 0.f, 0.f, -1};
                if (has_b) p = tv.points[(int64_t)b * LBVH_BLOCK + tl];
                point_tests += has_b ? 1u : 0u;
                const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                const float t = has_b ? fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) : __uint_as_float(0x7fc00000u);
                unsigned long long in_m = __ballot(t <= in_below);
                const unsigned long long maybe_m = __ballot(t <= in_upto) & ~in_m;
                if (maybe_m) in_m |= maybe_m & __ballot(knn_in_box(p.x, p.y, p.z, r, q.x, q.y, q.z));
                part = t_count(part, in_m);
                const float d2 = t_dist2(dx, dy, dz);
                const unsigned long long pm = in_m & __ballot(p.id != q.id) & __ballot(d2 <= tau2) & select_m;
                if (pm) {
                  const uint32_t mine16 = (uint32_t)(pm >> (team << 4)) & 0xffffu;  // my team's lanes with a candidate
                  if ((mine16 >> tl) & 1u)
                    my_cand[fill_n + __popc(mine16 & ((1u << tl) - 1u))] = BigKey{__float_as_uint(d2), first_level(p), (uint32_t)p.id, 0u};
                  fill_n += __popc(mine16);
                  if (__ballot(fill_n >= 16u) != 0ull) merge_buffer();
                }
              }
            }
          }
          t_wave_sync();
        }
      }
      if (__ballot(fill_n > 0u) != 0ull) merge_buffer();
      // ---- the level's outcome, per team ----
      const uint32_t cnt = t_team_sum(part);
      const uint32_t others = cnt ? cnt - 1u : 0u;  // a query lies in its own box
      const bool fin = active && !overflow && others >= (uint32_t)a.k;
      if (active && overflow) {  // (cannot happen: 63 siblings wait on each of at most six levels; reported, never wrong)
        failed += tl == 0 ? 1u : 0u;
        active = false;
      } else if (active && fin && !select_on) {
        again = true;  // the same level once more, with its list
      } else if (active) {
        again = false;
        prev_others = others;
        isect += cnt;
        levels_sum += tl == 0 ? 1ull : 0ull;
        if (fin) {
          for (int j = 0; j <= kc; j++) {
            const int en = 16 * j + tl;
            if (en >= a.k) continue;
            const BigKey ky = my_list[(size_t)j * 16 + tl];
            const int64_t o = (int64_t)row * a.k + en;
            const int32_t prim = (ky.d == 0x7f7fffffu && ky.i == 0u) ? -1 : (int32_t)ky.i;
            const float d = __uint_as_float(ky.d);
            if (a.out_idx) a.out_idx[o] = prim;
            if (a.out_dist) a.out_dist[o] = d;
            if (a.out_fb) {
              tknnNeigh ev;
              ev.ind = prim;
              ev.dist = d;
              ev.numNeighbors = en == 0 ? 0 : a.k;
              ev.pad_ = 0;
              ev.intersections = en == 0 ? isect : 0;
              a.out_fb[o] = ev;
            }
          }
          if (tl == 0) {
            if (a.out_isect) a.out_isect[row] = isect;
            if (a.out_level) a.out_level[row] = level;
            a.done[slot] = 1;
            isect_sum += (unsigned long long)isect;
          }
          max_level = max(max_level, level + 1);
          active = false;
        } else {
          level++;
          r = r * 2.0f;  // hostCode.cpp:321
          if (level >= a.max_rounds) {
            if (tl == 0) {
              a.isect_sorted[slot] = isect;
              a.next_level[slot] = level;
              unfinished++;
            }
            max_level = max(max_level, level);
            active = false;
          }
        }
      }
    }
  }
  const unsigned long long isum = t_wave_sum(isect_sum), lsum = t_wave_sum(levels_sum), nt = t_wave_sum(node_tests),
                           pt = t_wave_sum(point_tests) * LBVH_BLOCK / 16, usum = t_wave_sum((unsigned long long)unfinished),
                           fsum = t_wave_sum((unsigned long long)failed);
  const int ml = (int)t_wave_max((float)max_level);
  if (lane == 0) {
    unsigned long long *st = a.counters + kStatBase + (blockIdx.x & (kStatStripes - 1)) * kStatStride;  // (my stripe: see kStatBase)
    atomicMax(&st[1], (unsigned long long)ml);
    atomicAdd(&st[2], nt);
    atomicAdd(&st[3], pt);
    atomicAdd(&st[4], isum);
    atomicAdd(&st[6], lsum);
    if (usum) atomicAdd(&st[7], usum);
    if (fsum) atomicAdd(&st[8], fsum);
  }
}

}  // namespace

bool Engine::team_kernel_supports(int k) { return k >= 1 && k <= 64; }

bool Engine::bigk_supports(int k) { return k > 64 && k <= TKNN_MAX_K; }

// The team kernels' end-of-wave statistics live in stripes (kStatBase): zeroed before a launch of a walk (the packet kernel's
// prep launch does it itself), copied behind h_counters_[16] after it and folded into h_counters_[1 .. 9] once the stream is idle
void Engine::reset_stat_stripes(hipStream_t s) {
  OWLMI_HIP(hipMemsetAsync(counters_ + kStatBase, 0, kStatStripes * kStatStride * sizeof(unsigned long long), s));
}
void Engine::fetch_stat_stripes(hipStream_t s) {
  OWLMI_HIP(hipMemcpyAsync(h_counters_ + 16, counters_ + kStatBase, kStatStripes * kStatStride * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
}
void Engine::fold_stat_stripes(bool with_min) {
  for (int i = 1; i < 10; i++) h_counters_[i] = i == 9 && with_min ? ~0ull : 0ull;
  for (int j = 0; j < kStatStripes; j++) {
    const unsigned long long *st = h_counters_ + 16 + j * kStatStride;
    h_counters_[1] = std::max(h_counters_[1], st[1]);
    for (int i : {2, 3, 4, 6, 7, 8}) h_counters_[i] += st[i];
    h_counters_[5] |= st[5];
    if (with_min) h_counters_[9] = std::min(h_counters_[9], st[9]);
  }
}

// k > 64: every query through bigk_walk_kernel, one query per team, the k-lists in memory (one per resident team)
void Engine::solve_bigk(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s) {
  const int64_t n = bvh_.size();
  TeamArgs a;
  std::memset(&a, 0, sizeof a);
  a.bvh = bvh_.view();
  a.halo = halo_view();
  a.wide[0] = bvh_.wide_view();
  if (halo_count() > 0) a.wide[1] = halo_.wide_view();
  a.start_radius = sa.start_radius;
  a.start_radii = sa.d_start_radii;
  a.k = sa.k;
  a.max_rounds = sa.max_rounds;
  a.allow_unfinished = sa.allow_unfinished ? 1 : 0;
  // the first level that keeps a list whatever the level before has counted: where a box is expected to hold k / 2 others at the
  // scene's mean density (a work estimate only; a per-query radius schedule has none: every level keeps its list)
  a.first_step = sa.d_start_radii ? 0 : first_step_estimate(sa) - 1;
  a.out_idx = sa.d_idx;
  a.out_dist = sa.d_dist;
  a.out_isect = sa.d_isect;
  a.out_fb = sa.d_fb;
  a.out_level = sa.d_levels;
  a.done = done_;
  a.tie = tie_;
  a.tie_list = tie_list_;
  a.skip = sa.phase ? boundary_ : nullptr;
  a.skip_is = sa.phase == 1 ? 1 : 0;
  a.isect_sorted = isect_sorted_;
  a.next_level = next_level_;
  a.counters = counters_;
  const bool with_halo = halo_count() > 0;
  hipDeviceProp_t prop;
  OWLMI_HIP(hipGetDeviceProperties(&prop, device_));
  int per_cu = 4;
  const void *entry = with_halo ? (const void *)bigk_walk_kernel<true> : (const void *)bigk_walk_kernel<false>;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, entry, kTeamBlock, 0) != hipSuccess) per_cu = 4;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)prop.multiProcessorCount * std::max(1, per_cu)));
  const int chunks = (sa.k + 15) / 16;
  const size_t list_bytes = (size_t)blocks * 4 * (size_t)chunks * 16 * sizeof(BigKey);
  if (list_bytes > wave_ws_bytes_) {
    if (wave_ws_) (void)hipFree(wave_ws_);
    wave_ws_ = nullptr;
    wave_ws_bytes_ = 0;
    OWLMI_HIP(hipMalloc(&wave_ws_, list_bytes));
    wave_ws_bytes_ = list_bytes;
  }
  OWLMI_HIP(hipMemsetAsync(tie_, 0, (size_t)n, s));  // (three-word keys: no row is left to the tie pass)
  OWLMI_HIP(hipMemsetAsync(counters_, 0, kCounters * sizeof(unsigned long long), s));
  OWLMI_HIP(hipMemsetAsync(done_, 0, (size_t)n, s));
  OWLMI_HIP(hipMemsetAsync(isect_sorted_, 0, (size_t)n * sizeof(int64_t), s));
  OWLMI_HIP(hipMemsetAsync(next_level_, 0, (size_t)n * sizeof(int32_t), s));
  if (sa.d_levels && sa.phase < 2) OWLMI_HIP(hipMemsetAsync(sa.d_levels, 0xff, (size_t)n * sizeof(int32_t), s));  // (phases 2, 3 complete an earlier call's rows)
  reset_stat_stripes(s);
  OWLMI_HIP(hipEventRecord(ev_a_, s));
  {
    BigKey *lists = (BigKey *)wave_ws_;
    int ch = chunks;
    a.grab = 1;
    void *kargs[] = {(void *)&a, (void *)&lists, (void *)&ch};
    OWLMI_HIP(hipLaunchKernel(entry, dim3(blocks), dim3(kTeamBlock), kargs, 0, s));
  }
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipEventRecord(ev_b_, s));
  fetch_stat_stripes(s);
  OWLMI_HIP(hipStreamSynchronize(s));
  fold_stat_stripes(false);
  float ms = 0;
  OWLMI_HIP(hipEventElapsedTime(&ms, ev_a_, ev_b_));
  if (h_counters_[8]) throw ArgError{TKNN_E_UNSUPPORTED, "k > 64: a query's walk outgrew its stack (a pyramid of more than six levels?)"};
  tknnSolveInfo mine;
  std::memset(&mine, 0, sizeof mine);
  mine.rounds = (int)h_counters_[1];
  mine.node_tests = (int64_t)h_counters_[2];
  mine.point_tests = (int64_t)h_counters_[3];
  mine.total_intersections = (int64_t)h_counters_[4];
  mine.total_active_rounds = (int64_t)h_counters_[6];
  mine.unfinished = (int64_t)h_counters_[7];
  mine.solve_ms = ms;
  mine.dominant_kernel_ms = ms;
  mine.dominant_kernel_launches = 1;
  mine.kernel_used = TKNN_KERNEL_TEAM;
  mine.list_capacity = chunks * 16;
  float radius = sa.start_radius;
  for (int t = 1; t < mine.rounds; t++) radius *= 2;
  mine.final_radius = radius;
  if (mine.unfinished && !sa.allow_unfinished) throw RoundsExceeded{};
  ties_early_ = true;  // nothing flagged, nothing to redo
  early_tie_rows_ = early_tie_left_ = 0;
  early_tie_ms_ = 0.f;
  if (info) *info = mine;
}


void Engine::launch_tie_fix(const SolveArgs &sa, const int32_t *slots, int32_t nslots, int blocks, hipStream_t s, const int32_t *d_slot_count,
                            int64_t expected_rows) {
  TeamArgs a;
  std::memset(&a, 0, sizeof a);
  a.bvh = bvh_.view();
  a.halo = halo_view();
  a.wide[0] = bvh_.wide_view();
  if (halo_count() > 0) a.wide[1] = halo_.wide_view();
  a.start_radius = sa.start_radius;
  a.start_radii = sa.d_start_radii;
  a.k = sa.k;
  a.out_idx = sa.d_idx;
  a.out_dist = sa.d_dist;
  a.out_fb = sa.d_fb;
  a.tie = tie_;
  a.tie_list = tie_list_;
  a.slot_count = d_slot_count;
  // (10 M taxi-like points, k = 10 / 24, 0.78 / 2.3 M rows, 24 workgroups per CU: turns of 4 / 8 / 12 / 24 / 32 slots 2.55 / 1.42 / 1.08 /
  // 0.89 / 0.94 ms and 7.2 / 3.8 / 2.6 / 1.81 / 1.85 ms; 64 and more slots a turn: the waves' own chains of loads show, 2.8 ms and up)
  a.grab = grab_for(expected_rows, blocks, 6, 6);
  a.row_slot = getenv("TKNN_TIE_LOOK") && !strcmp(getenv("TKNN_TIE_LOOK"), "0") ? nullptr : bvh_.row_slot_device();  // (A/B switch)
  a.counters = counters_;
  using FixEntry = void (*)(TeamArgs, const int32_t *, int32_t);
  static const FixEntry entries[2][4] = {{tie_fix_kernel<false, 1>, tie_fix_kernel<false, 2>, tie_fix_kernel<false, 3>, tie_fix_kernel<false, 4>},
                                         {tie_fix_kernel<true, 1>, tie_fix_kernel<true, 2>, tie_fix_kernel<true, 3>, tie_fix_kernel<true, 4>}};
  const FixEntry entry = entries[halo_count() > 0 ? 1 : 0][nreg_for(sa.k) - 1];
  void *kargs[] = {(void *)&a, (void *)&slots, (void *)&nslots};
  OWLMI_HIP(hipLaunchKernel((const void *)entry, dim3(blocks), dim3(kTeamBlock), kargs, 0, s));
}

void Engine::fix_ties(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s) {
  if (ties_early_) {  // solve_team's own launch has seen them all
    if (early_tie_rows_ && getenv("TKNN_VERBOSE"))
      fprintf(stderr, "[ties] %lld rows redone in the reference's tie order: %.3f ms, %lld left\n", (long long)early_tie_rows_, early_tie_ms_, (long long)early_tie_left_);
    if (info) {
      info->tie_rows = early_tie_rows_;
      info->tie_rows_left = early_tie_left_;
      info->tie_ms = early_tie_ms_;
      info->solve_ms += early_tie_ms_;
    }
    return;
  }
  const int64_t n = bvh_.size();
  hipDeviceProp_t prop;
  OWLMI_HIP(hipGetDeviceProperties(&prop, device_));
  auto launch = [&](int blocks, const int32_t *slots, int32_t nslots, const int32_t *d_len = nullptr, int64_t expected = 0) {
    launch_tie_fix(sa, slots, nslots, blocks, s, d_len, expected);
  };
  OWLMI_HIP(hipMemsetAsync(counters_ + kTieCounter + 1, 0, 3 * sizeof(unsigned long long), s));  // work cursor, rows left
  // First go: the kernels' own list, count read on the device -- no host round trip before the launch;
  // the usual handful of rows (or none) costs one small launch behind the solve.
  OWLMI_HIP(hipEventRecord(ev_a_, s));
  launch(std::min(prop.multiProcessorCount * 4, kTieListCap / 4), tie_list_, -1);  // a team per listed row
  OWLMI_HIP(hipEventRecord(ev_b_, s));
  OWLMI_HIP(hipMemcpyAsync(h_counters_, counters_ + kTieCounter, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipStreamSynchronize(s));
  float ms = 0;
  OWLMI_HIP(hipEventElapsedTime(&ms, ev_a_, ev_b_));
  const int64_t flagged = (int64_t)h_counters_[0];
  if (flagged > kTieListCap) {
    // more than the list holds (quantised coordinates, lattices): all flagged slots, compacted from tie_
    // (rows redone twice come out the same: the gate is the row's k-th distance, which no order changes)
    if (n > slot_list_cap_) {
      if (slot_list_) (void)hipFree(slot_list_);
      slot_list_ = nullptr;
      OWLMI_HIP(hipMalloc((void **)&slot_list_, ((size_t)n + 1) * sizeof(int32_t)));
      slot_list_cap_ = n;
    }
    int32_t *d_count = slot_list_ + n;
    hipcub::CountingInputIterator<int32_t> iota(0);
    hipcub::TransformInputIterator<bool, HasTie, const uint8_t *> flags(tie_, HasTie{});
    size_t tmp_bytes = 0;
    OWLMI_HIP(hipcub::DeviceSelect::Flagged(nullptr, tmp_bytes, iota, flags, slot_list_, d_count, (int)n, s));
    if (tmp_bytes > wave_ws_bytes_) {
      if (wave_ws_) (void)hipFree(wave_ws_);
      wave_ws_ = nullptr;
      OWLMI_HIP(hipMalloc(&wave_ws_, tmp_bytes));
      wave_ws_bytes_ = tmp_bytes;
    }
    OWLMI_HIP(hipcub::DeviceSelect::Flagged(wave_ws_, tmp_bytes, iota, flags, slot_list_, d_count, (int)n, s));
    OWLMI_HIP(hipMemsetAsync(counters_ + kTieCounter + 1, 0, 3 * sizeof(unsigned long long), s));
    OWLMI_HIP(hipEventRecord(ev_a_, s));
    // the list's length is read on the device (d_count): `flagged` counts flag calls, an upper bound
    launch((int)std::min<int64_t>((std::min<int64_t>(flagged, n) + 3) / 4, (int64_t)prop.multiProcessorCount * 24), slot_list_, -2, d_count,
           std::min<int64_t>(flagged, n));
    OWLMI_HIP(hipEventRecord(ev_b_, s));
    OWLMI_HIP(hipMemcpyAsync(h_counters_, counters_ + kTieCounter, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    OWLMI_HIP(hipStreamSynchronize(s));
    float again = 0;
    OWLMI_HIP(hipEventElapsedTime(&again, ev_a_, ev_b_));
    ms += again;
  }
  if (flagged && getenv("TKNN_VERBOSE"))
    fprintf(stderr, "[ties] %lld rows redone in the reference's tie order: %.3f ms, %llu left (%llu stood after a look at the written row)\n", (long long)flagged, ms,
            h_counters_[2], h_counters_[3]);
  if (info) {
    info->tie_rows = flagged;
    info->tie_rows_left = (int64_t)h_counters_[2];
    info->tie_ms = ms;
    info->solve_ms += ms;
  }
}


// How many radius levels the first gather of every packet should serve: with the average density
// of the scene, the first level at which a box is expected to hold about k/2 other points.  Only
// a work estimate -- every level is still resolved exactly.
int Engine::first_step_estimate(const SolveArgs &sa) const {
  float radius = sa.start_radius;
  for (int m = 1; m < 3; m++) {
    if (expected_box_population(radius) >= 0.5 * sa.k) return m;
    radius *= 2.0f;
  }
  return 3;
}

bool Engine::solve_team(const SolveArgs &sa, tknnSolveInfo *info, hipStream_t s) {
  const int64_t n = bvh_.size();
  if (n >= (1ll << 28)) return false;  // leaf blocks are addressed by 32-bit byte offsets; the caller takes the wave kernel
  TeamArgs a;
  a.bvh = bvh_.view();
  a.halo = halo_view();
  a.wide[0] = bvh_.wide_view();
  if (halo_count() > 0)
    a.wide[1] = halo_.wide_view();
  else
    std::memset(&a.wide[1], 0, sizeof(a.wide[1]));
  a.start_radius = sa.start_radius;
  a.start_radii = sa.d_start_radii;
  a.k = sa.k;
  a.max_rounds = sa.max_rounds;
  a.allow_unfinished = sa.allow_unfinished ? 1 : 0;
  a.first_step = sa.d_start_radii ? 1 : first_step_estimate(sa);  // (the estimate is from ONE start radius and the mean density)
  {
    // the first gather also lists the blocks of the level after its step unless the boxes of the step's last level are
    // expected to hold k others several times over (team_kernel: the same rule between later steps)
    float radius = sa.start_radius;
    for (int t = 1; t < std::min(a.first_step, kMaxStep); t++) radius *= 2.0f;
    a.first_ext = !sa.d_start_radii && expected_box_population(radius) < 2.5 * sa.k ? 1 : 0;
    if (const char *e = getenv("TKNN_TEAM_EXT")) a.first_ext = atoi(e) > 0 ? a.first_ext : -1;  // measurements: 0 = no extended gathers at all
  }
  {
    // axes along which the built points differ at all (2-D inputs carry z = 0, hostCode.cpp:115-118);
    // a halo tree may hold anything
    int dims = 0;
    for (int ax = 0; ax < 3; ax++) dims += scene_[3 + ax] > scene_[ax] ? 1 : 0;
    if (halo_count() > 0) dims = 3;
    a.tie_span = dims >= 3 ? 1.73206f : (dims == 2 ? 1.41422f : 1.00001f);
  }
  a.diag = 0;
  if (TKNN_DIAG_BUILD)
    if (const char *d = getenv("TKNN_TEAM_DIAG")) a.diag = atoi(d);
  a.ngroups = (int32_t)((n + 63) / 64);
  a.out_idx = sa.d_idx;
  a.out_dist = sa.d_dist;
  a.out_isect = sa.d_isect;
  a.out_fb = sa.d_fb;
  a.out_level = sa.d_levels;
  a.done = done_;
  a.tie = tie_;
  a.tie_list = tie_list_;
  a.slot_count = nullptr;
  a.skip = sa.phase ? boundary_ : nullptr;
  a.skip_is = sa.phase == 1 ? 1 : 0;
  a.isect_sorted = isect_sorted_;
  a.next_level = next_level_;
  a.counters = counters_;

  hipDeviceProp_t prop;
  OWLMI_HIP(hipGetDeviceProperties(&prop, device_));
  int per_cu = 2;
  static_assert(kDbStripes * 8 >= 8 * kXcdCounterStride, "the packet counters borrow the words of RT-DBSCAN's striped statistics");
  const int nreg = nreg_for(sa.k);  // list registers per lane
  const size_t lds = (size_t)kTeamBlock / 64 * (size_t)(nreg == 1 ? TeamLayout<1>::kTeamLds : (nreg == 2 ? TeamLayout<2>::kTeamLds
                                                         : (nreg == 3 ? TeamLayout<3>::kTeamLds : TeamLayout<4>::kTeamLds)));
  const bool with_halo = halo_count() > 0;
  const int nreg_at = nreg - 1;  // index into the tables of instantiations
  const char *walk_all = getenv("TKNN_TEAM_WALK_ALL");    // measurements only: k > 32 without the packet kernel
  if (sa.k > 32 && walk_all && atoi(walk_all)) {
    // (33 <= k <= 64 as it was before the packet kernel had four list registers per lane:) every
    // query goes through the team walk, four list registers per lane, from level 0
    OWLMI_HIP(hipMemsetAsync(tie_, 0, (size_t)n, s));
    OWLMI_HIP(hipMemsetAsync(counters_, 0, kCounters * sizeof(unsigned long long), s));
    OWLMI_HIP(hipMemsetAsync(done_, 0, (size_t)n, s));
    OWLMI_HIP(hipMemsetAsync(isect_sorted_, 0, (size_t)n * sizeof(int64_t), s));
    OWLMI_HIP(hipMemsetAsync(next_level_, 0, (size_t)n * sizeof(int32_t), s));
    if (sa.d_levels) OWLMI_HIP(hipMemsetAsync(sa.d_levels, 0xff, (size_t)n * sizeof(int32_t), s));
    const int walk_blocks = (int)std::min<int64_t>((n + 3) / 4, (int64_t)prop.multiProcessorCount * kWalkBlocksPerCu);
    a.grab = 1;  // (queries differ too much for longer turns: measured, see TeamArgs::grab)
    reset_stat_stripes(s);
    OWLMI_HIP(hipEventRecord(ev_a_, s));
    if (with_halo)
      hipLaunchKernelGGL((team_walk_kernel<true, 4>), dim3(walk_blocks), dim3(kTeamBlock), 0, s, a, (const int32_t *)nullptr, (int32_t)n);
    else
      hipLaunchKernelGGL((team_walk_kernel<false, 4>), dim3(walk_blocks), dim3(kTeamBlock), 0, s, a, (const int32_t *)nullptr, (int32_t)n);
    OWLMI_HIP(hipGetLastError());
    OWLMI_HIP(hipEventRecord(ev_b_, s));
    fetch_stat_stripes(s);
    OWLMI_HIP(hipStreamSynchronize(s));
    fold_stat_stripes(false);
    float ms = 0;
    OWLMI_HIP(hipEventElapsedTime(&ms, ev_a_, ev_b_));
    tknnSolveInfo mine;
    std::memset(&mine, 0, sizeof mine);
    mine.rounds = (int)h_counters_[1];
    mine.node_tests = (int64_t)h_counters_[2];
    mine.point_tests = (int64_t)h_counters_[3];
    mine.total_intersections = (int64_t)h_counters_[4];
    mine.total_active_rounds = (int64_t)h_counters_[6];
    mine.unfinished = (int64_t)h_counters_[7];
    mine.solve_ms = ms;
    mine.dominant_kernel_ms = ms;
    mine.dominant_kernel_launches = 1;
    mine.kernel_used = TKNN_KERNEL_TEAM;
    mine.list_capacity = 64;
    if (mine.unfinished && !sa.allow_unfinished) throw RoundsExceeded{};
    if (h_counters_[8]) {  // stacks exhausted: those queries' state is untouched, lane rounds take them
      tknnSolveInfo rest;
      std::memset(&rest, 0, sizeof rest);
      continue_lane(sa, 0, &rest, s);
      mine.rounds = std::max(mine.rounds, rest.rounds);
      mine.node_tests += rest.node_tests;
      mine.point_tests += rest.point_tests;
      mine.total_intersections += rest.total_intersections;
      mine.total_active_rounds += rest.total_active_rounds;
      mine.unfinished += rest.unfinished;
      mine.solve_ms += rest.solve_ms;
    }
    float radius = sa.start_radius;
    for (int t = 1; t < mine.rounds; t++) radius *= 2;
    mine.final_radius = radius;
    if (info) *info = mine;
    return true;
  }
  const bool full_list = sa.k == 16 * nreg;  // no spare list entry to see a tie with the row's last in
  using TeamEntry = void (*)(TeamArgs);
  static const TeamEntry entries[2][4][2] = {
      {{team_kernel<false, 1, false>, team_kernel<false, 1, true>}, {team_kernel<false, 2, false>, team_kernel<false, 2, true>},
       {team_kernel<false, 3, false>, team_kernel<false, 3, true>}, {team_kernel<false, 4, false>, team_kernel<false, 4, true>}},
      {{team_kernel<true, 1, false>, team_kernel<true, 1, true>}, {team_kernel<true, 2, false>, team_kernel<true, 2, true>},
       {team_kernel<true, 3, false>, team_kernel<true, 3, true>}, {team_kernel<true, 4, false>, team_kernel<true, 4, true>}}};
  const TeamEntry entry = entries[with_halo ? 1 : 0][nreg_at][full_list ? 1 : 0];
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)entry, kTeamBlock, lds) != hipSuccess) per_cu = 2;
  // (the query divides the CU's LDS by the byte; the hardware hands it out in granules: scripts/microbench/lds_granule.hip --
  // a launch of more workgroups than fit would leave the surplus waiting for a slot and then for the last packets)
  per_cu = std::max(1, std::min(per_cu, lds_workgroups_per_cu(lds)));
  if (const char *cap = getenv("TKNN_TEAM_WAVES_PER_CU"))  // measurements only: how the packet kernel's time scales with the waves in flight
    per_cu = std::max(1, std::min(per_cu, atoi(cap)));
  const int64_t want = (a.ngroups + kTeamBlock / 64 - 1) / (kTeamBlock / 64);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)prop.multiProcessorCount * per_cu));

  // one launch instead of six fills (each costs a few microseconds of its own on the stream): done = 1,
  // tie = 0, all counters 0 except [9] (min hand-over level) = ~0, levels = -1
  // (phases 2 and 3 complete the rows and levels an earlier call has begun: levels are preset once, by phase 0 or 1)
  hipLaunchKernelGGL(team_prep_kernel, dim3(prop.multiProcessorCount * 4), dim3(256), 0, s, done_, tie_, n, counters_, sa.phase >= 2 ? nullptr : sa.d_levels);
  OWLMI_HIP(hipEventRecord(ev_a_, s));
  {
    void *kargs[] = {(void *)&a};
    OWLMI_HIP(hipLaunchKernel((const void *)entry, dim3(blocks), dim3(kTeamBlock), kargs, lds, s));
  }
  OWLMI_HIP(hipGetLastError());
  OWLMI_HIP(hipEventRecord(ev_b_, s));
  // the tie pass over the rows this kernel has flagged and listed, count read on the device: launched
  // before the host knows anything, so the usual handful costs no round trip of its own
  launch_tie_fix(sa, tie_list_, -1, std::min(prop.multiProcessorCount * 4, kTieListCap / 4), s);
  OWLMI_HIP(hipEventRecord(ev_c_, s));
  static_assert(kStatBase == Engine::kStatBase && kStatStripes == Engine::kStatStripes && kStatStride == Engine::kStatStride, "one layout");
  fetch_stat_stripes(s);
  OWLMI_HIP(hipMemcpyAsync(h_counters_ + 10, counters_ + kTieCounter, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  OWLMI_HIP(hipStreamSynchronize(s));
  h_counters_[0] = 0;
  fold_stat_stripes(true);
  float ms = 0;
  OWLMI_HIP(hipEventElapsedTime(&ms, ev_a_, ev_b_));
  if (h_counters_[8] == 0 && h_counters_[10] <= (unsigned long long)kTieListCap) {  // nothing handed over, every flagged row listed
    ties_early_ = true;
    early_tie_rows_ = (int64_t)h_counters_[10];
    early_tie_left_ = (int64_t)h_counters_[12];
    OWLMI_HIP(hipEventElapsedTime(&early_tie_ms_, ev_b_, ev_c_));
  }
#if TKNN_DIAG_BUILD
  if (getenv("TKNN_TEAM_DIAG")) {
    unsigned long long t[5];
    OWLMI_HIP(hipMemcpy(t, counters_ + 10, sizeof t, hipMemcpyDeviceToHost));
    const double tot = (double)(t[0] + t[1] + t[2] + t[3] + t[4]);
    unsigned long long wave_steps = 0;
    OWLMI_HIP(hipMemcpy(&wave_steps, counters_ + 15, sizeof wave_steps, hipMemcpyDeviceToHost));
    fprintf(stderr, "[team diag] block steps: %.3g wave steps x 4 teams for %.3g listed blocks (lockstep efficiency %.1f%%)\n",
            (double)wave_steps, (double)h_counters_[3] / LBVH_BLOCK, 100.0 * ((double)h_counters_[3] / LBVH_BLOCK) / (4.0 * (double)wave_steps));
    if (a.diag & 16) {
      unsigned long long g[2];
      OWLMI_HIP(hipMemcpy(g, counters_ + 26, sizeof g, hipMemcpyDeviceToHost));
      fprintf(stderr, "[team diag] inserts: %.4g candidates passed the gate after the sorted first block (%.2f per query), in %.4g lock-step rounds of the wave (%.2f per query)\n",
              (double)g[0], (double)g[0] / (double)n, (double)g[1], (double)g[1] / (double)n);
    }
    if (a.diag & 128) {
      unsigned long long g[12];
      OWLMI_HIP(hipMemcpy(g, counters_ + 28, sizeof g, hipMemcpyDeviceToHost));
      for (int pass = 0; pass < 2; pass++) {
        const unsigned long long *t4 = g + 8 * pass;
        const double tt = (double)(t4[0] + t4[1] + t4[2] + t4[3]);
        fprintf(stderr, "[team diag] %s pass, wave time: %.3g cycles in all; set-up %.1f%%  first group %.1f%%  other groups %.1f%%  epilogue %.1f%%\n",
                pass ? "SELECT" : "COUNT", tt, 100 * t4[0] / tt, 100 * t4[1] / tt, 100 * t4[2] / tt, 100 * t4[3] / tt);
      }
    }
    if (a.diag & 32) {
      unsigned long long g[2];
      OWLMI_HIP(hipMemcpy(g, counters_ + 24, sizeof g, hipMemcpyDeviceToHost));
      fprintf(stderr, "[team diag] gather: %.3g leaf blocks tested against the 64 queries, %.3g of them listed (%.1f%%)\n", (double)g[0], (double)g[1], 100.0 * (double)g[1] / (double)g[0]);
    }
    fprintf(stderr, "[team diag] mean busy time per wave %.2f ms of %.2f ms kernel time (%d waves; s_memtime at 100 MHz)\n",
            tot / 1e8 * 1e3 / blocks, ms, blocks);
    fprintf(stderr, "[team diag] wave-time shares: records %.1f%%  gather %.1f%%  count %.1f%%  select %.1f%%  rest %.1f%%\n",
            100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot, 100 * t[4] / tot);
  }
#endif
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
    info->kernel_used = TKNN_KERNEL_TEAM;
    info->list_capacity = 16 * nreg;
    info->unfinished = (int64_t)h_counters_[7];
  }
  const unsigned long long handed = h_counters_[8];
  if (handed) {
    // Stragglers whose candidate lists outgrew the LDS lists (outliers of a clustered set, whose boxes
    // grow over whole clusters; dense duplicates): team_walk_kernel, one query per team, from the level
    // at which each was handed over; what exhausts its stack goes on to per-round lane launches.  A
    // quarter or more of all queries (a start radius far too large for the density: hundreds of
    // candidates per query from level 0 on): the wave-packet kernel, which streams candidate sets of
    // any size, re-solves just those queries.
    tknnSolveInfo tail;
    std::memset(&tail, 0, sizeof tail);
    const char *force = getenv("TKNN_TEAM_TAIL");  // "walk" / "lane" / "wave": tests and measurements only
    // (k > 32: the team walk beats the wave kernel's 64-entry register lists on any share, 10 M uniform
    // points from level 0: 137 against 223 ms at k = 50)
    const bool by_wave = !sa.d_start_radii && wave_kernel_available() &&
                         (force ? !strcmp(force, "wave") : (handed * 4ull >= (unsigned long long)n && sa.k <= 32));
    const bool by_walk = !by_wave && !(force && !strcmp(force, "lane"));
    const int first_handover_level = (int)h_counters_[9];
    if (by_wave) {
      solve_wave(sa, &tail, s, /*only_unfinished=*/true);
    } else if (by_walk) {
      // one query per team (team_walk_kernel): the stragglers' sorted slots as a compact ascending list
      if (n > slot_list_cap_) {
        if (slot_list_) (void)hipFree(slot_list_);
        slot_list_ = nullptr;
        OWLMI_HIP(hipMalloc((void **)&slot_list_, ((size_t)n + 1) * sizeof(int32_t)));
        slot_list_cap_ = n;
      }
      int32_t *d_count = slot_list_ + n;
      hipcub::CountingInputIterator<int32_t> iota(0);
      hipcub::TransformInputIterator<bool, NotDone, const uint8_t *> flags(done_, NotDone{});
      size_t tmp_bytes = 0;
      OWLMI_HIP(hipcub::DeviceSelect::Flagged(nullptr, tmp_bytes, iota, flags, slot_list_, d_count, (int)n, s));
      if (tmp_bytes > wave_ws_bytes_) {
        if (wave_ws_) (void)hipFree(wave_ws_);
        wave_ws_ = nullptr;
        OWLMI_HIP(hipMalloc(&wave_ws_, tmp_bytes));
        wave_ws_bytes_ = tmp_bytes;
      }
      OWLMI_HIP(hipcub::DeviceSelect::Flagged(wave_ws_, tmp_bytes, iota, flags, slot_list_, d_count, (int)n, s));
      OWLMI_HIP(hipMemsetAsync(counters_, 0, 16 * sizeof(unsigned long long), s));
      reset_stat_stripes(s);
      const int walk_blocks = (int)std::min<int64_t>((int64_t)(handed + 3) / 4, (int64_t)prop.multiProcessorCount * kWalkBlocksPerCu);
      OWLMI_HIP(hipEventRecord(ev_a_, s));
      {
        using WalkEntry = void (*)(TeamArgs, const int32_t *, int32_t);
        static const WalkEntry walks[2][4] = {{team_walk_kernel<false, 1>, team_walk_kernel<false, 2>, team_walk_kernel<false, 3>, team_walk_kernel<false, 4>},
                                              {team_walk_kernel<true, 1>, team_walk_kernel<true, 2>, team_walk_kernel<true, 3>, team_walk_kernel<true, 4>}};
        const int32_t *slots = slot_list_;
        int32_t nslots = (int32_t)handed;
        a.grab = 1;
        void *kargs[] = {(void *)&a, (void *)&slots, (void *)&nslots};
        OWLMI_HIP(hipLaunchKernel((const void *)walks[with_halo ? 1 : 0][nreg_at], dim3(walk_blocks), dim3(kTeamBlock), kargs, 0, s));
      }
      OWLMI_HIP(hipGetLastError());
      OWLMI_HIP(hipEventRecord(ev_b_, s));
      fetch_stat_stripes(s);
      OWLMI_HIP(hipStreamSynchronize(s));
      fold_stat_stripes(false);
      float walk_ms = 0;
      OWLMI_HIP(hipEventElapsedTime(&walk_ms, ev_a_, ev_b_));
      tail.rounds = (int)h_counters_[1];
      tail.node_tests = (int64_t)h_counters_[2];
      tail.point_tests = (int64_t)h_counters_[3];
      tail.total_intersections = (int64_t)h_counters_[4];
      tail.total_active_rounds = (int64_t)h_counters_[6];
      tail.unfinished = (int64_t)h_counters_[7];
      tail.solve_ms = walk_ms;
      tail.dominant_kernel_launches = 1;
      const unsigned long long left = h_counters_[8];  // stack exhausted: state untouched, lane rounds take them
      if (tail.unfinished && !sa.allow_unfinished) throw RoundsExceeded{};
      if (left && sa.d_start_radii)
        throw ArgError{TKNN_E_UNSUPPORTED, "per-query start radii: a query's candidate walk outgrew the team walk's stack (the lane rounds that take over otherwise use one radius per launch)"};
      if (left) {
        tknnSolveInfo rest;
        std::memset(&rest, 0, sizeof rest);
        continue_lane(sa, first_handover_level, &rest, s);
        tail.rounds = std::max(tail.rounds, rest.rounds);
        tail.node_tests += rest.node_tests;
        tail.point_tests += rest.point_tests;
        tail.total_intersections += rest.total_intersections;
        tail.total_active_rounds += rest.total_active_rounds;
        tail.unfinished += rest.unfinished;
        tail.solve_ms += rest.solve_ms;
        tail.dominant_kernel_launches += rest.dominant_kernel_launches;
      }
    } else {
      if (sa.d_start_radii) throw ArgError{TKNN_E_UNSUPPORTED, "per-query start radii are served by the team kernels only (TKNN_TEAM_TAIL=lane)"};
      continue_lane(sa, first_handover_level, &tail, s);
    }
    if (getenv("TKNN_VERBOSE"))
      fprintf(stderr, "[team] %llu of %lld queries handed over from level %d on: team kernel %.2f ms, %s %.2f ms (%d launches)\n",
              handed, (long long)n, first_handover_level, ms, by_wave ? "wave kernel" : (by_walk ? "team walk" : "lane rounds"), tail.solve_ms,
              tail.dominant_kernel_launches);
    if (info) {
      info->rounds = std::max(info->rounds, tail.rounds);
      float radius = sa.start_radius;
      for (int t = 1; t < info->rounds; t++) radius *= 2;
      info->final_radius = radius;
      info->node_tests += tail.node_tests;
      info->point_tests += tail.point_tests;
      info->total_intersections += tail.total_intersections;
      info->total_active_rounds += tail.total_active_rounds;
      info->solve_ms += tail.solve_ms;
      info->unfinished += tail.unfinished;
    }
  }
  return true;
}

}  // namespace owlmi
