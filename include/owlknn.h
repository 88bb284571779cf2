/*
 * owlknn.h -- C-ABI of the MI355X-native TrueKNN engine (libowl_mi355x.so).
 *
 * This is the hot path of vani-nag/OWLRayTracing's samples/s01-trueknn, taken as ONE unit:
 *
 *   reference                                               here
 *   ------------------------------------------------------  --------------------------------
 *   owlDeviceBufferCreate(Sphere[n])  hostCode.cpp:165-166   tknnBuild: points already in HBM
 *   owlUserGeomGroupCreate + owlGroupBuildAccel (+ instance  tknnBuild: HIP LBVH (Morton sort +
 *     group)  hostCode.cpp:201-206 -> bounds program           radix tree) over the centres; the
 *     deviceCode.cu:38-56 + optixAccelBuild                    radius is applied at test time
 *     (owl/UserGeomGroup.cpp:161-217)
 *   while(!foundKNN){ owlLaunch2D; scan fb; radius*=2;       tknnSolve: the whole radius-doubling
 *     owlGeomSet1f; owlGroupRefitAccel x2 }                    solve, rounds resolved on the device
 *     hostCode.cpp:285-340 around __raygen__rayGen /
 *     __intersection__Spheres  deviceCode.cu:62-153
 *   frameBuffer of Neigh[n*k]  GeomTypes.h:22-28             d_fb (same 24-byte records) and/or
 *                                                            compact idx/dist/intersections
 *
 * The same library also exports the reference's own owl* entry points (include/owl/owl_host.h),
 * which run user raygen / intersect / bounds programs over the same LBVH; the tknn* calls are the
 * fused form of the loop above and give bit-identical rows (modulo the order of exact ties inside
 * one round, which the reference leaves to traversal order; tknn* orders ties by the round in which a
 * candidate was first seen -- the reference's lists persist over rounds -- and then by index).
 *
 * Conventions: plain C, pointers and sizes only.  `d_` pointers are device (HBM) addresses valid
 * on the engine's device.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Every call returns 0 on success or a negative TKNN_E_* code; tknnLastError() gives the text.
 * Nothing here falls back to the CPU.
 */
#ifndef OWLKNN_H
#define OWLKNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TKNN_API __attribute__((visibility("default")))

typedef struct tknnEngine_t *tknnEngine;

/* GeomTypes.h:22-28: 24-byte result record; only slot [q*k+0] carries numNeighbors/intersections */
typedef struct {
  int32_t ind;
  float dist;
  int32_t numNeighbors;
  int32_t pad_;
  int64_t intersections;
} tknnNeigh;

enum {
  TKNN_OK = 0,
  TKNN_E_ARG = -1,      /* bad argument (null pointer, n <= k, radius not finite-positive, ...)  */
  TKNN_E_HIP = -2,      /* a HIP call failed; no device, out of memory, launch failure           */
  TKNN_E_STATE = -3,    /* call order: solve before build, ...                                   */
  TKNN_E_ROUNDS = -4,   /* max_rounds reached with unfinished queries (reference: endless loop)  */
  TKNN_E_UNSUPPORTED = -5 /* k above TKNN_MAX_K; a kernel asked for by name that does not serve this k  */
};

/* The reference takes any k from its command line (samples/s01-trueknn/hostCode.cpp:111) and keeps the lists in global
 * memory (deviceCode.cu:77-134).  Here k <= 64 is served from registers (TKNN_KERNEL_LANE / _WAVE / _TEAM); 64 < k <=
 * TKNN_MAX_K by the team walk with the lists in memory (TKNN_KERNEL_AUTO or _TEAM). */
#define TKNN_MAX_K 1024
#define TKNN_MAX_K_REGISTERS 64

/* which traversal kernel tknnSolve uses */
enum {
  TKNN_KERNEL_AUTO = 0,
  TKNN_KERNEL_LANE = 1,  /* one query per lane, stackless rope traversal, one launch per round    */
  TKNN_KERNEL_WAVE = 2,  /* one 64-query packet per wave, persistent, all rounds in one launch    */
  TKNN_KERNEL_TEAM = 3   /* 16-lane teams, lanes = candidates of one query's leaf blocks           */
};

typedef struct {
  int32_t rounds;              /* radius levels needed = rounds of the reference loop             */
  float final_radius;          /* radius of the last round                                        */
  int64_t total_intersections; /* sum of Neigh.intersections = intersection-program calls         */
  int64_t node_tests;          /* box tests against BVH nodes (wave kernel: per packet)           */
  int64_t point_tests;         /* exact point-in-box tests executed (>= total_intersections)      */
  int64_t total_active_rounds; /* sum over queries of the rounds in which the query traced a ray  */
  float solve_ms;              /* device time of the traversal launches, HIP events on `stream`   */
  float dominant_kernel_ms;    /* average duration of one launch of the dominant kernel           */
  int32_t dominant_kernel_launches;
  int32_t kernel_used;         /* TKNN_KERNEL_*                                                   */
  int32_t list_capacity;       /* register k-list size the kernel was instantiated with           */
  int64_t unfinished;          /* queries left without k neighbours (only with allow_unfinished)  */
  int64_t tie_rows;            /* rows with bit-identical fp32 distances among their k + 1 best,   */
                               /* redone in the reference's tie order (first round, then index;    */
                               /* deviceCode.cu:77-85: lists persist over rounds)                  */
  int64_t tie_rows_left;       /* ... of which kept (dist, index) order: walk stack exhausted     */
  float tie_ms;                /* device time of that pass (included in solve_ms)                 */
  int32_t reserved_;
} tknnSolveInfo;

typedef struct {
  float build_ms;        /* device time of the LBVH build                                         */
  int64_t device_bytes;  /* HBM held by the tree                                                  */
  int32_t n;
} tknnBuildInfo;

TKNN_API const char *tknnLastError(void);
TKNN_API int tknnDeviceCount(void);

/* engine on the current HIP device (hipSetDevice before the call picks it) */
TKNN_API int tknnCreate(tknnEngine *out);
TKNN_API void tknnDestroy(tknnEngine e);

/* Build the LBVH over n points.  d_xyz: n packed fp32 triples (the Sphere buffer, 12 B each; 2-D
 * data carries z = 0 as in hostCode.cpp:115-118).  The engine keeps its own Morton-ordered copy;
 * d_xyz may be freed afterwards. */
TKNN_API int tknnBuild(tknnEngine e, const float *d_xyz, int64_t n, tknnBuildInfo *info, void *stream);

/* Solve TrueKNN for every point (queries = points, deviceCode.cu:140-153).
 *   k, start_radius   as argv[5], argv[4] of the sample; n > k and 0 < start_radius < inf required
 *   max_rounds        give up (TKNN_E_ROUNDS) after this many radius levels; <= 0 means 64, more than 127 means 127
 *   d_idx, d_dist     n*k each, row q = neighbours of point q (caller's index), ascending
 *                     (dist, index); either may be NULL
 *   d_intersections   n, Neigh.intersections of slot 0 of each row; may be NULL
 *   d_fb              n*k tknnNeigh records in the state the reference's frameBuffer has after its
 *                     last round; may be NULL
 */
TKNN_API int tknnSolve(tknnEngine e, int k, float start_radius, int kernel, int max_rounds,
                       int32_t *d_idx, float *d_dist, int64_t *d_intersections, tknnNeigh *d_fb,
                       tknnSolveInfo *info, void *stream);

/* ---- sharded use: one engine per GPU owns a tile of a larger point set ----
 * tknnBuildIds: like tknnBuild, but point i is reported as d_ids[i] (a global index) in neighbour
 *   lists and excluded as "self" by that id; rows are still addressed by the local position i.
 * tknnSetHalo: a second, read-only point set (border points received from neighbouring tiles, with
 *   their global ids) that every query also searches; m = 0 removes it.
 * tknnSolveEx: tknnSolve with options: d_levels (n, may be NULL) receives the 0-based radius level at
 *   which each query finished, or -1; with allow_unfinished != 0 reaching max_rounds is not an
 *   error: unfinished queries keep level -1, their rows are not written and info->unfinished counts
 *   them (the caller widens the halo and solves again).
 * tknnHaloSelect: the send side of the exchange.  d_boxes: nboxes x {lo xyz, hi xyz} fp32 closed
 *   boxes (the peers' tile cells widened by the halo radius, rounded outward by the caller);
 *   d_box_peer: the peer (0 <= peer < npeers <= 64) each box belongs to.  Count pass (d_rows NULL):
 *   d_counts[npeers] = my points inside at least one box of each peer.  Write pass: d_rows receives
 *   16-byte rows {x, y, z, id bits}, peer p's rows contiguous from row d_offsets[p] (the caller's
 *   exclusive scan of the counts); the order inside a segment is unspecified. */
typedef struct {
  int32_t k;
  float start_radius;
  int32_t kernel;
  int32_t max_rounds;
  int32_t allow_unfinished;
  int32_t phase; /* 0: every query.  1: only the queries NO point of another tile can reach within the radius the last
                    tknnHaloSelect count pass was given boxes for ("interior"), searched in the own tree alone -- the call may
                    run on its own stream and host thread while the halo is exchanged and tknnSetHalo builds its tree.
                    2: the other ("boundary") queries, own + halo tree; rows and levels of a phase-1 call are kept.
                    3: only the queries an earlier call with allow_unfinished left without a row (d_levels[row] == -1; d_levels
                    required), from level 0 over own + halo tree; rows and levels of the others are kept -- the sharded
                    driver's straggler rounds: the halo has been widened by the shell the next radius level needs.
                    Team kernels only (TKNN_KERNEL_AUTO / _TEAM; any k up to TKNN_MAX_K). */
  int32_t *d_idx;
  float *d_dist;
  int64_t *d_intersections;
  tknnNeigh *d_fb;
  int32_t *d_levels;
  const float *d_start_radii; /* NULL, or n floats: row q starts with d_start_radii[q] instead of start_radius and doubles from
                                 there -- a per-query radius schedule (opt-in: the reference has ONE
                                 radius, samples/s01-trueknn/hostCode.cpp:185,325).  Row q is what the reference's loop gives
                                 for q when started at that radius; every value must be finite and > 0.  Team kernels only
                                 (any k up to TKNN_MAX_K), no halo tree.  info->final_radius then refers to start_radius. */
} tknnSolveOptions;
TKNN_API int tknnBuildIds(tknnEngine e, const float *d_xyz, const int32_t *d_ids, int64_t n,
                          tknnBuildInfo *info, void *stream);
TKNN_API int tknnSetHalo(tknnEngine e, const float *d_xyz, const int32_t *d_ids, int64_t m, void *stream);
TKNN_API int tknnSolveEx(tknnEngine e, const tknnSolveOptions *options, tknnSolveInfo *info, void *stream);
TKNN_API int tknnHaloSelect(tknnEngine e, const float *d_boxes, const int32_t *d_box_peer, int32_t nboxes,
                            int32_t npeers, int64_t *d_counts, const int64_t *d_offsets, float *d_rows,
                            void *stream);

/* The same in ONE pass, for a caller that knows how many rows each peer's segment may hold (both ends of a pair agree on the
 * size of their message from the pair's last exchange): rows are written at d_offsets[peer] .. + d_caps[peer], rows that do not fit
 * are dropped, d_counts[peer] receives the exact number of rows selected (> d_caps[peer]: the caller must fall back to
 * tknnHaloSelect for that peer).  No count pass, no host round trip before the rows exist.  Marks the boundary queries like the
 * count pass of tknnHaloSelect. */
TKNN_API int tknnHaloSelectFixed(tknnEngine e, const float *d_boxes, const int32_t *d_box_peer, int32_t nboxes, int32_t npeers,
                                 const int64_t *d_caps, const int64_t *d_offsets, float *d_rows, int64_t *d_counts, void *stream);

/* ---- exact kNN on request ---------------------------------------------------
 * tknnSolve reproduces the reference, whose rows are box-candidate kNN, not exact kNN: a query
 * that finished with box half-width r_q never saw points outside that box, although its k-th
 * distance d_k may exceed r_q (15-20 % of rows on uniform data).  tknnRepairExact rewrites exactly
 * those rows (d_k > r_q) with the true k nearest neighbours, same (dist, index) order, by one more
 * traversal with radius d_k.  Opt-in post-processing: it is never applied by tknnSolve, so parity
 * with the reference is unaffected.  Inputs: the rows and d_levels of a tknnSolveEx call and the
 * start radius it used; intersections / frameBuffer images are not touched. */
TKNN_API int tknnRepairExact(tknnEngine e, int k, float start_radius, const int32_t *d_levels,
                             int32_t *d_idx, float *d_dist, int64_t *repaired, void *stream);

/* ---- RT-DBSCAN over the same tree -------------------------------------
 * The reference tree holds no RT-DBSCAN source (README.md:8-9 mentions the method only), so the
 * semantics are this build's own spec (oracle/dbscan_oracle.c): N(p) = {q : dist <= eps} with p
 * included and the TrueKNN fp32 distance arithmetic; core iff |N(p)| >= min_pts; clusters =
 * components of core points, numbered by ascending smallest core index; a border point joins the
 * lowest-numbered adjacent cluster; noise = -1 (sklearn.cluster.DBSCAN's labelling).
 *   d_labels n int32 (required) | d_core n uint8 (may be NULL) | d_counts n int32 |N(p)| (may be
 *   NULL; asking for it disables the early exit of the core test).  Call tknnBuild first. */
typedef struct {
  int32_t clusters;
  float solve_ms;    /* HIP events around all launches of the call */
  float core_ms;     /* the three traversal kernels (events on the launch stream): core flags, */
  float union_ms;    /* unions of neighbouring groups of core points: all launches of the union kernel, */
  float label_ms;    /* labels of border points (tknnDbscanAssign: its one traversal)         */
  int32_t union_launches; /* launches union_ms covers (2: groups that nearly touch, then the rest) */
  int64_t node_tests;         /* tree-node box tests over the three traversals */
  int64_t point_tests;        /* points whose distance to a query was computed (12 algorithmic bytes each) */
  int64_t core_point_tests;   /* ... per traversal */
  int64_t union_point_tests;
  int64_t label_point_tests;
  int64_t union_node_tests;   /* node boxes (32 bytes each) the union kernel's walks looked at */
  int64_t groups;             /* maximal tight nodes with a core point: the vertices of the union pass */
} tknnDbscanInfo;
TKNN_API int tknnDbscan(tknnEngine e, float eps, int min_pts, int32_t *d_labels, uint8_t *d_core,
                        int32_t *d_counts, tknnDbscanInfo *info, void *stream);
/* Labels, core flags and counts are indexed by ROW (position of the point in the buffer given to
 * tknnBuild / tknnBuildIds), also for engines built with ids; clusters are numbered by ascending
 * smallest core row.
 * tknnDbscanAssign: the last step alone, with labels decided by the caller (sharded use: labels
 * agreed between tiles).  d_core_label[row] >= 0: the point is core and has that label; < 0: it is
 * not core.  d_labels[row] = that label for core points, the smallest label among the core points
 * within eps for the others, -1 if there is none.  Core points within eps of each other must carry
 * the same label (true of any DBSCAN clustering). */
TKNN_API int tknnDbscanAssign(tknnEngine e, float eps, const int32_t *d_core_label, int32_t *d_labels,
                              tknnDbscanInfo *info, void *stream);

/* ---- RT-DBSCAN with an auto-grown eps (BASELINE.json configs[4]) ----------------------------------------------
 * No counterpart in the reference (it has no RT-DBSCAN source; BASELINE.md section 4: "spec TBD"), so the rule is this
 * build's own spec (oracle/dbscan_oracle.c, dbref_dbscan_auto), built on the reference's one growth rule, the radius
 * doubling of samples/s01-trueknn/hostCode.cpp:310-330: eps starts at eps0 and doubles (fp32) until at most
 * floor(max_noise * n) points are noise; the labelling returned is tknnDbscan's at that eps.  Growth rounds only
 * count noise (core flags are kept from round to round: neighbourhoods only grow), one full clustering follows.
 * TKNN_E_ROUNDS if max_rounds rounds do not get there (labels then hold the last round's). */
typedef struct {
  tknnDbscanInfo last; /* the full clustering at the final eps */
  int32_t rounds;      /* growth rounds run (1 = eps0 was enough) */
  float eps;           /* the final eps = eps0 * 2^(rounds-1) */
  int64_t noise;       /* points labelled -1 at the final eps */
  float probe_ms;      /* all growth rounds (HIP events) */
  int32_t pad_;
} tknnDbscanAutoInfo;
TKNN_API int tknnDbscanAuto(tknnEngine e, float eps0, int min_pts, double max_noise, int max_rounds, int32_t *d_labels,
                            uint8_t *d_core, tknnDbscanAutoInfo *info, void *stream);
/* One growth round's question alone, for a caller that runs the loop itself (the sharded driver: the tiles' halos grow with
 * eps): d_noise[row] = 1 if the point would be labelled -1 by tknnDbscan(eps, min_pts) -- it is not core and has no core
 * point within eps --, 0 otherwise; no clusters are built.  *noise_count (may be NULL) = how many. */
/* Sharded RT-DBSCAN (owlraytracing_amd/distributed.py): d_out[d_segment[i]] = min(d_out[d_segment[i]], d_value[i])
 * over the n elements with d_segment[i] >= 0; d_out is preset by the caller.  The one per-point step of the label
 * propagation over tiles: the smallest global id of every local cluster.  On the engine's device. */
TKNN_API int tknnSegmentMin(tknnEngine e, const int32_t *d_segment, const int64_t *d_value, int64_t n, int64_t *d_out, void *stream);

TKNN_API int tknnDbscanNoise(tknnEngine e, float eps, int min_pts, uint8_t *d_noise, int64_t *noise_count, void *stream);

/* Test / debug export of the tree to host memory (any pointer may be NULL):
 *   nodes      (n-1) x 8 dwords {lo[3], split, hi[3], other}   (include/owl/lbvh_device.h)
 *   rope_node  n-1, rope_leaf n, prim_id n (caller index of sorted slot)            */
TKNN_API int tknnExportTree(tknnEngine e, void *nodes, int32_t *rope_node, int32_t *rope_leaf,
                            int32_t *prim_id, void *stream);

/* Test hook: the builder's two side tables of a point tree with n > 1 (host buffers).
 *   split_owner  n-1: the internal node that splits its range after sorted position s -- an internal node i is a left
 *                child iff i is the LAST position of its range (parent = split_owner[i]), else a right child (parent =
 *                split_owner[i-1]); RT-DBSCAN climbs with it (owlraytracing_amd/csrc/dbscan.hip, db_uniform_kernel)
 *   block_paths  ceil(n/64) x 5: per block of 64 sorted slots the deepest internal node whose range holds the whole block
 *                (last word) and its four nearest ancestors, farthest first, the root where the path is shorter            */
TKNN_API int tknnExportTreeTables(tknnEngine e, int32_t *split_owner, int32_t *block_paths, void *stream);

/* Test hook for the wave kernel's candidate test.  For each pair (q[i], r[i]) writes lo[i], hi[i]
 * such that, for every fp32 c,   lo <= c <= hi   <=>   fl(c - r) <= q <= fl(c + r)
 * (the box the bounds program of deviceCode.cu:38-56 writes, tested against the query point). */
TKNN_API int tknnDebugThresholds(const float *d_q, const float *d_r, int64_t n, float *d_lo,
                                 float *d_hi, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OWLKNN_H */
