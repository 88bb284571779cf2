/*
 * trueknn_oracle.c -- CPU restatement of the TrueKNN hot path of vani-nag/OWLRayTracing.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported CPU baseline.  The shipped path is the HIP engine.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, known-answer tests or
 * fixtures for this path (its tests/ are three OptiX render smoke tests) and cannot
 * be built here (needs nvcc + the closed OptiX 7 SDK; BVH build and traversal live
 * inside the NVIDIA driver).  This file therefore restates the algorithm from the
 * reference's source text; it is cross-checked by an independent numpy/scipy
 * restatement (oracle/trueknn_numpy.py), not by reference output.
 *
 * What is restated (paths relative to the reference tree):
 *   samples/s01-trueknn/GeomTypes.h:22-28     struct Neigh {ind, dist, numNeighbors, intersections}
 *   samples/s01-trueknn/hostCode.cpp:115-130  2-D input => z = 0; result rows initialised to
 *                                             {-1, FLOAT_MAX, k, 0} (k slots per query)
 *   samples/s01-trueknn/deviceCode.cu:38-56   bounds program: box = [c - rad, c + rad] (fp32),
 *                                             built with box3f::extend = componentwise min/max
 *                                             (owl/include/owl/common/math/box.h:128-132)
 *   samples/s01-trueknn/deviceCode.cu:140-153 raygen: query q active iff fb[q*k].numNeighbors > 0;
 *                                             ray = (origin c_q, dir (0,0,1), tmin 0, tmax 1e-16),
 *                                             i.e. a point query
 *   samples/s01-trueknn/deviceCode.cu:62-138  intersection program (tk_intersect below is a
 *                                             line-by-line restatement, same comparison operators)
 *   samples/s01-trueknn/hostCode.cpp:285-340  round loop: launch; if any fb[j*k].numNeighbors > 0
 *                                             then radius *= 2 (fp32), refit, relaunch
 *
 * Decisions the reference leaves to closed hardware, fixed here and mirrored by the HIP engine:
 *   (1) candidate test: primitive p is reported for query q iff, per axis,
 *       fl(c_p - rad) <= q <= fl(c_p + rad) on the fp32 values the bounds program writes
 *       (closed box; RT cores may be conservative at the boundary, that is not restatable).
 *   (2) visit order of candidates is unspecified in the reference; the canonical order here is
 *       ascending primitive index, which together with the strict '<' of deviceCode.cu:116,125
 *       yields rows ordered by (dist, index).  Other orders are selectable to test that only
 *       ties depend on it.  One exception, because the lists persist over rounds
 *       (deviceCode.cu:77-85 skips what is listed already): of two candidates at bit-identical fp32
 *       distances the one that entered in an EARLIER round stays ahead whatever its index.  This
 *       replay reproduces that, and so do the HIP engines (their kernels list by (dist, index) and
 *       flag the rows where the round matters; a second pass redoes those with the full key --
 *       tests/conftest.py::assert_rows_equal demands index-for-index equality).
 *   (3) distance arithmetic: d = sqrtf((dx*dx + dy*dy) + dz*dz), every operation rounded to fp32
 *       on its own, i.e. deviceCode.cu:110-113 exactly as written, and a correctly rounded IEEE
 *       sqrt.  The reference's Release build lets nvcc contract the sum into fmas of its choosing
 *       and uses sqrt.approx.ftz (owl/cmake/configure_optix.cmake:49-53); neither is reproducible
 *       off that compiler and hardware, and which products get fused differs between compilers
 *       (hipcc's default contraction of the same line gives yet another rounding), so the
 *       uncontracted expression is the canonical one.  The HIP engine computes it the same way,
 *       and tools/owl_embed.py compiles user device programs with -ffp-contract=off so the
 *       unchanged deviceCode.cu does too.
 *
 * Build: see oracle/Makefile (-ffp-contract=off: nothing contracts).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* wall time the last tkref_trueknn call spent in its query loops (candidate enumeration + the
 * intersection program), excluding the per-round grid construction over all points, which a
 * full run amortises over all n queries; read by bench.py's cpu_baseline leg */
static double g_query_seconds = 0.0;
double tkref_last_query_seconds(void) { return g_query_seconds; }
static double tk_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* GeomTypes.h:22-28 -- 4+4+4(+4 pad)+8 = 24 bytes */
typedef struct {
  int32_t ind;
  float dist;
  int32_t numNeighbors;
  int64_t intersections;
} tk_neigh;

enum { TK_ORDER_ASCENDING = 0, TK_ORDER_DESCENDING = 1, TK_ORDER_SHUFFLED = 2 };

/* hostCode.cpp:41 '#define FLOAT_MAX 3.402823466e+38' (a double literal) narrowed into Neigh.dist */
static const float TK_FLOAT_MAX = (float)3.402823466e+38;

int tkref_sizeof_neigh(void) { return (int)sizeof(tk_neigh); }

/* hostCode.cpp:127-130 */
void tkref_init_rows(tk_neigh *fb, int64_t n, int k) {
  for (int64_t j = 0; j < n; j++)
    for (int i = 0; i < k; i++) {
      tk_neigh *e = &fb[j * k + i];
      memset(e, 0, sizeof *e);
      e->ind = -1;
      e->dist = TK_FLOAT_MAX;
      e->numNeighbors = k;
      e->intersections = 0;
    }
}

/* deviceCode.cu:38-56 with box.h:128-132: empty box extended by (c - rad) then (c + rad). */
static inline void tk_bounds(const float *c, float rad, float *lo, float *hi) {
  for (int a = 0; a < 3; a++) {
    float m = c[a] - rad, p = c[a] + rad;
    lo[a] = fminf(fminf(INFINITY, m), p);
    hi[a] = fmaxf(fmaxf(-INFINITY, m), p);
  }
}

/* decision (1): closed point-in-box test on the fp32 box of tk_bounds */
static inline int tk_box_hit(const float *q, const float *c, float rad) {
  float lo[3], hi[3];
  tk_bounds(c, rad, lo, hi);
  return lo[0] <= q[0] && q[0] <= hi[0] && lo[1] <= q[1] && q[1] <= hi[1] && lo[2] <= q[2] &&
         q[2] <= hi[2];
}

/* decision (3) */
float tkref_distance(const float *c_prim, const float *org) {
  float x = c_prim[0] - org[0];
  float y = c_prim[1] - org[1];
  float z = c_prim[2] - org[2];
  return sqrtf(((x * x) + (y * y)) + (z * z)); /* -ffp-contract=off: as written */
}

/* deviceCode.cu:62-138, one call per (query xID, candidate primID) */
static void tk_intersect(tk_neigh *row /* = &frameBuffer[xID * k] */, const float *xyz, int k, int32_t xID, int32_t primID) {
  row[0].intersections += 1;                                 /* :74 */
  for (int i = 0; i < k; i++)                                /* :77-85 */
    if (row[i].ind == primID) return;
  float maxDist = row[k - 1].dist;                           /* :100 */
  if (xID != primID) {                                       /* :103 */
    float distance = tkref_distance(xyz + 3 * (int64_t)primID, xyz + 3 * (int64_t)xID);
    if (distance < maxDist) {                                /* :116 */
      if (row[0].numNeighbors > 0) row[0].numNeighbors -= 1; /* :118-119 */
      int q = 0, w = k - 1;
      for (; q < k; q++)                                     /* :121-127 */
        if (distance < row[q].dist) break;
      for (; w > q; w--) {                                   /* :129-132 */
        row[w].dist = row[w - 1].dist;
        row[w].ind = row[w - 1].ind;
      }
      row[w].dist = distance;                                /* :133-134 */
      row[w].ind = primID;
    }
  }
}

/* ------------------------------------------------------------------ */
/* candidate enumeration: a uniform grid rebuilt per round.  It only   */
/* proposes primitives; tk_box_hit decides.                            */
/* ------------------------------------------------------------------ */
typedef struct {
  double org[3], inv[3];
  int dim[3];
  int64_t *start; /* ncell+1 */
  int32_t *items; /* n, ascending primitive index inside each cell */
} tk_grid;

static inline int tk_cell_of(const tk_grid *g, int a, double v) {
  double t = (v - g->org[a]) * g->inv[a];
  if (!(t > 0)) return 0;
  if (t >= g->dim[a]) return g->dim[a] - 1;
  return (int)t;
}

static int tk_grid_build(tk_grid *g, const float *xyz, int64_t n, double reach) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) {
      double v = xyz[3 * i + a];
      if (v < lo[a]) lo[a] = v;
      if (v > hi[a]) hi[a] = v;
    }
  double cell = reach > 0 ? reach : 1.0;
  /* cap the cell count near 4n so memory stays O(n) whatever the radius */
  for (;;) {
    double tot = 1;
    for (int a = 0; a < 3; a++) {
      double e = (hi[a] - lo[a]) / cell;
      double d = floor(e) + 1;
      if (!(d >= 1)) d = 1;
      tot *= d;
    }
    if (tot <= 4.0 * (double)n + 64) break;
    cell *= 1.26;
  }
  int64_t ncell = 1;
  for (int a = 0; a < 3; a++) {
    g->org[a] = lo[a];
    g->inv[a] = 1.0 / cell;
    double d = floor((hi[a] - lo[a]) / cell) + 1;
    if (!(d >= 1)) d = 1;
    g->dim[a] = (int)d;
    ncell *= g->dim[a];
  }
  g->start = (int64_t *)calloc((size_t)ncell + 1, sizeof(int64_t));
  g->items = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  if (!g->start || !g->items) return -1;
  for (int64_t i = 0; i < n; i++) {
    int cx = tk_cell_of(g, 0, xyz[3 * i]), cy = tk_cell_of(g, 1, xyz[3 * i + 1]),
        cz = tk_cell_of(g, 2, xyz[3 * i + 2]);
    g->start[((int64_t)cz * g->dim[1] + cy) * g->dim[0] + cx + 1]++;
  }
  for (int64_t c = 0; c < ncell; c++) g->start[c + 1] += g->start[c];
  int64_t *fill = (int64_t *)malloc((size_t)ncell * sizeof(int64_t));
  if (!fill) return -1;
  memcpy(fill, g->start, (size_t)ncell * sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) { /* ascending i => ascending inside each cell */
    int cx = tk_cell_of(g, 0, xyz[3 * i]), cy = tk_cell_of(g, 1, xyz[3 * i + 1]),
        cz = tk_cell_of(g, 2, xyz[3 * i + 2]);
    g->items[fill[((int64_t)cz * g->dim[1] + cy) * g->dim[0] + cx]++] = (int32_t)i;
  }
  free(fill);
  return 0;
}

static void tk_grid_free(tk_grid *g) {
  free(g->start);
  free(g->items);
  g->start = NULL;
  g->items = NULL;
}

static int tk_cmp_i32(const void *a, const void *b) {
  int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
  return (x > y) - (x < y);
}

static inline uint64_t tk_mix(uint64_t x) { /* splitmix64 */
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* all primitives whose box (radius rad) contains query point q, in the requested visit order */
static int64_t tk_candidates(const tk_grid *g, const float *xyz, const float *q, float rad,
                             int order, uint64_t seed, int32_t **buf, int64_t *cap) {
  double reach = fabs((double)rad) * 1.0001 + 1e-30;
  int c0[3], c1[3];
  for (int a = 0; a < 3; a++) {
    double slack = reach + 1e-6 * fabs((double)q[a]);
    c0[a] = tk_cell_of(g, a, (double)q[a] - slack);
    c1[a] = tk_cell_of(g, a, (double)q[a] + slack);
  }
  int64_t m = 0;
  for (int cz = c0[2]; cz <= c1[2]; cz++)
    for (int cy = c0[1]; cy <= c1[1]; cy++)
      for (int cx = c0[0]; cx <= c1[0]; cx++) {
        int64_t c = ((int64_t)cz * g->dim[1] + cy) * g->dim[0] + cx;
        for (int64_t s = g->start[c]; s < g->start[c + 1]; s++) {
          int32_t p = g->items[s];
          if (!tk_box_hit(q, xyz + 3 * (int64_t)p, rad)) continue;
          if (m == *cap) {
            *cap = *cap ? *cap * 2 : 256;
            *buf = (int32_t *)realloc(*buf, (size_t)*cap * sizeof(int32_t));
            if (!*buf) return -1;
          }
          (*buf)[m++] = p;
        }
      }
  qsort(*buf, (size_t)m, sizeof(int32_t), tk_cmp_i32);
  if (order == TK_ORDER_DESCENDING) {
    for (int64_t i = 0, j = m - 1; i < j; i++, j--) {
      int32_t t = (*buf)[i];
      (*buf)[i] = (*buf)[j];
      (*buf)[j] = t;
    }
  } else if (order == TK_ORDER_SHUFFLED) {
    uint64_t s = seed;
    for (int64_t i = m - 1; i > 0; i--) {
      s = tk_mix(s);
      int64_t j = (int64_t)(s % (uint64_t)(i + 1));
      int32_t t = (*buf)[i];
      (*buf)[i] = (*buf)[j];
      (*buf)[j] = t;
    }
  }
  return m;
}

/*
 * The whole solve (hostCode.cpp:285-340 around deviceCode.cu:62-153).
 *   xyz        n x 3 fp32 points (2-D input already padded with z = 0, hostCode.cpp:115-118)
 *   fb         n*k rows, initialised with tkref_init_rows (or carrying state of earlier rounds)
 *   query_ids  NULL = every point is a query (the reference); else only these rows are traced
 *              and consulted by the round loop (a row's result does not depend on other rows)
 *   returns    number of rounds run, or <0: -1 bad arguments, -2 out of memory,
 *              -3 max_rounds reached with unfinished queries (the reference would loop forever,
 *              e.g. n <= k)
 */
static int tk_solve(const float *xyz, int64_t n, int k, float start_radius, int order, uint64_t seed,
                    const int32_t *query_ids, int64_t n_queries, int max_rounds, tk_neigh *fb,
                    float *final_radius, int compact) {
  if (!xyz || !fb || n <= 0 || k <= 0 || n > INT32_MAX) return -1;
  if (compact && !query_ids) return -1;
  if (!query_ids) n_queries = n;
  /* row of the t-th query: at its index in the reference's frameBuffer, or -- compact -- the t-th row of an array that
   * holds the sampled queries only (100 M points x k = 10 would be 24 GB of frameBuffer for a thousand sampled rows) */
#define TK_ROW(t, xID) (fb + (compact ? (int64_t)(t) : (int64_t)(xID)) * k)
  float radius = start_radius;
  int rounds = 0;
  g_query_seconds = 0.0;
  for (;;) {
    if (rounds >= max_rounds) return -3;
    rounds++;
    tk_grid g;
    if (tk_grid_build(&g, xyz, n, fabs((double)radius) * 1.0001 + 1e-30)) return -2;
    int failed = 0;
    double t_start = tk_now();
#pragma omp parallel
    {
      int32_t *buf = NULL;
      int64_t cap = 0;
#pragma omp for schedule(dynamic, 256)
      for (int64_t t = 0; t < n_queries; t++) {
        int32_t xID = query_ids ? query_ids[t] : (int32_t)t;
        tk_neigh *row = TK_ROW(t, xID);
        if (!(row[0].numNeighbors > 0)) continue; /* deviceCode.cu:148 */
        int64_t m = tk_candidates(&g, xyz, xyz + 3 * (int64_t)xID, radius, order,
                                  seed ^ tk_mix((uint64_t)xID * 1315423911ull + (uint64_t)rounds),
                                  &buf, &cap);
        if (m < 0) {
          failed = 1;
          continue;
        }
        for (int64_t j = 0; j < m; j++) tk_intersect(row, xyz, k, xID, buf[j]);
      }
      free(buf);
    }
    g_query_seconds += tk_now() - t_start;
    tk_grid_free(&g);
    if (failed) return -2;
    int again = 0; /* hostCode.cpp:310-330 */
    for (int64_t t = 0; t < n_queries; t++) {
      int32_t j = query_ids ? query_ids[t] : (int32_t)t;
      if (TK_ROW(t, j)[0].numNeighbors > 0) {
        again = 1;
        radius *= 2;
        break;
      }
    }
    if (!again) break;
  }
  if (final_radius) *final_radius = radius;
  return rounds;
#undef TK_ROW
}

int tkref_trueknn(const float *xyz, int64_t n, int k, float start_radius, int order, uint64_t seed,
                  const int32_t *query_ids, int64_t n_queries, int max_rounds, tk_neigh *fb,
                  float *final_radius) {
  return tk_solve(xyz, n, k, start_radius, order, seed, query_ids, n_queries, max_rounds, fb, final_radius, 0);
}

/* The same solve for sampled queries only, rows stored compactly: `rows` holds n_queries * k records (initialised with
 * tkref_init_rows(rows, n_queries, k)), row t belongs to query_ids[t].  For checks at sizes where the reference's
 * n * k frameBuffer would not fit the checker's host (BASELINE config 4: 10^8 points). */
int tkref_trueknn_rows(const float *xyz, int64_t n, int k, float start_radius, int order, uint64_t seed,
                       const int32_t *query_ids, int64_t n_queries, int max_rounds, tk_neigh *rows,
                       float *final_radius) {
  return tk_solve(xyz, n, k, start_radius, order, seed, query_ids, n_queries, max_rounds, rows, final_radius, 1);
}

/*
 * Exact brute-force kNN (the "CPU brute-force O(n^2)" baseline BASELINE.json names; the
 * reference itself ships none).  Same distance arithmetic as above, rows ordered by
 * (dist, index), self excluded by index.  Not the parity oracle: the reference's output is
 * box-candidate kNN, which differs from exact kNN for a sizeable share of queries.
 */
int tkref_bruteforce(const float *xyz, int64_t n, int k, const int32_t *query_ids,
                     int64_t n_queries, int32_t *out_idx, float *out_dist) {
  if (!xyz || n <= 0 || k <= 0 || !out_idx || !out_dist) return -1;
  if (!query_ids) n_queries = n;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t t = 0; t < n_queries; t++) {
    int32_t q = query_ids ? query_ids[t] : (int32_t)t;
    int32_t *ri = out_idx + t * k;
    float *rd = out_dist + t * k;
    for (int i = 0; i < k; i++) {
      ri[i] = -1;
      rd[i] = INFINITY;
    }
    const float *org = xyz + 3 * (int64_t)q;
    for (int64_t p = 0; p < n; p++) {
      if (p == q) continue;
      float d = tkref_distance(xyz + 3 * p, org);
      if (!(d < rd[k - 1])) continue; /* ascending p + strict '<' => (dist, index) order */
      int w = k - 1;
      while (w > 0 && d < rd[w - 1]) {
        rd[w] = rd[w - 1];
        ri[w] = ri[w - 1];
        w--;
      }
      rd[w] = d;
      ri[w] = (int32_t)p;
    }
  }
  return 0;
}

int tkref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
