/*
 * dbscan_oracle.c -- CPU statement of the RT-DBSCAN result this repo's HIP path must produce.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * PARITY UNPINNED, and more than for TrueKNN: the reference tree holds NO RT-DBSCAN source at all
 * (samples/s02-rtdbscan does not exist; README.md:8-9 only says the method "offloads distance
 * computations in DBSCAN to the ray tracing cores and performs other clustering operations in
 * shader cores").  So this file is a SPEC written for this build, not a restatement:
 *
 *   neighbourhood   N(p) = { q : dist(p, q) <= eps }, p itself included; dist is the fp32
 *                   arithmetic of the TrueKNN intersection program (deviceCode.cu:110-113 as
 *                   written): sqrtf((dx*dx + dy*dy) + dz*dz), every operation rounded
 *   core point      |N(p)| >= minPts
 *   cluster         connected component of core points under  dist <= eps
 *   border point    non-core with a core point in N(p): joins the adjacent cluster with the
 *                   smallest label
 *   labels          clusters numbered 0,1,... by ascending smallest core index; noise = -1
 *
 * With these rules the labelling is unique and is exactly what sklearn.cluster.DBSCAN returns
 * (it expands clusters in index order, so cluster ids follow the smallest core index and a border
 * point is taken by the lowest-numbered cluster that reaches it), except that sklearn measures
 * distances in float64: tests compare with sklearn only on inputs where no pair lies within a few
 * ulps of eps.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float db_dist(const float *a, const float *b) {
  float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
  return sqrtf(((x * x) + (y * y)) + (z * z));
}

typedef struct {
  double org[3], inv;
  int dim[3];
  int64_t *start;
  int32_t *items;
} db_grid;

static inline int db_cell(const db_grid *g, int a, double v) {
  double t = (v - g->org[a]) * g->inv;
  if (!(t > 0)) return 0;
  if (t >= g->dim[a]) return g->dim[a] - 1;
  return (int)t;
}

static int db_grid_build(db_grid *g, const float *xyz, int64_t n, double cell) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) {
      double v = xyz[3 * i + a];
      if (v < lo[a]) lo[a] = v;
      if (v > hi[a]) hi[a] = v;
    }
  for (;;) {
    double tot = 1;
    for (int a = 0; a < 3; a++) tot *= floor((hi[a] - lo[a]) / cell) + 1;
    if (tot <= 8.0 * (double)n + 64) break;
    cell *= 1.26;
  }
  int64_t ncell = 1;
  g->inv = 1.0 / cell;
  for (int a = 0; a < 3; a++) {
    g->org[a] = lo[a];
    g->dim[a] = (int)(floor((hi[a] - lo[a]) / cell) + 1);
    ncell *= g->dim[a];
  }
  g->start = (int64_t *)calloc((size_t)ncell + 1, sizeof(int64_t));
  g->items = (int32_t *)malloc((size_t)(n ? n : 1) * sizeof(int32_t));
  int64_t *fill = (int64_t *)malloc((size_t)ncell * sizeof(int64_t));
  if (!g->start || !g->items || !fill) return -1;
#define DB_CELL_OF(i) \
  (((int64_t)db_cell(g, 2, xyz[3 * (i) + 2]) * g->dim[1] + db_cell(g, 1, xyz[3 * (i) + 1])) * g->dim[0] + db_cell(g, 0, xyz[3 * (i)]))
  for (int64_t i = 0; i < n; i++) g->start[DB_CELL_OF(i) + 1]++;
  for (int64_t c = 0; c < ncell; c++) g->start[c + 1] += g->start[c];
  memcpy(fill, g->start, (size_t)ncell * sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) g->items[fill[DB_CELL_OF(i)]++] = (int32_t)i;
  free(fill);
  return 0;
}

/* calls f(p, ctx) for every p with dist(p, q) <= eps (q itself included); returns count */
typedef void (*db_visit)(int32_t p, void *ctx);
static int64_t db_neighbours(const db_grid *g, const float *xyz, int32_t q, float eps, db_visit f, void *ctx) {
  const float *c = xyz + 3 * (int64_t)q;
  double reach = (double)eps * 1.0001 + 1e-30;
  int c0[3], c1[3];
  for (int a = 0; a < 3; a++) {
    double slack = reach + 1e-6 * fabs((double)c[a]);
    c0[a] = db_cell(g, a, (double)c[a] - slack);
    c1[a] = db_cell(g, a, (double)c[a] + slack);
  }
  int64_t m = 0;
  for (int z = c0[2]; z <= c1[2]; z++)
    for (int y = c0[1]; y <= c1[1]; y++)
      for (int x = c0[0]; x <= c1[0]; x++) {
        int64_t cell = ((int64_t)z * g->dim[1] + y) * g->dim[0] + x;
        for (int64_t s = g->start[cell]; s < g->start[cell + 1]; s++) {
          int32_t p = g->items[s];
          if (db_dist(xyz + 3 * (int64_t)p, c) <= eps) {
            m++;
            if (f) f(p, ctx);
          }
        }
      }
  return m;
}

static int32_t db_find(int32_t *parent, int32_t x) {
  while (parent[x] != x) {
    parent[x] = parent[parent[x]];
    x = parent[x];
  }
  return x;
}

typedef struct {
  int32_t *parent;
  const uint8_t *core;
  int32_t self;
  int32_t best;
} db_ctx;

static void db_union_visit(int32_t p, void *vctx) {
  db_ctx *c = (db_ctx *)vctx;
  if (!c->core[p] || p == c->self) return;
  int32_t a = db_find(c->parent, c->self), b = db_find(c->parent, p);
  if (a == b) return;
  if (a < b)
    c->parent[b] = a; /* the smaller index stays root: roots are the smallest core index */
  else
    c->parent[a] = b;
}

static void db_border_visit(int32_t p, void *vctx) {
  db_ctx *c = (db_ctx *)vctx;
  if (!c->core[p]) return;
  int32_t r = db_find(c->parent, p);
  if (c->best < 0 || r < c->best) c->best = r;
}

/* xyz: n x 3 fp32.  labels: n int32 out.  core: n uint8 out.  counts: n int32 out (|N(p)|, may be NULL).
 * returns the number of clusters, or <0 on bad arguments / out of memory. */
int dbref_dbscan(const float *xyz, int64_t n, float eps, int min_pts, int32_t *labels, uint8_t *core,
                 int32_t *counts) {
  if (!xyz || !labels || !core || n <= 0 || !(eps > 0) || min_pts < 1) return -1;
  db_grid g;
  if (db_grid_build(&g, xyz, n, (double)eps)) return -2;
  int32_t *parent = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  if (!parent) return -2;
  for (int64_t i = 0; i < n; i++) {
    int64_t m = db_neighbours(&g, xyz, (int32_t)i, eps, NULL, NULL);
    core[i] = m >= min_pts;
    if (counts) counts[i] = (int32_t)m;
    parent[i] = (int32_t)i;
  }
  db_ctx ctx;
  ctx.parent = parent;
  ctx.core = core;
  for (int64_t i = 0; i < n; i++)
    if (core[i]) {
      ctx.self = (int32_t)i;
      db_neighbours(&g, xyz, (int32_t)i, eps, db_union_visit, &ctx);
    }
  /* roots in ascending index order get labels 0,1,...; borders take the smallest adjacent root */
  int32_t *rank = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  if (!rank) return -2;
  int32_t nclusters = 0;
  for (int64_t i = 0; i < n; i++) rank[i] = (core[i] && db_find(parent, (int32_t)i) == i) ? nclusters++ : -1;
  for (int64_t i = 0; i < n; i++) {
    if (core[i]) {
      labels[i] = rank[db_find(parent, (int32_t)i)];
    } else {
      ctx.best = -1;
      db_neighbours(&g, xyz, (int32_t)i, eps, db_border_visit, &ctx);
      labels[i] = ctx.best < 0 ? -1 : rank[ctx.best];
    }
  }
  free(rank);
  free(parent);
  free(g.start);
  free(g.items);
  return nclusters;
}


/* ---- the same spec on all host cores: the CPU baseline bench.py times beside RT-DBSCAN ----------------
 * ("threaded grid DBSCAN on the GPU box").  Same grid, same distance
 * arithmetic, same labelling rule as dbref_dbscan above, so the two agree label for label
 * (tests/test_dbscan.py); what differs is the schedule: core flags stop counting at minPts, unions run
 * concurrently on a lock-free union-find (a root only ever gets a smaller parent, so the component's
 * root is its smallest core index whatever the interleaving), borders and the ranking are parallel.
 * seconds (may be NULL): [0] grid, [1] core flags, [2] unions, [3] labels. */
#include <omp.h>

static int64_t db_count_upto(const db_grid *g, const float *xyz, int32_t q, float eps, int64_t stop) {
  const float *c = xyz + 3 * (int64_t)q;
  double reach = (double)eps * 1.0001 + 1e-30;
  int c0[3], c1[3];
  for (int a = 0; a < 3; a++) {
    double slack = reach + 1e-6 * fabs((double)c[a]);
    c0[a] = db_cell(g, a, (double)c[a] - slack);
    c1[a] = db_cell(g, a, (double)c[a] + slack);
  }
  int64_t m = 0;
  for (int z = c0[2]; z <= c1[2]; z++)
    for (int y = c0[1]; y <= c1[1]; y++)
      for (int x = c0[0]; x <= c1[0]; x++) {
        int64_t cell = ((int64_t)z * g->dim[1] + y) * g->dim[0] + x;
        for (int64_t s = g->start[cell]; s < g->start[cell + 1]; s++)
          if (db_dist(xyz + 3 * (int64_t)g->items[s], c) <= eps && ++m >= stop) return m;
      }
  return m;
}

static inline int32_t db_find_mt(int32_t *parent, int32_t x) {
  for (;;) {
    int32_t p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED);
    if (p == x) return x;
    int32_t gp = __atomic_load_n(&parent[p], __ATOMIC_RELAXED);
    if (gp != p) __atomic_compare_exchange_n(&parent[x], &p, gp, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED); /* halving */
    x = p;
  }
}

static void db_unite_mt(int32_t *parent, int32_t a, int32_t b) {
  for (;;) {
    a = db_find_mt(parent, a);
    b = db_find_mt(parent, b);
    if (a == b) return;
    if (a > b) {
      int32_t t = a;
      a = b;
      b = t;
    }
    int32_t expect = b; /* b is a root: hang it under the smaller root a */
    if (__atomic_compare_exchange_n(&parent[b], &expect, a, 0, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) return;
  }
}

int dbref_dbscan_mt(const float *xyz, int64_t n, float eps, int min_pts, int32_t *labels, uint8_t *core,
                    double *seconds) {
  if (!xyz || !labels || !core || n <= 0 || !(eps > 0) || min_pts < 1) return -1;
  double t0 = omp_get_wtime();
  db_grid g;
  if (db_grid_build(&g, xyz, n, (double)eps)) return -2;
  int32_t *parent = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  int32_t *rank = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  if (!parent || !rank) return -2;
  double t1 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < n; i++) {
    core[i] = db_count_upto(&g, xyz, (int32_t)i, eps, min_pts) >= min_pts;
    parent[i] = (int32_t)i;
  }
  double t2 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < n; i++) {
    if (!core[i]) continue;
    const float *c = xyz + 3 * i;
    double reach = (double)eps * 1.0001 + 1e-30;
    int c0[3], c1[3];
    for (int a = 0; a < 3; a++) {
      double slack = reach + 1e-6 * fabs((double)c[a]);
      c0[a] = db_cell(&g, a, (double)c[a] - slack);
      c1[a] = db_cell(&g, a, (double)c[a] + slack);
    }
    int32_t mine = db_find_mt(parent, (int32_t)i);
    for (int z = c0[2]; z <= c1[2]; z++)
      for (int y = c0[1]; y <= c1[1]; y++)
        for (int x = c0[0]; x <= c1[0]; x++) {
          int64_t cell = ((int64_t)z * g.dim[1] + y) * g.dim[0] + x;
          for (int64_t s = g.start[cell]; s < g.start[cell + 1]; s++) {
            int32_t p = g.items[s];
            /* each pair once (from its larger index); a cached root spares the distance for pairs already joined */
            if (p >= i || !core[p]) continue;
            if (__atomic_load_n(&parent[p], __ATOMIC_RELAXED) == mine) continue;
            if (db_dist(xyz + 3 * (int64_t)p, c) <= eps) {
              db_unite_mt(parent, (int32_t)i, p);
              mine = db_find_mt(parent, (int32_t)i);
            }
          }
        }
  }
  double t3 = omp_get_wtime();
  /* roots in ascending index order get labels 0,1,... (serial prefix over n flags: memory-bound, short) */
  int32_t nclusters = 0;
  for (int64_t i = 0; i < n; i++) rank[i] = (core[i] && parent[i] == (int32_t)i) ? nclusters++ : -1;
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < n; i++) {
    if (core[i]) {
      labels[i] = rank[db_find_mt(parent, (int32_t)i)];
      continue;
    }
    const float *c = xyz + 3 * i;
    double reach = (double)eps * 1.0001 + 1e-30;
    int c0[3], c1[3];
    for (int a = 0; a < 3; a++) {
      double slack = reach + 1e-6 * fabs((double)c[a]);
      c0[a] = db_cell(&g, a, (double)c[a] - slack);
      c1[a] = db_cell(&g, a, (double)c[a] + slack);
    }
    int32_t best = -1;
    for (int z = c0[2]; z <= c1[2]; z++)
      for (int y = c0[1]; y <= c1[1]; y++)
        for (int x = c0[0]; x <= c1[0]; x++) {
          int64_t cell = ((int64_t)z * g.dim[1] + y) * g.dim[0] + x;
          for (int64_t s = g.start[cell]; s < g.start[cell + 1]; s++) {
            int32_t p = g.items[s];
            if (!core[p] || db_dist(xyz + 3 * (int64_t)p, c) > eps) continue;
            int32_t r = db_find_mt(parent, p);
            if (best < 0 || r < best) best = r;
          }
        }
    labels[i] = best < 0 ? -1 : rank[best];
  }
  double t4 = omp_get_wtime();
  if (seconds) {
    seconds[0] = t1 - t0;
    seconds[1] = t2 - t1;
    seconds[2] = t3 - t2;
    seconds[3] = t4 - t3;
  }
  free(rank);
  free(parent);
  free(g.start);
  free(g.items);
  return nclusters;
}


/* ---- "eps auto-grown" (BASELINE.json configs[4]) ---------------------------------------------------------------
 * The reference has nothing of the kind (no RT-DBSCAN source at all, and BASELINE.md section 4 marks the
 * rule "spec TBD"), so this, too, is a SPEC written for this build.  It is built on the one growth rule the reference
 * has, the radius doubling of its round loop (samples/s01-trueknn/hostCode.cpp:310-330: while some query is
 * unfinished, radius *= 2 and go again), carried over from "a query without k neighbours" to "a point without a
 * cluster":
 *
 *   eps_0 = the given start value (> 0; the start-radius sampler of owlraytracing_amd/radius.py supplies one);
 *   round t: DBSCAN(eps_t, minPts) as specified above; noise_t = number of points labelled -1;
 *   finished at the first t with  noise_t <= floor(max_noise * n)  (max_noise in [0, 1]); else eps_{t+1} = eps_t * 2
 *   in fp32 (hostCode.cpp:321) and another round -- at most max_rounds rounds, like the reference's loop a global
 *   value: one eps for all points.
 *   Result: the labelling of the final round, its eps and the number of rounds run.
 *
 * Facts an implementation may use (both follow from N_eps(p) growing with eps): a core point stays core, and a point
 * that is not noise stays not noise, so noise_t never grows and only the final round needs clusters.
 * Returns the number of clusters of the final round; < 0: bad arguments / out of memory; -3: max_rounds exhausted
 * (labels then hold the last round's). */
int dbref_dbscan_auto(const float *xyz, int64_t n, float eps0, int min_pts, double max_noise, int max_rounds,
                      int32_t *labels, uint8_t *core, float *eps_final, int *rounds, int64_t *noise_final) {
  if (!xyz || !labels || !core || n <= 0 || !(eps0 > 0) || min_pts < 1 || !(max_noise >= 0) || !(max_noise <= 1) || max_rounds < 1) return -1;
  const int64_t bound = (int64_t)floor(max_noise * (double)n);
  float eps = eps0;
  for (int t = 0; t < max_rounds; t++) {
    int rc = dbref_dbscan_mt(xyz, n, eps, min_pts, labels, core, NULL);
    if (rc < 0) return rc;
    int64_t noise = 0;
    for (int64_t i = 0; i < n; i++) noise += labels[i] < 0;
    if (eps_final) *eps_final = eps;
    if (rounds) *rounds = t + 1;
    if (noise_final) *noise_final = noise;
    if (noise <= bound) return rc;
    eps = eps * 2.0f; /* hostCode.cpp:321 */
  }
  return -3;
}


/* ---- count-only growth round and sampled eps-ball check: checks at sizes where the full spec takes too long --------
 * (BASELINE configs[4]: 50 M 2-D points, eps auto-grown; bench.py's spot check of config 3 inside a run)
 *
 * dbref_noise_count_mt: the number of points DBSCAN(eps, minPts) labels -1, without building clusters -- a point is
 * noise iff it is not core and has no core point within eps (spec above); core flags stop counting at minPts, the
 * search for a core neighbour stops at the first.  core (n bytes, may be NULL) receives the core flags.
 * Returns the noise count, or < 0 (bad arguments / out of memory). */
static int db_has_core_neighbour(const db_grid *g, const float *xyz, const uint8_t *core, int32_t q, float eps) {
  const float *c = xyz + 3 * (int64_t)q;
  double reach = (double)eps * 1.0001 + 1e-30;
  int c0[3], c1[3];
  for (int a = 0; a < 3; a++) {
    double slack = reach + 1e-6 * fabs((double)c[a]);
    c0[a] = db_cell(g, a, (double)c[a] - slack);
    c1[a] = db_cell(g, a, (double)c[a] + slack);
  }
  for (int z = c0[2]; z <= c1[2]; z++)
    for (int y = c0[1]; y <= c1[1]; y++)
      for (int x = c0[0]; x <= c1[0]; x++) {
        int64_t cell = ((int64_t)z * g->dim[1] + y) * g->dim[0] + x;
        for (int64_t s = g->start[cell]; s < g->start[cell + 1]; s++) {
          int32_t p = g->items[s];
          if (core[p] && db_dist(xyz + 3 * (int64_t)p, c) <= eps) return 1;
        }
      }
  return 0;
}

int64_t dbref_noise_count_mt(const float *xyz, int64_t n, float eps, int min_pts, uint8_t *core_out) {
  if (!xyz || n <= 0 || !(eps > 0) || min_pts < 1) return -1;
  db_grid g;
  if (db_grid_build(&g, xyz, n, (double)eps)) return -2;
  uint8_t *core = core_out ? core_out : (uint8_t *)malloc((size_t)n);
  if (!core) return -2;
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < n; i++) core[i] = db_count_upto(&g, xyz, (int32_t)i, eps, min_pts) >= min_pts;
  int64_t noise = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : noise)
  for (int64_t i = 0; i < n; i++)
    if (!core[i] && !db_has_core_neighbour(&g, xyz, core, (int32_t)i, eps)) noise++;
  if (!core_out) free(core);
  free(g.start);
  free(g.items);
  return noise;
}

/* dbref_ball_check: labels / core flags of a finished clustering (from anywhere: the HIP path) checked on sampled
 * points by recomputing their eps-balls with the spec's arithmetic:
 *   core[q] == (|N(q)| >= minPts);
 *   a core q has a label >= 0 and shares it with every core point of its ball;
 *   a border q (not core, some core point in its ball) carries the smallest label among those core points;
 *   a q with no core point in its ball is labelled -1.
 * Returns the number of sampled points that violate any of these (first_bad: one of them, or -1), < 0 on errors. */
int64_t dbref_ball_check(const float *xyz, int64_t n, float eps, int min_pts, const int32_t *labels, const uint8_t *core,
                         const int32_t *sample, int64_t m, int32_t *first_bad) {
  if (!xyz || !labels || !core || !sample || n <= 0 || m < 0 || !(eps > 0) || min_pts < 1) return -1;
  db_grid g;
  if (db_grid_build(&g, xyz, n, (double)eps)) return -2;
  int64_t bad = 0;
  int32_t first = -1;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : bad)
  for (int64_t t = 0; t < m; t++) {
    const int32_t q = sample[t];
    if (q < 0 || q >= n) {
      bad++;
      continue;
    }
    const float *c = xyz + 3 * (int64_t)q;
    double reach = (double)eps * 1.0001 + 1e-30;
    int c0[3], c1[3];
    for (int a = 0; a < 3; a++) {
      double slack = reach + 1e-6 * fabs((double)c[a]);
      c0[a] = db_cell(&g, a, (double)c[a] - slack);
      c1[a] = db_cell(&g, a, (double)c[a] + slack);
    }
    int64_t count = 0;
    int32_t best = -1;
    int differs = 0;
    for (int z = c0[2]; z <= c1[2]; z++)
      for (int y = c0[1]; y <= c1[1]; y++)
        for (int x = c0[0]; x <= c1[0]; x++) {
          int64_t cell = ((int64_t)z * g.dim[1] + y) * g.dim[0] + x;
          for (int64_t s = g.start[cell]; s < g.start[cell + 1]; s++) {
            int32_t p = g.items[s];
            if (db_dist(xyz + 3 * (int64_t)p, c) > eps) continue;
            count++;
            if (!core[p]) continue;
            if (labels[p] != labels[q]) differs = 1;
            if (best < 0 || labels[p] < best) best = labels[p];
          }
        }
    int ok = (core[q] != 0) == (count >= min_pts);
    if (core[q])
      ok = ok && labels[q] >= 0 && !differs;
    else
      ok = ok && labels[q] == (best < 0 ? -1 : best);
    if (!ok) {
      bad++;
#pragma omp critical
      if (first < 0 || q < first) first = q;
    }
  }
  if (first_bad) *first_bad = first;
  free(g.start);
  free(g.items);
  return bad;
}
