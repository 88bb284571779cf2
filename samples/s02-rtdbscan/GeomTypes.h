// GeomTypes.h -- host/device records of sample02-rtdbscan.
//
// RT-DBSCAN as an OWL application (the reference names the method, README.md:8-9, but ships no
// samples/s02-rtdbscan; this one is written for this repository, in the shape of samples/s01-trueknn:
// points become axis-aligned boxes of half-width eps, every point shoots a point-like ray, the
// intersection program runs for every box the ray's origin lies in and does the true sphere test).
#pragma once
#include <owl/owl.h>
#include <owl/common/math/vec.h>

using namespace owl;

/* one input point (the Sphere of s01) */
struct Point {
  vec3f center;
};

/* geometry record: what the bounds and intersection programs see */
struct PointsGeom {
  Point *prims;
  float eps;
};

/* ray-generation record */
struct RayGenData {
  OptixTraversableHandle world;
  int n;
};

/* what a launch does: the intersection program is one and the same, the phase selects its action */
enum {
  DB_COUNT = 1,   /* |N(q)|: points within eps of q, q included                                  */
  DB_MARK = 2,    /* no ray: core[q] = count[q] >= minPts, parent[q] = q, label[q] = -1          */
  DB_UNION = 3,   /* core q: unite with every core point within eps (union-find, atomics)        */
  DB_FLATTEN = 4, /* no ray: core q: label[q] = root of q = smallest core index of its cluster   */
  DB_BORDER = 5   /* q not core: label[q] = smallest root among the core points within eps, or -1 */
};

/* launch parameters */
struct Globals {
  Point *points;
  float eps;
  int minPts;
  int phase;
  int *count;
  unsigned char *core;
  int *parent;
  int *label;
};
