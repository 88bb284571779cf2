// deviceCode.cu -- device programs of sample02-rtdbscan: "distance computations on the ray-tracing
// side, the other clustering operations in shader code" (the reference's README.md:9 on RT-DBSCAN).
//
// Spec (oracle/dbscan_oracle.c, = sklearn.cluster.DBSCAN's labelling): N(p) = {q : dist <= eps},
// p included; core iff |N(p)| >= minPts; clusters = components of core points, numbered by ascending
// smallest core index; a border point joins the lowest-numbered adjacent cluster; noise = -1.
// Distance arithmetic: sqrt((x*x + y*y) + z*z) as in samples/s01-trueknn/deviceCode.cu:110-113.
#include "GeomTypes.h"
#include <optix_device.h>

__constant__ Globals optixLaunchParams;

OPTIX_BOUNDS_PROGRAM(Points)(const void *geomData, box3f &primBounds, const int primID) {
  const PointsGeom &self = *(const PointsGeom *)geomData;
  const vec3f c = self.prims[primID].center;
  primBounds = box3f().extend(c - self.eps).extend(c + self.eps);
}

// union-find over point indices; the smaller index stays root, so a cluster's root is its smallest
// core index.  Every read of a parent pointer is an atomic at the memory side (another wave may have
// hooked the root a moment ago); path halving, lock-free hooking by compare-and-swap.
__device__ inline int ufLoad(int *p) { return atomicAdd(p, 0); }
__device__ inline int ufFind(int *parent, int x) {
  for (;;) {
    const int p = ufLoad(parent + x);
    if (p == x) return x;
    const int g = ufLoad(parent + p);
    if (g == p) return p;
    atomicMin(parent + x, g);  // halve the path (parents only ever decrease)
    x = g;
  }
}
__device__ inline void ufUnite(int *parent, int a, int b) {
  for (;;) {
    a = ufFind(parent, a);
    b = ufFind(parent, b);
    if (a == b) return;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    if (atomicCAS(parent + hi, hi, lo) == hi) return;  // fails if `hi` stopped being a root meanwhile
  }
}

OPTIX_INTERSECT_PROGRAM(Points)() {
  const int p = optixGetPrimitiveIndex();
  const int q = optixGetLaunchIndex().x;
  const Globals &lp = optixLaunchParams;
  const PointsGeom &self = owl::getProgramData<PointsGeom>();
  const vec3f org = optixGetWorldRayOrigin();
  const vec3f c = self.prims[p].center;
  const float x = c.x - org.x, y = c.y - org.y, z = c.z - org.z;
  const float distance = std::sqrt((x * x) + (y * y) + (z * z));
  if (!(distance <= lp.eps)) return;  // inside the box, outside the sphere
  if (lp.phase == DB_COUNT) {
    lp.count[q] += 1;  // (q itself is among its candidates: |N(q)| counts it)
  } else if (lp.phase == DB_UNION) {
    if (p != q && lp.core[p]) ufUnite(lp.parent, q, p);
  } else if (lp.phase == DB_BORDER) {
    if (lp.core[p]) {
      const int root = ufFind(lp.parent, p);
      if (lp.label[q] < 0 || root < lp.label[q]) lp.label[q] = root;
    }
  }
}

OPTIX_RAYGEN_PROGRAM(rayGen)() {
  const RayGenData &self = owl::getProgramData<RayGenData>();
  const Globals &lp = optixLaunchParams;
  const int q = optixGetLaunchIndex().x;
  if (q >= self.n) return;
  if (lp.phase == DB_MARK) {
    lp.core[q] = lp.count[q] >= lp.minPts ? 1 : 0;
    lp.parent[q] = q;
    lp.label[q] = -1;
    return;
  }
  if (lp.phase == DB_FLATTEN) {
    if (lp.core[q]) lp.label[q] = ufFind(lp.parent, q);
    return;
  }
  if (lp.phase == DB_UNION && !lp.core[q]) return;
  if (lp.phase == DB_BORDER && lp.core[q]) return;
  // a point query: the ray's extent is below fp32 resolution (samples/s01-trueknn/deviceCode.cu:140-153)
  owl::Ray ray(lp.points[q].center, vec3f(0, 0, 1), 0.f, 1.e-16f);
  int unused = 0;
  owl::traceRay(self.world, ray, unused);
}
