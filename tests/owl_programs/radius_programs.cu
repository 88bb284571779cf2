// radius_programs.cu -- device programs for the GPU tests of the OWL program model.
// Not the reference's TrueKNN program: a fixed-radius neighbour COUNT with a true sphere test (the
// RT-DBSCAN phase-1 pattern) plus a closest-hit / miss pair, so that
// bounds, intersect, closest-hit, miss, launch params, per-ray data and optixReportIntersection
// all run on the MI355X runtime.
#include <owl/owl.h>
#include <optix_device.h>

using namespace owl;

struct BallsGeom {
  vec3f *centers;
  float radius;
};
struct CountParams {
  int *count;        // per query: points with |c - q| <= radius, self excluded
  float *nearest;    // per query: smallest such distance (inf if none)
  long long *calls;  // per query: intersection-program invocations (box candidates)
  vec3f *queries;
  int first_hit_mode;  // 0: point queries counting neighbours, 1: +z rays reporting the first ball hit
  int *first_hit;      // per query (mode 1): primitive id of the closest hit, -1 on miss
};
struct CountRayGen {
  OptixTraversableHandle world;
  int n_queries;
};
__constant__ CountParams optixLaunchParams;

struct PerRay {
  int hit;
};

OPTIX_BOUNDS_PROGRAM(Balls)(const void *geomData, box3f &bounds, const int primID) {
  const BallsGeom &g = *(const BallsGeom *)geomData;
  const vec3f c = g.centers[primID];
  bounds = box3f(c - g.radius, c + g.radius);
}

OPTIX_INTERSECT_PROGRAM(Balls)() {
  const int prim = optixGetPrimitiveIndex();
  const int q = optixGetLaunchIndex().x;
  const BallsGeom &g = owl::getProgramData<BallsGeom>();
  const vec3f o = optixGetWorldRayOrigin();
  const vec3f c = g.centers[prim];
  if (optixLaunchParams.first_hit_mode) {
    // ray along +z from o: hits the ball's axis-aligned "front" if (x,y) is inside the disc
    const float dx = c.x - o.x, dy = c.y - o.y;
    if (dx * dx + dy * dy <= g.radius * g.radius) optixReportIntersection(c.z - o.z, 0);
    return;
  }
  optixLaunchParams.calls[q] += 1;
  if (prim == q) return;
  const float x = c.x - o.x, y = c.y - o.y, z = c.z - o.z;
  const float d = sqrtf((x * x) + (y * y) + (z * z));
  if (d <= g.radius) {
    optixLaunchParams.count[q] += 1;
    if (d < optixLaunchParams.nearest[q]) optixLaunchParams.nearest[q] = d;
  }
}

OPTIX_CLOSEST_HIT_PROGRAM(Balls)() { owl::getPRD<PerRay>().hit = (int)optixGetPrimitiveIndex(); }

OPTIX_MISS_PROGRAM(nothing)() { owl::getPRD<PerRay>().hit = -1; }

OPTIX_RAYGEN_PROGRAM(queries)() {
  const CountRayGen &self = owl::getProgramData<CountRayGen>();
  const int q = optixGetLaunchIndex().x;
  if (q >= self.n_queries) return;
  PerRay prd;
  prd.hit = -2;
  if (optixLaunchParams.first_hit_mode) {
    owl::Ray ray(optixLaunchParams.queries[q], vec3f(0.f, 0.f, 1.f), 0.f, 1e30f);
    owl::traceRay(self.world, ray, prd);
    optixLaunchParams.first_hit[q] = prd.hit;
  } else {
    owl::Ray ray(optixLaunchParams.queries[q], vec3f(0.f, 0.f, 1.f), 0.f, 1.e-16f);
    owl::traceRay(self.world, ray, prd);
  }
}
