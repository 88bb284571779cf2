// api_programs.cu -- device programs for the GPU tests of the rest of the OWL surface:
// variable kinds OWL_BUFFER / OWL_BUFFER_SIZE / OWL_DEVICE, host-pinned output, any-hit programs, instance
// transforms and ids, launch parameters of two OWLParams objects.  Written for the tests, not the reference's code.
#include <owl/owl.h>
#include <owl/owl_device_buffer.h>
#include <optix_device.h>

using namespace owl;

struct CubesGeom {
  vec3f *centers;
  float half;
  int reject_odd;  // the any-hit program ignores primitives with an odd index
};
struct ApiParams {
  owl::device::Buffer out;  // OWL_BUFFER: {type, count, data} -- int4 per query: prim, instance id, device, tag
  unsigned long long out_size;  // OWL_BUFFER_SIZE of the same buffer
  float *hit_t;             // host-pinned: distance of the closest accepted hit (-1 on a miss)
  vec3f *origins;
  int tag;                  // differs between the two OWLParams objects of the async test
};
struct ApiRayGen {
  OptixTraversableHandle world;
  int device;  // OWL_DEVICE
  int n;
};
__constant__ ApiParams optixLaunchParams;

struct Hit {
  int prim, inst;
  float t;
};

OPTIX_BOUNDS_PROGRAM(Cubes)(const void *geomData, box3f &bounds, const int primID) {
  const CubesGeom &g = *(const CubesGeom *)geomData;
  const vec3f c = g.centers[primID];
  bounds = box3f(c - g.half, c + g.half);
}

// rays travel along +x: the cube is hit where its x-slab starts, if (y, z) lies inside
OPTIX_INTERSECT_PROGRAM(Cubes)() {
  const CubesGeom &g = owl::getProgramData<CubesGeom>();
  const int prim = optixGetPrimitiveIndex();
  const vec3f o = optixGetObjectRayOrigin();
  const vec3f c = g.centers[prim];
  if (fabsf(c.y - o.y) <= g.half && fabsf(c.z - o.z) <= g.half) optixReportIntersection((c.x - g.half) - o.x, 0, (unsigned)prim);
}

OPTIX_ANY_HIT_PROGRAM(Cubes)() {
  const CubesGeom &g = owl::getProgramData<CubesGeom>();
  if (g.reject_odd && (optixGetAttribute_0() & 1u)) optixIgnoreIntersection();
}

OPTIX_CLOSEST_HIT_PROGRAM(Cubes)() {
  Hit &h = owl::getPRD<Hit>();
  h.prim = (int)optixGetPrimitiveIndex();
  h.inst = (int)optixGetInstanceId();
  h.t = optixGetRayTmax();
}

OPTIX_MISS_PROGRAM(none)() {
  Hit &h = owl::getPRD<Hit>();
  h.prim = -1;
  h.inst = -1;
  h.t = -1.f;
}

OPTIX_RAYGEN_PROGRAM(shoot)() {
  const ApiRayGen &self = owl::getProgramData<ApiRayGen>();
  const int q = optixGetLaunchIndex().x;
  if (q >= self.n) return;
  Hit h = {-2, -2, -2.f};
  owl::Ray ray(optixLaunchParams.origins[q], vec3f(1.f, 0.f, 0.f), 0.f, 1e30f);
  owl::traceRay(self.world, ray, h);
  if ((unsigned long long)q < optixLaunchParams.out.count && optixLaunchParams.out.count == optixLaunchParams.out_size) {
    int4 *out = (int4 *)optixLaunchParams.out.data;
    out[q] = make_int4(h.prim, h.inst, self.device, optixLaunchParams.tag);
  }
  optixLaunchParams.hit_t[q] = h.t;
}
