// owl_device.h -- device-side OWL API for programs compiled by hipcc for gfx950.
//
// Source-compatible with the reference's owl/include/owl/owl_device.h (getProgramData :53-66,
// getPRD :104-114, RayT/Ray :129-147, traceRay/trace :150-202, the OPTIX_*_PROGRAM macros
// :205-256), so samples/s01-trueknn/deviceCode.cu compiles unchanged.  What differs is below the
// surface: a program is a plain device function, reached through a device function pointer the
// macro exports next to it, and a raygen program is wrapped into a HIP kernel that sets up the
// per-thread state the optix* intrinsics read (owl/device_runtime.h).
#pragma once
#if !defined(__HIPCC__)
#error "owl/owl_device.h is device-side: include it from code compiled by hipcc"
#endif
#include <optix_device.h>

#include "owl/common/math/box.h"
#include "owl/common/math/vec.h"
#include "owl/device_runtime.h"

namespace owl {
using namespace owl::common;

inline __device__ vec2i getLaunchIndex() {
  const uint3 i = optixGetLaunchIndex();
  return vec2i((int)i.x, (int)i.y);
}
inline __device__ vec2i getLaunchDims() {
  const uint3 d = optixGetLaunchDimensions();
  return vec2i((int)d.x, (int)d.y);
}
inline __device__ const void *getProgramDataPointer() { return (const void *)optixGetSbtDataPointer(); }
template <typename T>
inline __device__ const T &getProgramData() { return *(const T *)getProgramDataPointer(); }

inline __device__ float linear_to_srgb(float x) { return x <= 0.0031308f ? 12.92f * x : 1.055f * powf(x, 1.f / 2.4f) - 0.055f; }
inline __device__ uint32_t make_8bit(const float f) {
  const int v = (int)(f * 256.f);
  return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
inline __device__ uint32_t make_rgba(const vec3f c) { return make_8bit(c.x) | (make_8bit(c.y) << 8) | (make_8bit(c.z) << 16) | (0xffu << 24); }
inline __device__ uint32_t make_rgba(const vec4f c) { return make_8bit(c.x) | (make_8bit(c.y) << 8) | (make_8bit(c.z) << 16) | (make_8bit(c.w) << 24); }

// per-ray data travels as a pointer split over the two payload registers
static __forceinline__ __device__ void *unpackPointer(uint32_t hi, uint32_t lo) { return (void *)(((uint64_t)hi << 32) | lo); }
static __forceinline__ __device__ void packPointer(void *ptr, uint32_t &hi, uint32_t &lo) {
  const uint64_t u = (uint64_t)ptr;
  hi = (uint32_t)(u >> 32);
  lo = (uint32_t)u;
}
static __forceinline__ __device__ void *getPRDPointer() { return unpackPointer(optixGetPayload_0(), optixGetPayload_1()); }
template <typename T>
static __forceinline__ __device__ T &getPRD() { return *(T *)getPRDPointer(); }

template <int _rayType = 0, int _numRayTypes = 1>
struct RayT {
  enum { rayType = _rayType };
  enum { numRayTypes = _numRayTypes };
  inline __device__ RayT() {}
  inline __device__ RayT(const vec3f &o, const vec3f &d, float t0, float t1) : origin(o), direction(d), tmin(t0), tmax(t1) {}
  vec3f origin, direction;
  float tmin = 0.f, tmax = 1e30f, time = 0.f;
};
typedef RayT<0, 1> Ray;

template <typename RayType, typename PRD>
inline __device__ void traceRay(OptixTraversableHandle traversable, const RayType &ray, PRD &prd, uint32_t rayFlags = 0u) {
  unsigned int p0 = 0, p1 = 0;
  owl::packPointer((void *)&prd, p0, p1);
  optixTrace(traversable, (float3)ray.origin, (float3)ray.direction, ray.tmin, ray.tmax, ray.time,
             (OptixVisibilityMask)-1, rayFlags, ray.rayType, ray.numRayTypes, ray.rayType, p0, p1);
}
template <typename PRD>
inline __device__ void trace(OptixTraversableHandle traversable, const Ray &ray, int numRayTypes, PRD &prd, int sbtOffset = 0) {
  unsigned int p0 = 0, p1 = 0;
  owl::packPointer((void *)&prd, p0, p1);
  optixTrace(traversable, (float3)ray.origin, (float3)ray.direction, ray.tmin, ray.tmax, ray.time,
             (OptixVisibilityMask)-1, 0u, ray.rayType + numRayTypes * sbtOffset, numRayTypes, ray.rayType, p0, p1);
}

}  // namespace owl

// ---- program definition macros -----------------------------------------------------------------
// A callable program `__<kind>__<name>` is a device function plus an exported pointer variable
// `__owl_fp____<kind>__<name>` the host resolves by symbol name after loading the code object.
#define OWL_DEVICE_PROGRAM_(symbol)                                                        \
  extern "C" __device__ void symbol();                                                     \
  extern "C" __device__ owl::device::ProgramFn __owl_fp__##symbol = symbol;                \
  extern "C" __device__ void symbol

#define OPTIX_INTERSECT_PROGRAM(programName) OWL_DEVICE_PROGRAM_(__intersection__##programName)
#define OPTIX_CLOSEST_HIT_PROGRAM(programName) OWL_DEVICE_PROGRAM_(__closesthit__##programName)
#define OPTIX_ANY_HIT_PROGRAM(programName) OWL_DEVICE_PROGRAM_(__anyhit__##programName)
#define OPTIX_MISS_PROGRAM(programName) OWL_DEVICE_PROGRAM_(__miss__##programName)

// A raygen program becomes the kernel `__raygen__<name>(LaunchDesc)`: one thread per launch index.
#define OPTIX_RAYGEN_PROGRAM(programName)                                                          \
  __device__ void __owl_raygen_body__##programName();                                              \
  extern "C" __global__ void __launch_bounds__(OWL_RAYGEN_BLOCK)                                   \
      __raygen__##programName(owl::device::LaunchDesc desc) {                                      \
    if (owl::device::raygen_prologue(desc)) __owl_raygen_body__##programName();                    \
  }                                                                                                \
  __device__ void __owl_raygen_body__##programName

// A bounds program becomes the kernel `__boundsFuncKernel__<name>(geomData, boundsArray, numPrims)`
// (same name and arguments as the reference's generated kernel, owl_device.h:229-256); the host
// launches it over a 1-D grid, thread = primitive.
#define OPTIX_BOUNDS_PROGRAM(progName)                                                             \
  inline __device__ void __boundsFunc__##progName(const void *geomData, owl::common::box3f &bounds,\
                                                  const int32_t primID);                           \
  extern "C" __global__ void __boundsFuncKernel__##progName(const void *geomData,                  \
                                                            owl::common::box3f *const boundsArray, \
                                                            const uint32_t numPrims) {             \
    const uint64_t primID = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;                       \
    if (primID < numPrims) __boundsFunc__##progName(geomData, boundsArray[primID], (int32_t)primID); \
  }                                                                                                \
  inline __device__ void __boundsFunc__##progName
