// device_runtime.h -- the device half of the OWL program model on MI355X (gfx950).
//
// On NVIDIA the reference's device programs run inside OptiX: `optixTrace` hands a ray to the RT
// cores, which walk the driver's BVH and call back the `__intersection__*` program bound through
// the shader binding table; the `optixGet*` intrinsics read per-ray state held by the hardware
// (reference: owl/include/owl/owl_device.h:53-66,129-174; samples/s01-trueknn/deviceCode.cu).
// Here the same names are ordinary inline HIP device functions:
//   * per-thread ray state lives in LDS (struct-of-arrays, one column per thread of the block);
//   * `optixTrace` is a stackless rope walk of the LBVH of include/owl/lbvh_device.h with a slab
//     ray/AABB test, one ray per lane;
//   * programs are reached through device function pointers stored in the geometry records
//     (the SBT): the OPTIX_*_PROGRAM macros of owl_device.h export one pointer variable per
//     program, the host reads it after loading the code object and writes it into the record.
// Everything is header-inline so a user's deviceCode.cu compiles to ONE gfx950 code object.
#pragma once
#if !defined(__HIPCC__)
#error "owl/device_runtime.h is device code: compile with hipcc --offload-arch=gfx950"
#endif
#include <hip/hip_runtime.h>
#include <optix.h>
#include <stdint.h>

#include "owl/lbvh_device.h"

#ifndef OWL_RAYGEN_BLOCK
#define OWL_RAYGEN_BLOCK 256 /* threads per raygen workgroup; the host launches with this size */
#endif
#define OWL_MAX_RAY_TYPES 4

namespace owl {
namespace device {

typedef void (*ProgramFn)();

// ---- records the host writes (layouts mirrored in owlraytracing_amd/csrc/owl_runtime.cpp) ----
struct GeomRecord {  // one per geometry of a user-geom group ("hit group record")
  uint64_t intersect[OWL_MAX_RAY_TYPES];    // device address of __intersection__<name>, 0 = none
  uint64_t closest_hit[OWL_MAX_RAY_TYPES];  // device address of __closesthit__<name>
  uint64_t any_hit[OWL_MAX_RAY_TYPES];
  const void *data;     // the geometry's variable struct (what getProgramData<T>() returns)
  uint32_t prim_begin;  // first primitive of this geometry in the group's numbering
  uint32_t prim_count;
};
enum { ACCEL_USER_GROUP = 1, ACCEL_INSTANCE_GROUP = 2 };
struct AccelHeader {
  uint32_t kind;
  uint32_t count;  // geometries (user group) or instances (instance group)
};
struct UserGroupAccel {
  AccelHeader h;
  LbvhView bvh;  // box flavour: boxes[] are what the bounds program wrote, in Morton order
  const GeomRecord *geoms;
};
struct Instance {
  float o2w[12];  // object-to-world, row-major 3x4
  float w2o[12];
  uint64_t child;  // UserGroupAccel*
  uint32_t instance_id;
  uint32_t identity;  // 1: both transforms are the identity
};
struct InstanceGroupAccel {
  AccelHeader h;
  const Instance *instances;
};
struct MissRecord {
  uint64_t prog;
  const void *data;
};
struct LaunchDesc {  // the single argument of every __raygen__ kernel
  uint32_t dims[3];
  uint32_t num_miss;
  const void *raygen_data;
  const MissRecord *miss;
  // null, or a permutation of the launch indices (1-D launches): thread t runs launch index order[t].  OptiX promises no
  // order among the indices of a launch; the host hands out the Morton order of the traced geometry's primitives when a launch
  // has as many indices as that geometry has primitives (index i = "the query at primitive i" in the neighbour-query
  // programs this backend is for), so that the threads of a wave walk the same part of the tree
  const int32_t *order;
};

// ---- per-thread state in LDS --------------------------------------------------------------
struct BlockState {
  LaunchDesc desc;
  float org[3][OWL_RAYGEN_BLOCK], dir[3][OWL_RAYGEN_BLOCK];          // world-space ray
  float oorg[3][OWL_RAYGEN_BLOCK], odir[3][OWL_RAYGEN_BLOCK];        // object-space ray
  float tmin[OWL_RAYGEN_BLOCK], tmax[OWL_RAYGEN_BLOCK], time[OWL_RAYGEN_BLOCK];
  uint32_t payload[2][OWL_RAYGEN_BLOCK];
  uint32_t prim[OWL_RAYGEN_BLOCK];
  uint32_t sbt_lo[OWL_RAYGEN_BLOCK], sbt_hi[OWL_RAYGEN_BLOCK];      // current program's data pointer
  uint32_t inst_id[OWL_RAYGEN_BLOCK], inst_index[OWL_RAYGEN_BLOCK];
  uint32_t hit_kind[OWL_RAYGEN_BLOCK], hit_attr[2][OWL_RAYGEN_BLOCK];
  uint32_t flags[OWL_RAYGEN_BLOCK];  // bit 0: terminate requested, bit 1: a hit is recorded, bit 2: the any-hit program ignored the candidate
  uint32_t anyhit_lo[OWL_RAYGEN_BLOCK], anyhit_hi[OWL_RAYGEN_BLOCK];  // __anyhit__ program of the geometry whose intersection program runs (0: none)
};
static __shared__ BlockState owl_block_state;

}  // namespace device
}  // namespace owl
// The layout of what the host hands to device code (LaunchDesc, the records) is part of this header AND of libowl_mi355x.so:
// a code object built with one and launched by the other would read its arguments at the wrong offsets.  The code
// object carries the header's layout word, the host compares it with its own when it loads the module (owlBuildPrograms) and
// refuses a mismatch.  High 16 bits: revision of this header's host-visible structs; low 16: sizeof(LaunchDesc).
#ifndef OWL_MI355X_DEVICE_ABI /* (tests build a module with another word to see it refused) */
#define OWL_MI355X_DEVICE_ABI ((2u << 16) | (unsigned)sizeof(owl::device::LaunchDesc))
#endif
extern "C" __attribute__((weak, used, visibility("default"))) __device__ const unsigned owl_mi355x_device_abi = OWL_MI355X_DEVICE_ABI;
namespace owl {
namespace device {

__device__ __forceinline__ BlockState &state() { return owl_block_state; }
__device__ __forceinline__ uint32_t tid() {
  return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
}
__device__ __forceinline__ void set_sbt(const void *p) {
  const uint64_t u = (uint64_t)p;
  state().sbt_lo[tid()] = (uint32_t)u;
  state().sbt_hi[tid()] = (uint32_t)(u >> 32);
}

// called at the top of every generated __raygen__ kernel; returns false for padding threads
__device__ __forceinline__ bool raygen_prologue(const LaunchDesc &desc) {
  if (tid() == 0) state().desc = desc;
  __syncthreads();
  const uint64_t total = (uint64_t)desc.dims[0] * desc.dims[1] * desc.dims[2];
  const uint64_t lin = (uint64_t)blockIdx.x * OWL_RAYGEN_BLOCK + tid();
  set_sbt(desc.raygen_data);
  state().flags[tid()] = 0;
  return lin < total;
}

}  // namespace device
}  // namespace owl

// ---- the optix* intrinsics -------------------------------------------------------------------
__device__ __forceinline__ uint3 optixGetLaunchDimensions() {
  const owl::device::LaunchDesc &d = owl::device::state().desc;
  return make_uint3(d.dims[0], d.dims[1], d.dims[2]);
}
__device__ __forceinline__ uint3 optixGetLaunchIndex() {
  const owl::device::LaunchDesc &d = owl::device::state().desc;
  uint64_t lin = (uint64_t)blockIdx.x * OWL_RAYGEN_BLOCK + owl::device::tid();
  if (d.order) lin = (uint64_t)(uint32_t)d.order[lin];
  const uint32_t x = (uint32_t)(lin % d.dims[0]);
  const uint64_t rest = lin / d.dims[0];
  return make_uint3(x, (uint32_t)(rest % d.dims[1]), (uint32_t)(rest / d.dims[1]));
}
__device__ __forceinline__ unsigned long long optixGetSbtDataPointer() {
  const uint32_t t = owl::device::tid();
  return ((unsigned long long)owl::device::state().sbt_hi[t] << 32) | owl::device::state().sbt_lo[t];
}
__device__ __forceinline__ unsigned int optixGetPrimitiveIndex() { return owl::device::state().prim[owl::device::tid()]; }
__device__ __forceinline__ unsigned int optixGetInstanceId() { return owl::device::state().inst_id[owl::device::tid()]; }
__device__ __forceinline__ unsigned int optixGetInstanceIndex() { return owl::device::state().inst_index[owl::device::tid()]; }
__device__ __forceinline__ float3 optixGetWorldRayOrigin() {
  const uint32_t t = owl::device::tid();
  return make_float3(owl::device::state().org[0][t], owl::device::state().org[1][t], owl::device::state().org[2][t]);
}
__device__ __forceinline__ float3 optixGetWorldRayDirection() {
  const uint32_t t = owl::device::tid();
  return make_float3(owl::device::state().dir[0][t], owl::device::state().dir[1][t], owl::device::state().dir[2][t]);
}
__device__ __forceinline__ float3 optixGetObjectRayOrigin() {
  const uint32_t t = owl::device::tid();
  return make_float3(owl::device::state().oorg[0][t], owl::device::state().oorg[1][t], owl::device::state().oorg[2][t]);
}
__device__ __forceinline__ float3 optixGetObjectRayDirection() {
  const uint32_t t = owl::device::tid();
  return make_float3(owl::device::state().odir[0][t], owl::device::state().odir[1][t], owl::device::state().odir[2][t]);
}
__device__ __forceinline__ float optixGetRayTmin() { return owl::device::state().tmin[owl::device::tid()]; }
__device__ __forceinline__ float optixGetRayTmax() { return owl::device::state().tmax[owl::device::tid()]; }
__device__ __forceinline__ float optixGetRayTime() { return owl::device::state().time[owl::device::tid()]; }
__device__ __forceinline__ unsigned int optixGetPayload_0() { return owl::device::state().payload[0][owl::device::tid()]; }
__device__ __forceinline__ unsigned int optixGetPayload_1() { return owl::device::state().payload[1][owl::device::tid()]; }
__device__ __forceinline__ void optixSetPayload_0(unsigned int v) { owl::device::state().payload[0][owl::device::tid()] = v; }
__device__ __forceinline__ void optixSetPayload_1(unsigned int v) { owl::device::state().payload[1][owl::device::tid()] = v; }
__device__ __forceinline__ unsigned int optixGetHitKind() { return owl::device::state().hit_kind[owl::device::tid()]; }
__device__ __forceinline__ unsigned int optixGetAttribute_0() { return owl::device::state().hit_attr[0][owl::device::tid()]; }
__device__ __forceinline__ unsigned int optixGetAttribute_1() { return owl::device::state().hit_attr[1][owl::device::tid()]; }
__device__ __forceinline__ void optixTerminateRay() { owl::device::state().flags[owl::device::tid()] |= 1u; }
// any-hit programs: reject the candidate the intersection program has just reported (the ray keeps its tmax)
__device__ __forceinline__ void optixIgnoreIntersection() { owl::device::state().flags[owl::device::tid()] |= 4u; }

// An intersection program reports a hit at parameter t: a candidate if inside (tmin, tmax) of the current
// ray.  The geometry's any-hit program, if it has one for this ray type, then sees the candidate (t as the
// ray's tmax, hit kind and attributes) and may reject it with optixIgnoreIntersection or accept it and end
// the traversal with optixTerminateRay; an accepted candidate shortens the ray to t (closest-hit
// semantics).  Returns whether the hit was accepted, like OptiX.
__device__ __forceinline__ bool optixReportIntersection(float t, unsigned int kind, unsigned int a0 = 0,
                                                        unsigned int a1 = 0) {
  const uint32_t i = owl::device::tid();
  owl::device::BlockState &s = owl::device::state();
  if (!(t > s.tmin[i]) || !(t < s.tmax[i])) return false;
  const float old_tmax = s.tmax[i];
  const uint32_t old_kind = s.hit_kind[i], old_a0 = s.hit_attr[0][i], old_a1 = s.hit_attr[1][i];
  s.tmax[i] = t;
  s.hit_kind[i] = kind;
  s.hit_attr[0][i] = a0;
  s.hit_attr[1][i] = a1;
  const uint64_t any_hit = ((uint64_t)s.anyhit_hi[i] << 32) | s.anyhit_lo[i];
  if (any_hit) {
    s.flags[i] &= ~4u;
    ((owl::device::ProgramFn)any_hit)();
    if (s.flags[i] & 4u) {  // ignored: as if nothing had been reported
      s.flags[i] &= ~4u;
      s.tmax[i] = old_tmax;
      s.hit_kind[i] = old_kind;
      s.hit_attr[0][i] = old_a0;
      s.hit_attr[1][i] = old_a1;
      return false;
    }
  }
  s.flags[i] |= 2u;  // a hit is recorded
  return true;
}

namespace owl {
namespace device {

// slab test of the segment o + t*d, t in [t0, t1], against the closed box [lo, hi]
__device__ __forceinline__ bool ray_hits_box(const float *lo, const float *hi, const float o[3], const float d[3],
                                             float t0, float t1) {
#pragma clang fp contract(off)
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (d[a] == 0.f) {
      if (!(lo[a] <= o[a] && o[a] <= hi[a])) return false;
    } else {
      float ta = (lo[a] - o[a]) / d[a], tb = (hi[a] - o[a]) / d[a];
      float tn = fminf(ta, tb), tf = fmaxf(ta, tb);
      t0 = fmaxf(t0, tn);
      t1 = fminf(t1, tf);
      if (!(t0 <= t1)) return false;
    }
  }
  return true;
}

// walk one user-geometry group with the ray currently stored in the thread's object-space slots
struct HitRecord {  // what the closest-hit program must see again after traversal has moved on
  uint64_t geom;    // GeomRecord* of the accepted hit, 0 = none
  uint32_t prim, inst_id, inst_index;
};

__device__ __forceinline__ void trace_user_group(const UserGroupAccel *accel, unsigned ray_type, HitRecord *hit) {
  BlockState &s = state();
  const uint32_t i = tid();
  const LbvhView &bvh = accel->bvh;
  const float o[3] = {s.oorg[0][i], s.oorg[1][i], s.oorg[2][i]};
  const float d[3] = {s.odir[0][i], s.odir[1][i], s.odir[2][i]};
  const float t0 = s.tmin[i];
  int32_t ref = bvh.root;
  while (ref != LBVH_END) {
    if (s.flags[i] & 1u) break;  // optixTerminateRay
    if (ref >= 0) {
      const LbvhNode nd = bvh.nodes[ref];
      ref = ray_hits_box(nd.lo, nd.hi, o, d, t0, s.tmax[i]) ? lbvh_left_ref(ref, nd) : bvh.rope_node[ref];
    } else {
      const int32_t slot = ~ref;
      const LbvhBox b = bvh.boxes[slot];
      if (ray_hits_box(b.lo, b.hi, o, d, t0, s.tmax[i])) {
        const uint32_t prim = (uint32_t)bvh.prim_id[slot];
        uint32_t g = 0;
        while (g + 1 < accel->h.count && prim >= accel->geoms[g + 1].prim_begin) g++;
        const GeomRecord &rec = accel->geoms[g];
        const uint64_t fn = rec.intersect[ray_type < OWL_MAX_RAY_TYPES ? ray_type : 0];
        if (fn) {
          s.prim[i] = prim - rec.prim_begin;
          set_sbt(rec.data);
          const uint64_t ah = rec.any_hit[ray_type < OWL_MAX_RAY_TYPES ? ray_type : 0];
          s.anyhit_lo[i] = (uint32_t)ah;
          s.anyhit_hi[i] = (uint32_t)(ah >> 32);
          const uint32_t had = s.flags[i] & 2u;
          s.flags[i] &= ~2u;
          ((ProgramFn)fn)();
          if (s.flags[i] & 2u) {  // accepted: closest so far
            hit->geom = (uint64_t)&rec;
            hit->prim = s.prim[i];
            hit->inst_id = s.inst_id[i];
            hit->inst_index = s.inst_index[i];
          } else {
            s.flags[i] |= had;
          }
        }
      }
      ref = bvh.rope_leaf[slot];
    }
  }
}

__device__ __forceinline__ void xfm_point(const float m[12], const float p[3], float out[3]) {
#pragma unroll
  for (int r = 0; r < 3; r++) out[r] = m[4 * r] * p[0] + m[4 * r + 1] * p[1] + m[4 * r + 2] * p[2] + m[4 * r + 3];
}
__device__ __forceinline__ void xfm_vector(const float m[12], const float v[3], float out[3]) {
#pragma unroll
  for (int r = 0; r < 3; r++) out[r] = m[4 * r] * v[0] + m[4 * r + 1] * v[1] + m[4 * r + 2] * v[2];
}

}  // namespace device
}  // namespace owl

// optixTrace: same parameter list as OptiX 7 (reference call site owl_device.h:161-173).
__device__ __forceinline__ void optixTrace(OptixTraversableHandle handle, float3 rayOrigin, float3 rayDirection,
                                           float tmin, float tmax, float rayTime, OptixVisibilityMask /*mask*/,
                                           unsigned int /*rayFlags*/, unsigned int SBToffset,
                                           unsigned int /*SBTstride*/, unsigned int missSBTIndex, unsigned int &p0,
                                           unsigned int &p1) {
  using namespace owl::device;
  BlockState &s = state();
  const uint32_t i = tid();
  const uint64_t caller_sbt = optixGetSbtDataPointer();
  s.org[0][i] = rayOrigin.x;
  s.org[1][i] = rayOrigin.y;
  s.org[2][i] = rayOrigin.z;
  s.dir[0][i] = rayDirection.x;
  s.dir[1][i] = rayDirection.y;
  s.dir[2][i] = rayDirection.z;
  s.tmin[i] = tmin;
  s.tmax[i] = tmax;
  s.time[i] = rayTime;
  s.payload[0][i] = p0;
  s.payload[1][i] = p1;
  s.flags[i] = 0;
  HitRecord hit = {0, 0, 0, 0};
  const AccelHeader *hdr = (const AccelHeader *)handle;
  if (hdr) {
    if (hdr->kind == ACCEL_USER_GROUP) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        s.oorg[a][i] = s.org[a][i];
        s.odir[a][i] = s.dir[a][i];
      }
      s.inst_id[i] = 0;
      s.inst_index[i] = 0;
      trace_user_group((const UserGroupAccel *)hdr, SBToffset, &hit);
    } else if (hdr->kind == ACCEL_INSTANCE_GROUP) {
      const InstanceGroupAccel *ig = (const InstanceGroupAccel *)hdr;
      for (uint32_t n = 0; n < hdr->count && !(s.flags[i] & 1u); n++) {
        const Instance &inst = ig->instances[n];
        const AccelHeader *child = (const AccelHeader *)inst.child;
        if (!child || child->kind != ACCEL_USER_GROUP) continue;  // one instancing level (OWL's default depth)
        const float wo[3] = {s.org[0][i], s.org[1][i], s.org[2][i]};
        const float wd[3] = {s.dir[0][i], s.dir[1][i], s.dir[2][i]};
        float oo[3] = {wo[0], wo[1], wo[2]}, od[3] = {wd[0], wd[1], wd[2]};
        if (!inst.identity) {
          xfm_point(inst.w2o, wo, oo);
          xfm_vector(inst.w2o, wd, od);
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
          s.oorg[a][i] = oo[a];
          s.odir[a][i] = od[a];
        }
        s.inst_id[i] = inst.instance_id;
        s.inst_index[i] = n;
        trace_user_group((const UserGroupAccel *)child, SBToffset, &hit);
      }
    }
  }
  // closest-hit or miss program, with the SBT pointer of that program
  if (hit.geom) {
    const GeomRecord *rec = (const GeomRecord *)hit.geom;
    const uint64_t fn = rec->closest_hit[SBToffset < OWL_MAX_RAY_TYPES ? SBToffset : 0];
    if (fn) {
      s.prim[i] = hit.prim;
      s.inst_id[i] = hit.inst_id;
      s.inst_index[i] = hit.inst_index;
      set_sbt(rec->data);
      ((ProgramFn)fn)();
    }
  } else if (missSBTIndex < s.desc.num_miss && s.desc.miss) {
    const MissRecord &m = s.desc.miss[missSBTIndex];
    if (m.prog) {
      set_sbt(m.data);
      ((ProgramFn)m.prog)();
    }
  }
  p0 = s.payload[0][i];
  p1 = s.payload[1][i];
  set_sbt((const void *)caller_sbt);
}
