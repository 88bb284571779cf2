/* optix.h -- shim: host-visible OptiX type names that appear in OWL prototypes and in user
 * structs (GeomTypes.h:61 uses OptixTraversableHandle inside RayGenData). */
#ifndef OWL_SHIM_OPTIX_H
#define OWL_SHIM_OPTIX_H
#include <stdint.h>
/* a traversable handle is the device address of an accel descriptor (include/owl/device_runtime.h) */
typedef unsigned long long OptixTraversableHandle;
typedef unsigned int OptixVisibilityMask;
typedef struct owlShimOptixDeviceContext *OptixDeviceContext;
enum {
  OPTIX_RAY_FLAG_NONE = 0u,
  OPTIX_RAY_FLAG_DISABLE_ANYHIT = 1u << 0,
  OPTIX_RAY_FLAG_ENFORCE_ANYHIT = 1u << 1,
  OPTIX_RAY_FLAG_TERMINATE_ON_FIRST_HIT = 1u << 2,
  OPTIX_RAY_FLAG_DISABLE_CLOSESTHIT = 1u << 3
};
#endif
