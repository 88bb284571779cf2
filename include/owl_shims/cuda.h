/* cuda.h -- shim: the handful of CUDA driver-API type names owl_host.h mentions, mapped onto HIP.
 * Strict-C99 clean (no HIP header is pulled in here: hipStream_t IS `struct ihipStream_t *`), so
 * owl/owl.h stays includable from plain C like the reference's header
 * (tests/t00-c99-compliant-header).  Not a CUDA compatibility layer. */
#ifndef OWL_SHIM_CUDA_H
#define OWL_SHIM_CUDA_H
#include <stddef.h>
#include <stdint.h>
typedef struct ihipStream_t *CUstream;
typedef struct ihipStream_t *cudaStream_t;
typedef unsigned long long CUtexObject;
typedef unsigned long long CUdeviceptr;
#endif
