/* cuda_runtime.h -- shim: the CUDA runtime calls OWL host code uses, as inline forwards to HIP
 * (samples/s01-trueknn/hostCode.cpp:291,293 calls cudaDeviceSynchronize). */
#ifndef OWL_SHIM_CUDA_RUNTIME_H
#define OWL_SHIM_CUDA_RUNTIME_H
#include <cuda.h>
#include <driver_types.h>
#if !defined(__HIP_PLATFORM_AMD__)
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
#define cudaMemcpyDefault hipMemcpyDefault
#define cudaMemcpyKind hipMemcpyKind
static inline cudaError_t cudaDeviceSynchronize(void) { return hipDeviceSynchronize(); }
static inline cudaError_t cudaStreamSynchronize(cudaStream_t s) { return hipStreamSynchronize(s); }
static inline cudaError_t cudaMalloc(void **p, size_t n) { return hipMalloc(p, n); }
static inline cudaError_t cudaMallocManaged(void **p, size_t n) { return hipMallocManaged(p, n, hipMemAttachGlobal); }
static inline cudaError_t cudaFree(void *p) { return hipFree(p); }
static inline cudaError_t cudaMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) { return hipMemcpy(d, s, n, k); }
static inline cudaError_t cudaMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, cudaStream_t st) {
  return hipMemcpyAsync(d, s, n, k, st);
}
static inline cudaError_t cudaMemset(void *d, int v, size_t n) { return hipMemset(d, v, n); }
static inline cudaError_t cudaGetLastError(void) { return hipGetLastError(); }
static inline const char *cudaGetErrorString(cudaError_t e) { return hipGetErrorString(e); }
static inline cudaError_t cudaSetDevice(int d) { return hipSetDevice(d); }
static inline cudaError_t cudaGetDevice(int *d) { return hipGetDevice(d); }
static inline cudaError_t cudaGetDeviceCount(int *n) { return hipGetDeviceCount(n); }
#endif
