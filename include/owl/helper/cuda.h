// helper/cuda.h -- error-check macros in the reference's spelling (owl/helper/cuda.h:22-53):
// a failing runtime call throws std::runtime_error.
#pragma once
#include <cuda_runtime.h>

#include <stdexcept>
#include <string>

#define CUDA_CHECK(call)                                                                       \
  do {                                                                                         \
    cudaError_t owl_rc_ = (call);                                                              \
    if (owl_rc_ != cudaSuccess)                                                                \
      throw std::runtime_error(std::string("CUDA_CHECK(" #call ") failed: ") + cudaGetErrorString(owl_rc_)); \
  } while (0)
#define CUDA_CHECK2(msg, call)                                                                 \
  do {                                                                                         \
    cudaError_t owl_rc_ = (call);                                                              \
    if (owl_rc_ != cudaSuccess)                                                                \
      throw std::runtime_error(std::string(msg ? msg : "") + " " #call " failed: " + cudaGetErrorString(owl_rc_)); \
  } while (0)
#define CUDA_CALL(call) CUDA_CHECK(cuda##call)
#define CUDA_SYNC_CHECK() CUDA_CHECK(cudaDeviceSynchronize())
