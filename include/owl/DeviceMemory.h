// DeviceMemory.h -- owl::DeviceMemory, a minimal owning wrapper around one HBM allocation
// (alloc / allocManaged / upload / download / free), as used by OWL applications that mix their
// own device buffers with OWL objects (reference: owl/DeviceMemory.h:23-113).
#pragma once
#include <cassert>
#include <vector>

#include "owl/helper/cuda.h"

namespace owl {

struct DeviceMemory {
  size_t sizeInBytes = 0;
  CUdeviceptr d_pointer = 0;

  DeviceMemory() = default;
  DeviceMemory(const DeviceMemory &) = delete;
  DeviceMemory &operator=(const DeviceMemory &) = delete;
  ~DeviceMemory() {
    if (d_pointer) (void)cudaFree((void *)d_pointer);
  }

  bool empty() const { return sizeInBytes == 0; }
  bool alloced() const { return !empty(); }
  bool notEmpty() const { return !empty(); }
  size_t size() const { return sizeInBytes; }
  void *get() { return (void *)d_pointer; }

  void alloc(size_t bytes) { reserve(bytes, false); }
  void allocManaged(size_t bytes) { reserve(bytes, true); }
  void free() {
    if (d_pointer) CUDA_CHECK(cudaFree((void *)d_pointer));
    d_pointer = 0;
    sizeInBytes = 0;
  }
  void upload(const void *host, const char *what = nullptr) {
    CUDA_CHECK2(what, cudaMemcpy((void *)d_pointer, host, sizeInBytes, cudaMemcpyHostToDevice));
  }
  void uploadAsync(const void *host, cudaStream_t stream) {
    CUDA_CHECK(cudaMemcpyAsync((void *)d_pointer, host, sizeInBytes, cudaMemcpyHostToDevice, stream));
  }
  void download(void *host) { CUDA_CHECK(cudaMemcpy(host, (void *)d_pointer, sizeInBytes, cudaMemcpyDeviceToHost)); }
  template <typename T>
  void upload(const std::vector<T> &v) {
    if (v.size() * sizeof(T) != sizeInBytes) reserve(v.size() * sizeof(T), false);
    upload((const void *)v.data());
  }

 private:
  void reserve(size_t bytes, bool managed) {
    free();
    if (bytes == 0) return;
    void *p = nullptr;
    CUDA_CHECK(managed ? cudaMallocManaged(&p, bytes) : cudaMalloc(&p, bytes));
    d_pointer = (CUdeviceptr)p;
    sizeInBytes = bytes;
  }
};

}  // namespace owl
