// How many one-wave workgroups a CU holds as a function of their LDS size: what the runtime's occupancy query says, and what
// a launch shows (every workgroup spins for 50 us; 256 CUs x 96 workgroups take 96 / resident x 50 us).
//   hipcc --offload-arch=gfx950 -O2 scripts/microbench/lds_granule.hip -o build/lds_granule && build/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
extern __shared__ int dyn[];
__global__ void __launch_bounds__(64) probe(int *out, long long ticks) {
  dyn[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (out) out[threadIdx.x] = dyn[63 - threadIdx.x];
}
int main() {
  CK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  int prev = -1;
  for (int bytes = 4096; bytes <= 40960; bytes += 16) {
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe, 64, bytes));
    if (nb != prev) printf("occupancy query: from %6d bytes %d workgroups per CU\n", bytes, nb);
    prev = nb;
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const long long ticks = (long long)rate_khz * 50 / 1000;  // 50 us
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int per_cu = 96, grid = prop.multiProcessorCount * per_cu;
  const int sizes[] = {6400, 6416, 7680, 7696, 8192, 8208, 8960, 8976, 9728, 10240, 10256, 11520, 11536, 12112, 12800, 12816, 14080, 14096,
                       15360, 15376, 16384, 16400, 16640, 16656, 17920, 17936};
  for (int bytes : sizes) {
    hipLaunchKernelGGL(probe, dim3(grid), dim3(64), bytes, 0, (int *)nullptr, ticks);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(probe, dim3(grid), dim3(64), bytes, 0, (int *)nullptr, ticks);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("launch: %6d bytes  %.3f ms  = %.2f turns of 50 us  => about %.1f workgroups resident per CU\n", bytes, ms, ms / 0.05, per_cu / (ms / 0.05));
  }
  return 0;
}
