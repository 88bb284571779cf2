// issue_rate.hip -- developer microbenchmark (not part of the library): how many wave64 instructions per
// cycle one SIMD of gfx950 issues, per instruction kind and per waves/SIMD.  Settles what "VALU busy"
// means for trueknn_team.hip.   hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 32;

template <int KIND>
__global__ void __launch_bounds__(64) probe(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long m0 = 0x5555555555555555ull, m1 = 0x3333333333333333ull;
  unsigned int s0 = 1, s1 = 2;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < kUnroll / 8; u++) {
      if (KIND == 0) {  // independent v_add_f32
        asm volatile("v_add_f32 %0, %0, %0\nv_add_f32 %1, %1, %1\nv_add_f32 %2, %2, %2\nv_add_f32 %3, %3, %3\n"
                     "v_add_f32 %4, %4, %4\nv_add_f32 %5, %5, %5\nv_add_f32 %6, %6, %6\nv_add_f32 %7, %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (KIND == 1) {  // dependent chain of v_add_f32
        asm volatile("v_add_f32 %0, %0, %0\nv_add_f32 %0, %0, %0\nv_add_f32 %0, %0, %0\nv_add_f32 %0, %0, %0\n"
                     "v_add_f32 %0, %0, %0\nv_add_f32 %0, %0, %0\nv_add_f32 %0, %0, %0\nv_add_f32 %0, %0, %0"
                     : "+v"(a0));
      } else if (KIND == 2) {  // v_cmp writing an SGPR pair + v_cndmask reading it (the kernel's predicate style)
        asm volatile("v_cmp_le_f32 %8, %0, %1\nv_cndmask_b32 %2, %2, %3, %8\nv_cmp_le_f32 %9, %4, %5\nv_cndmask_b32 %6, %6, %7, %9\n"
                     "v_cmp_le_f32 %8, %1, %0\nv_cndmask_b32 %3, %3, %2, %8\nv_cmp_le_f32 %9, %5, %4\nv_cndmask_b32 %7, %7, %6, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(m0), "+s"(m1));
      } else if (KIND == 3) {  // DPP moves
        asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (KIND == 4) {  // 64-bit unsigned compares
        asm volatile("v_cmp_lt_u64 %8, %0, %2\nv_cmp_lt_u64 %9, %4, %6\nv_cmp_lt_u64 %8, %2, %0\nv_cmp_lt_u64 %9, %6, %4\n"
                     "v_cmp_lt_u64 %8, %0, %4\nv_cmp_lt_u64 %9, %2, %6\nv_cmp_lt_u64 %8, %4, %0\nv_cmp_lt_u64 %9, %6, %2"
                     : "+v"(*(unsigned long long *)&a0), "+v"(a1), "+v"(*(unsigned long long *)&a2), "+v"(a3), "+v"(*(unsigned long long *)&a4), "+v"(a5),
                       "+v"(*(unsigned long long *)&a6), "+v"(a7), "+s"(m0), "+s"(m1));
      } else if (KIND == 5) {  // SALU only
        asm volatile("s_add_u32 %0, %0, %1\ns_and_b64 %2, %2, %3\ns_add_u32 %1, %1, %0\ns_or_b64 %3, %3, %2\n"
                     "s_add_u32 %0, %0, %1\ns_and_b64 %2, %2, %3\ns_add_u32 %1, %1, %0\ns_or_b64 %3, %3, %2"
                     : "+s"(s0), "+s"(s1), "+s"(m0), "+s"(m1) : : "scc");
      } else if (KIND == 6) {  // VALU and SALU interleaved one to one (independent of each other)
        asm volatile("v_add_f32 %0, %0, %0\ns_add_u32 %4, %4, %5\nv_add_f32 %1, %1, %1\ns_and_b64 %6, %6, %7\n"
                     "v_add_f32 %2, %2, %2\ns_add_u32 %5, %5, %4\nv_add_f32 %3, %3, %3\ns_or_b64 %7, %7, %6"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(m0), "+s"(m1) : : "scc");
      } else if (KIND == 7) {  // v_readlane (VALU -> SGPR)
        asm volatile("v_readlane_b32 %4, %0, 3\nv_readlane_b32 %5, %1, 5\nv_readlane_b32 %4, %2, 7\nv_readlane_b32 %5, %3, 9\n"
                     "v_readlane_b32 %4, %1, 3\nv_readlane_b32 %5, %0, 5\nv_readlane_b32 %4, %3, 7\nv_readlane_b32 %5, %2, 9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1));
      } else if (KIND == 8) {  // packed f32
        asm volatile("v_pk_add_f32 %0, %0, %0\nv_pk_add_f32 %1, %1, %1\nv_pk_add_f32 %2, %2, %2\nv_pk_add_f32 %3, %3, %3\n"
                     "v_pk_add_f32 %0, %0, %0\nv_pk_add_f32 %1, %1, %1\nv_pk_add_f32 %2, %2, %2\nv_pk_add_f32 %3, %3, %3"
                     : "+v"(*(unsigned long long *)&a0), "+v"(*(unsigned long long *)&a2), "+v"(*(unsigned long long *)&a4), "+v"(*(unsigned long long *)&a6));
      } else if (KIND == 9) {  // v_max3 with abs modifiers (VOP3)
        asm volatile("v_max3_f32 %0, |%1|, |%2|, |%3|\nv_max3_f32 %4, |%5|, |%6|, |%7|\nv_max3_f32 %1, |%0|, |%2|, |%3|\nv_max3_f32 %5, |%4|, |%6|, |%7|\n"
                     "v_max3_f32 %2, |%1|, |%0|, |%3|\nv_max3_f32 %6, |%5|, |%4|, |%7|\nv_max3_f32 %3, |%1|, |%2|, |%0|\nv_max3_f32 %7, |%5|, |%6|, |%4|"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (KIND == 10) {  // v_cmp to VCC only
        asm volatile("v_cmp_le_f32 vcc, %0, %1\nv_cmp_le_f32 vcc, %2, %3\nv_cmp_le_f32 vcc, %4, %5\nv_cmp_le_f32 vcc, %6, %7\n"
                     "v_cmp_le_f32 vcc, %1, %0\nv_cmp_le_f32 vcc, %3, %2\nv_cmp_le_f32 vcc, %5, %4\nv_cmp_le_f32 vcc, %7, %6"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
      } else if (KIND == 11) {  // v_cndmask reading VCC only (VCC set once outside)
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %2, %2, %3, vcc\nv_cndmask_b32 %4, %4, %5, vcc\nv_cndmask_b32 %6, %6, %7, vcc\n"
                     "v_cndmask_b32 %1, %1, %0, vcc\nv_cndmask_b32 %3, %3, %2, vcc\nv_cndmask_b32 %5, %5, %4, vcc\nv_cndmask_b32 %7, %7, %6, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
      } else if (KIND == 12) {  // v_cmp to an SGPR pair only (VOP3), nothing reads it
        asm volatile("v_cmp_le_f32 %8, %0, %1\nv_cmp_le_f32 %9, %2, %3\nv_cmp_le_f32 %8, %4, %5\nv_cmp_le_f32 %9, %6, %7\n"
                     "v_cmp_le_f32 %8, %1, %0\nv_cmp_le_f32 %9, %3, %2\nv_cmp_le_f32 %8, %5, %4\nv_cmp_le_f32 %9, %7, %6"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(m0), "+s"(m1));
      } else if (KIND == 13) {  // ds_bpermute, eight in flight then one wait
        asm volatile("ds_bpermute_b32 %0, %1, %0\nds_bpermute_b32 %2, %1, %2\nds_bpermute_b32 %3, %1, %3\nds_bpermute_b32 %4, %1, %4\n"
                     "ds_bpermute_b32 %5, %1, %5\nds_bpermute_b32 %6, %1, %6\nds_bpermute_b32 %7, %1, %7\nds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (KIND == 14) {  // ds_bpermute, each waited for (the kernel's insert chain)
        asm volatile("ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n"
                     "ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n"
                     "ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n"
                     "ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)"
                     : "+v"(a0), "+v"(a1));
      } else if (KIND == 15) {  // v_addc with an SGPR carry-in (the kernel's counting)
        asm volatile("v_addc_co_u32_e64 %0, vcc, 0, %0, %8\nv_addc_co_u32_e64 %1, vcc, 0, %1, %9\nv_addc_co_u32_e64 %2, vcc, 0, %2, %8\nv_addc_co_u32_e64 %3, vcc, 0, %3, %9\n"
                     "v_addc_co_u32_e64 %4, vcc, 0, %4, %8\nv_addc_co_u32_e64 %5, vcc, 0, %5, %9\nv_addc_co_u32_e64 %6, vcc, 0, %6, %8\nv_addc_co_u32_e64 %7, vcc, 0, %7, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(m0), "+s"(m1) : : "vcc");
      } else if (KIND == 16) {  // two VALU then one independent SALU (the kernel's mix is about 5 : 3)
        asm volatile("v_add_f32 %0, %0, %0\nv_add_f32 %1, %1, %1\ns_add_u32 %4, %4, %5\nv_add_f32 %2, %2, %2\nv_add_f32 %3, %3, %3\ns_and_b64 %6, %6, %7\n"
                     "v_add_f32 %0, %0, %0\nv_add_f32 %1, %1, %1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(m0), "+s"(m1) : : "scc");
      } else if (KIND == 17) {  // taken scalar branches
        asm volatile("s_cmp_lg_u32 %0, 0x7fffffff\ns_cbranch_scc1 1f\ns_nop 0\n1:\ns_cmp_lg_u32 %0, 0x7fffffff\ns_cbranch_scc1 2f\ns_nop 0\n2:\n"
                     "s_cmp_lg_u32 %0, 0x7fffffff\ns_cbranch_scc1 3f\ns_nop 0\n3:\ns_cmp_lg_u32 %0, 0x7fffffff\ns_cbranch_scc1 4f\ns_nop 0\n4:"
                     : "+s"(s0) : : "scc");
      } else if (KIND == 18) {  // v_cndmask (VOP3) with a constant SGPR-pair mask
        asm volatile("v_cndmask_b32 %0, %0, %1, %8\nv_cndmask_b32 %2, %2, %3, %9\nv_cndmask_b32 %4, %4, %5, %8\nv_cndmask_b32 %6, %6, %7, %9\n"
                     "v_cndmask_b32 %1, %1, %0, %8\nv_cndmask_b32 %3, %3, %2, %9\nv_cndmask_b32 %5, %5, %4, %8\nv_cndmask_b32 %7, %7, %6, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(m0), "+s"(m1));
      } else if (KIND == 19) {  // v_cmp -> vcc, v_cndmask <- vcc pairs
        asm volatile("v_cmp_le_f32 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\nv_cmp_le_f32 vcc, %4, %5\nv_cndmask_b32 %6, %6, %7, vcc\n"
                     "v_cmp_le_f32 vcc, %1, %0\nv_cndmask_b32 %3, %3, %2, vcc\nv_cmp_le_f32 vcc, %5, %4\nv_cndmask_b32 %7, %7, %6, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
      } else if (KIND == 20) {  // s_cselect -> vcc, v_cndmask <- vcc (the kernel's uniform selects)
        asm volatile("s_cmp_lg_u32 %4, 7\ns_cselect_b64 vcc, -1, 0\nv_cndmask_b32 %0, %0, %1, vcc\ns_cmp_lg_u32 %4, 9\ns_cselect_b64 vcc, -1, 0\nv_cndmask_b32 %2, %2, %3, vcc\n"
                     "s_cmp_lg_u32 %4, 11\ns_cselect_b64 vcc, -1, 0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0) : : "vcc", "scc");
      } else if (KIND == 21) {  // v_sub_f32 x3 + v_max3 |abs| : the Chebyshev distance of the block test
        asm volatile("v_sub_f32 %0, %1, %2\nv_sub_f32 %3, %1, %4\nv_sub_f32 %5, %1, %6\nv_max3_f32 %7, |%0|, |%3|, |%5|\n"
                     "v_sub_f32 %0, %7, %2\nv_sub_f32 %3, %7, %4\nv_sub_f32 %5, %7, %6\nv_max3_f32 %1, |%0|, |%3|, |%5|"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (KIND == 22) {  // v_mul + v_fma (what a contracted distance would be) vs the packed form (kind 8)
        asm volatile("v_mul_f32 %0, %1, %1\nv_fma_f32 %0, %2, %2, %0\nv_fma_f32 %0, %3, %3, %0\nv_mul_f32 %4, %5, %5\nv_fma_f32 %4, %6, %6, %4\nv_fma_f32 %4, %7, %7, %4\n"
                     "v_mul_f32 %1, %0, %0\nv_mul_f32 %5, %4, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (KIND == 23) {  // v_and_or_b32 (VOP3 integer), v_add_u32 (VOP2)
        asm volatile("v_and_or_b32 %0, %1, 60, %2\nv_add_u32 %3, %4, %5\nv_and_or_b32 %6, %7, 60, %0\nv_add_u32 %1, %3, %6\n"
                     "v_and_or_b32 %2, %1, 60, %0\nv_add_u32 %4, %3, %2\nv_and_or_b32 %5, %4, 60, %6\nv_add_u32 %7, %5, %1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(m0 + m1) + (float)(s0 + s1);
}

template <int KIND>
double run(int waves_per_simd, int cus, float *out, int iters) {
  const int blocks = cus * 4 * waves_per_simd;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters / 8);
  CHECK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  // wave-instructions per SIMD = waves_per_simd * iters * kUnroll; cycles = ms * clock
  return (double)ms;
}

int main(int argc, char **argv) {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  float *out;
  CHECK(hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * sizeof(float)));
  const int iters = 20000;
  const char *names[] = {"v_add_f32 independent", "v_add_f32 dependent chain", "v_cmp->sgpr + v_cndmask", "v_mov_dpp", "v_cmp_lt_u64",
                         "SALU only", "VALU+SALU 1:1", "v_readlane", "v_pk_add_f32", "v_max3_f32 |abs|",
                         "v_cmp -> vcc", "v_cndmask <- vcc", "v_cmp -> sgpr (VOP3)", "ds_bpermute x8, one wait", "ds_bpermute, waited each", "v_addc carry-in sgpr",
                         "2 VALU : 1 SALU (per 8)", "s_cmp + taken s_cbranch (x4)",
                         "v_cndmask <- sgpr pair", "v_cmp->vcc + v_cndmask<-vcc", "s_cselect->vcc + v_cndmask", "3 v_sub + v_max3", "mul+fma+fma", "v_and_or / v_add_u32"};
  printf("%d CUs, %.2f GHz nominal; cycles per wave64 instruction PER SIMD (nominal clock)\n", cus, ghz);
  printf("%-28s %8s %8s %8s %8s\n", "kind", "1 wave", "2 waves", "4 waves", "8 waves");
  for (int kind = (argc > 1 ? atoi(argv[1]) : 0); kind < 24; kind++) {
    printf("%-28s", names[kind]); fflush(stdout);
    for (int w : {1, 2, 4, 8}) {
      double ms = 0;
      switch (kind) {
        case 0: ms = run<0>(w, cus, out, iters); break;
        case 1: ms = run<1>(w, cus, out, iters); break;
        case 2: ms = run<2>(w, cus, out, iters); break;
        case 3: ms = run<3>(w, cus, out, iters); break;
        case 4: ms = run<4>(w, cus, out, iters); break;
        case 5: ms = run<5>(w, cus, out, iters); break;
        case 6: ms = run<6>(w, cus, out, iters); break;
        case 7: ms = run<7>(w, cus, out, iters); break;
        case 8: ms = run<8>(w, cus, out, iters); break;
        case 9: ms = run<9>(w, cus, out, iters); break;
        case 10: ms = run<10>(w, cus, out, iters); break;
        case 11: ms = run<11>(w, cus, out, iters); break;
        case 12: ms = run<12>(w, cus, out, iters); break;
        case 13: ms = run<13>(w, cus, out, iters); break;
        case 14: ms = run<14>(w, cus, out, iters); break;
        case 15: ms = run<15>(w, cus, out, iters); break;
        case 16: ms = run<16>(w, cus, out, iters); break;
        case 17: ms = run<17>(w, cus, out, iters); break;
        case 18: ms = run<18>(w, cus, out, iters); break;
        case 19: ms = run<19>(w, cus, out, iters); break;
        case 20: ms = run<20>(w, cus, out, iters); break;
        case 21: ms = run<21>(w, cus, out, iters); break;
        case 22: ms = run<22>(w, cus, out, iters); break;
        default: ms = run<23>(w, cus, out, iters); break;
      }
      const double instr_per_simd = (double)w * iters * kUnroll;
      printf(" %8.2f", ms * 1e-3 * ghz * 1e9 / instr_per_simd); fflush(stdout);
    }
    printf("\n");
  }
  return 0;
}
