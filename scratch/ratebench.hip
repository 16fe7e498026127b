// issue rates of the instructions around a radix-2^29 product column on gfx950: v_mad_u64_u32, v_lshrrev_b64, v_lshl_add_u64, v_alignbit_b32,
// 32-bit shift / and / add, 64-bit add.  Eight independent chains per lane, 4 waves per SIMD; prints cycles per wave-instruction.
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/ratebench scratch/ratebench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHAINS8(OP)                                                                                            \
  for (int it = 0; it < iters; it++) {                                                                         \
    _Pragma("unroll") for (int u = 0; u < 8; u++)                                                              \
    {                                                                                                          \
      OP(c0) OP(c1) OP(c2) OP(c3) OP(c4) OP(c5) OP(c6) OP(c7)                                                  \
    }                                                                                                          \
  }
#define KERNEL(name, T, OP)                                                                                    \
  __global__ void name(uint64_t* out, const uint32_t* in, int iters)                                           \
  {                                                                                                            \
    const uint32_t a = in[threadIdx.x & 7] | 1u, b = in[(threadIdx.x + 1) & 7] | 3u;                           \
    T c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = 5, c5 = 6, c6 = 7, c7 = 8;                                  \
    CHAINS8(OP)                                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);            \
  }
#define OP_MAD(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
#define OP_SHR64(c) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(c));
#define OP_LSHLADD64(c) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(c) : "v"((uint64_t)a));
#define OP_ALIGN(c) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(c) : "v"(a));
#define OP_SHR32(c) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(c));
#define OP_AND(c) asm volatile("v_and_b32 %0, %1, %0" : "+v"(c) : "v"(a));
#define OP_ADD(c) asm volatile("v_add_u32 %0, %1, %0" : "+v"(c) : "v"(a));
#define OP_ADD3(c) asm volatile("v_add3_u32 %0, %1, %0, %2" : "+v"(c) : "v"(a), "v"(b));
#define OP_MULLO(c) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(c) : "v"(a));
#define OP_ADD64(c) c += (uint64_t)a; asm volatile("" : "+v"(c));
KERNEL(k_mad, uint64_t, OP_MAD)
KERNEL(k_shr64, uint64_t, OP_SHR64)
KERNEL(k_lshladd64, uint64_t, OP_LSHLADD64)
KERNEL(k_align, uint32_t, OP_ALIGN)
KERNEL(k_shr32, uint32_t, OP_SHR32)
KERNEL(k_and, uint32_t, OP_AND)
KERNEL(k_add, uint32_t, OP_ADD)
KERNEL(k_add3, uint32_t, OP_ADD3)
KERNEL(k_mullo, uint32_t, OP_MULLO)
KERNEL(k_add64, uint64_t, OP_ADD64)
template <class K>
static void run(const char* name, K k, int per_iter)
{
  uint64_t* out; uint32_t* in;
  hipMalloc(&out, 1024 * 256 * 8 * 8); hipMalloc(&in, 64);
  uint32_t h[8] = {3, 5, 7, 11, 13, 17, 19, 23}; hipMemcpy(in, h, 32, hipMemcpyHostToDevice);
  const int iters = 2000, blocks = 256 * 4; // 4 workgroups of 256 per CU = 4 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, 10);
  hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const double wave_instr_per_simd = (double)iters * 64 * per_iter * 4; // 8 unroll × 8 chains = 64 per iter per wave, 4 waves per SIMD
  printf("%-14s %.3f ms  -> %.2f cycles per wave-instruction at %d MHz (nominal)\n", name, ms, ms * 1e-3 * khz * 1e3 / wave_instr_per_simd, khz / 1000);
}
int main()
{
  run("v_mad_u64_u32", k_mad, 1); run("v_lshrrev_b64", k_shr64, 1); run("v_lshl_add_u64", k_lshladd64, 1); run("v_alignbit_b32", k_align, 1);
  run("v_lshrrev_b32", k_shr32, 1); run("v_and_b32", k_and, 1); run("v_add_u32", k_add, 1); run("v_add3_u32", k_add3, 1); run("v_mul_lo_u32", k_mullo, 1);
  run("add64 (2 instr)", k_add64, 1);
  return 0;
}
