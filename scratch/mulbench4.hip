// issue-rate probes + radix-2^29 Montgomery multiply vs the FIPS 32-bit multiply of ff.h
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../icicle-snark_amd/csrc/ff.h"
#include "f29.h"
using namespace bn254;

__global__ void k_mad(uint64_t* out, const uint32_t* in, int iters)
{
  uint32_t a = in[threadIdx.x & 7], b = in[(threadIdx.x + 1) & 7];
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      c0 += (uint64_t)a * b; c1 += (uint64_t)a * b; c2 += (uint64_t)a * b; c3 += (uint64_t)a * b;
      c4 += (uint64_t)a * b; c5 += (uint64_t)a * b; c6 += (uint64_t)a * b; c7 += (uint64_t)a * b;
      asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
__global__ void k_add(uint64_t* out, const uint32_t* in, int iters)
{
  uint32_t a = in[threadIdx.x & 7];
  uint32_t c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      c0 += a; c1 += a; c2 += a; c3 += a; c4 += a; c5 += a; c6 += a; c7 += a;
      asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
__global__ void k_mullo(uint64_t* out, const uint32_t* in, int iters)
{
  uint32_t a = in[threadIdx.x & 7] | 1;
  uint32_t c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      c0 *= a; c1 *= a; c2 *= a; c3 *= a; c4 *= a; c5 *= a; c6 *= a; c7 *= a;
      asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
__global__ void k_fq(fe* out, const fe* in, int iters)
{
  fe acc = in[threadIdx.x & 7], m = in[(threadIdx.x + 3) & 7];
  for (int it = 0; it < iters; it++) acc = Fq::mul(acc, m);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_f29(f29::fe9* out, const f29::fe9* in, int iters)
{
  f29::fe9 acc = in[threadIdx.x & 7], m = in[(threadIdx.x + 3) & 7];
  for (int it = 0; it < iters; it++) acc = f29::mul(acc, m);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <class F> float timeit(F launch)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
  const int blocks = 4096, threads = 256, iters = 1000;
  uint64_t* o64; (void)hipMalloc((void**)&o64, (size_t)blocks * threads * 8);
  uint32_t hin[8] = {0x12345671, 0x2345678b, 3, 5, 7, 11, 13, 17};
  uint32_t* din; (void)hipMalloc((void**)&din, 32); (void)hipMemcpy(din, hin, 32, hipMemcpyHostToDevice);
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const double simds = pr.multiProcessorCount * 4.0, clk = pr.clockRate * 1e3;
  printf("CUs %d clock %.0f MHz\n", pr.multiProcessorCount, clk / 1e6);
  auto rate = [&](const char* name, float ms, double instr_per_thread) {
    double wave_instr = (double)blocks * threads / 64 * instr_per_thread;
    printf("%-18s %.3f ms  -> %.2f cycles per wave64 instruction per SIMD (nominal clock)\n", name, ms, ms * 1e-3 * clk * simds / wave_instr);
  };
  rate("v_mad_u64_u32", timeit([&] { hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(threads), 0, 0, o64, din, iters); }), iters * 64.0);
  rate("v_add_u32", timeit([&] { hipLaunchKernelGGL(k_add, dim3(blocks), dim3(threads), 0, 0, o64, din, iters); }), iters * 64.0);
  rate("v_mul_lo_u32", timeit([&] { hipLaunchKernelGGL(k_mullo, dim3(blocks), dim3(threads), 0, 0, o64, din, iters); }), iters * 64.0);

  // field multiplies
  fe h[8];
  for (int i = 0; i < 8; i++) { for (int j = 0; j < 8; j++) h[i].l[j] = 0x12345678u * (i + 1) + j * 0x9e3779b9u; h[i].l[7] &= 0x0fffffff; }
  fe* dfe; (void)hipMalloc((void**)&dfe, sizeof h); (void)hipMemcpy(dfe, h, sizeof h, hipMemcpyHostToDevice);
  fe* ofe; (void)hipMalloc((void**)&ofe, (size_t)blocks * threads * sizeof(fe));
  f29::fe9 h9[8];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 9; j++) h9[i].l[j] = (0x12345678u * (i + 1) + j * 0x9e3779b9u) & (j == 8 ? 0x3fffffu : f29::MASK);
  f29::fe9* d9; (void)hipMalloc((void**)&d9, sizeof h9); (void)hipMemcpy(d9, h9, sizeof h9, hipMemcpyHostToDevice);
  f29::fe9* o9; (void)hipMalloc((void**)&o9, (size_t)blocks * threads * sizeof(f29::fe9));
  const double n = (double)blocks * threads * iters;
  float m0 = timeit([&] { hipLaunchKernelGGL(k_fq, dim3(blocks), dim3(threads), 0, 0, ofe, dfe, iters); });
  float m1 = timeit([&] { hipLaunchKernelGGL(k_f29, dim3(blocks), dim3(threads), 0, 0, o9, d9, iters); });
  printf("Fq mul 8x32 FIPS asm: %.1f Gmul/s | 9x29 carry-free: %.1f Gmul/s\n", n / m0 / 1e6, n / m1 / 1e6);
  // correctness sample: one multiply, dumped for the host check
  hipLaunchKernelGGL(k_f29, dim3(1), dim3(64), 0, 0, o9, d9, 1);
  f29::fe9 res[8];
  (void)hipMemcpy(res, o9, sizeof res, hipMemcpyDeviceToHost);
  for (int t = 0; t < 8; t++) {
    printf("CHK");
    for (int j = 0; j < 9; j++) printf(" %u", h9[t & 7].l[j]);
    for (int j = 0; j < 9; j++) printf(" %u", h9[(t + 3) & 7].l[j]);
    for (int j = 0; j < 9; j++) printf(" %u", res[t].l[j]);
    printf("\n");
  }
  return 0;
}
