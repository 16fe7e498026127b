#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "mul2.h"
using namespace bn254;
template <int V, int ILP>
__global__ void k_fqmul(fe* out, const fe* in, int iters)
{
  fe acc[ILP];
  fe m = in[threadIdx.x & 7];
  for (int i = 0; i < ILP; i++) acc[i] = in[(threadIdx.x + i) & 7];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = V == 0 ? Fq::mul(acc[i], m) : mont_mul_fips<FqP>(acc[i], m);
  }
  fe s = acc[0];
  for (int i = 1; i < ILP; i++) s = Fq::add(s, acc[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
  const int blocks = 2048, threads = 256, iters = 2000;
  fe* buf0; fe* buf1; (void)hipMalloc((void**)&buf0, blocks * threads * sizeof(fe)); (void)hipMalloc((void**)&buf1, blocks * threads * sizeof(fe));
  fe h[8];
  for (int i = 0; i < 8; i++) { for (int j = 0; j < 8; j++) h[i].l[j] = 0x12345678u * (i + 1) + j * 0x9e3779b9u; h[i].l[7] &= 0x0fffffff; }
  fe* din; (void)hipMalloc((void**)&din, sizeof h); (void)hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    float ms0 = timeit([&] { hipLaunchKernelGGL((k_fqmul<0, 1>), dim3(blocks), dim3(threads), 0, 0, buf0, din, iters); });
    float ms1 = timeit([&] { hipLaunchKernelGGL((k_fqmul<1, 1>), dim3(blocks), dim3(threads), 0, 0, buf1, din, iters); });
    float ms2 = timeit([&] { hipLaunchKernelGGL((k_fqmul<1, 2>), dim3(blocks), dim3(threads), 0, 0, buf1, din, iters); });
    const double n = (double)blocks * threads * iters;
    printf("CIOS(C++) %.3f ms %.1f Gmul/s | FIPS(asm) %.3f ms %.1f Gmul/s | FIPS ILP2 %.1f Gmul/s\n", ms0, n / ms0 / 1e6, ms1, n / ms1 / 1e6, 2 * n / ms2 / 1e6);
  }
  hipLaunchKernelGGL((k_fqmul<0, 1>), dim3(blocks), dim3(threads), 0, 0, buf0, din, 50);
  hipLaunchKernelGGL((k_fqmul<1, 1>), dim3(blocks), dim3(threads), 0, 0, buf1, din, 50);
  hipDeviceSynchronize();
  fe a[512], b[512];
  (void)hipMemcpy(a, buf0, sizeof a, hipMemcpyDeviceToHost); (void)hipMemcpy(b, buf1, sizeof b, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 512; i++) for (int j = 0; j < 8; j++) if (a[i].l[j] != b[i].l[j]) bad++;
  printf("mismatching limbs: %d\n", bad);
  return 0;
}
