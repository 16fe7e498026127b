#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "mul3.h"
using namespace bn254;
template <int V>
__global__ void k_fqmul(fe* out, const fe* in, int iters)
{
  extern __shared__ char lds[];
  fe acc = in[threadIdx.x & 7];
  fe m = in[(threadIdx.x + 3) & 7];
  for (int it = 0; it < iters; it++) {
#if defined(__HIP_DEVICE_COMPILE__)
    acc = V == 0 ? Fq::mul(acc, m) : mont_mul_fips2<FqP>(acc, m);
#endif
  }
  if (iters < 0) lds[threadIdx.x] = 1;
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
  const int blocks = 2048, threads = 256, iters = 2000;
  fe* buf0; fe* buf1; (void)hipMalloc((void**)&buf0, blocks * threads * sizeof(fe)); (void)hipMalloc((void**)&buf1, blocks * threads * sizeof(fe));
  fe h[8];
  for (int i = 0; i < 8; i++) { for (int j = 0; j < 8; j++) h[i].l[j] = 0x12345678u * (i + 1) + j * 0x9e3779b9u; h[i].l[7] &= 0x0fffffff; }
  fe* din; (void)hipMalloc((void**)&din, sizeof h); (void)hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)k_fqmul<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)k_fqmul<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const double n = (double)blocks * threads * iters;
  for (int wpc : {32, 16, 8, 4}) {               // waves per CU via LDS footprint (256-thread blocks = 4 waves)
    size_t lds = wpc >= 32 ? 0 : (size_t)(160 * 1024) / (wpc / 4) - 1024;
    float ms0 = timeit([&] { hipLaunchKernelGGL((k_fqmul<0>), dim3(blocks), dim3(threads), lds, 0, buf0, din, iters); });
    float ms1 = timeit([&] { hipLaunchKernelGGL((k_fqmul<1>), dim3(blocks), dim3(threads), lds, 0, buf1, din, iters); });
    printf("waves/CU %2d (%d per SIMD): FIPS 1-chain %.1f Gmul/s | 2-chain %.1f Gmul/s\n", wpc, wpc / 4, n / ms0 / 1e6, n / ms1 / 1e6);
  }
  hipLaunchKernelGGL((k_fqmul<0>), dim3(blocks), dim3(threads), 0, 0, buf0, din, 50);
  hipLaunchKernelGGL((k_fqmul<1>), dim3(blocks), dim3(threads), 0, 0, buf1, din, 50);
  hipDeviceSynchronize();
  fe a[512], b[512];
  (void)hipMemcpy(a, buf0, sizeof a, hipMemcpyDeviceToHost); (void)hipMemcpy(b, buf1, sizeof b, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 512; i++) for (int j = 0; j < 8; j++) if (a[i].l[j] != b[i].l[j]) bad++;
  printf("mismatching limbs: %d\n", bad);
  return 0;
}
