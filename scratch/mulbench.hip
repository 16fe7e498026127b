// micro-benchmarks: raw instruction rates and Fq::mul throughput on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../icicle-snark_amd/csrc/ff.h"
using namespace bn254;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ILP>
__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b, int iters)
{
  uint64_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
  uint32_t x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (uint64_t)x * (uint32_t)(y + i) + acc[i];
    x = (uint32_t)acc[0];
  }
  uint64_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mullo(uint32_t* out, uint32_t a, uint32_t b, int iters)
{
  uint32_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = acc[i] * (b + i);
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mulhi(uint32_t* out, uint32_t a, uint32_t b, int iters)
{
  uint32_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __umulhi(acc[i], b + i) + 12345u;
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_fma64(double* out, double a, double b, int iters)
{
  double acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_add32(uint32_t* out, uint32_t a, int iters)
{
  uint32_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (acc[i] + a) ^ (acc[i] >> 3);
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mad24(uint32_t* out, uint32_t a, uint32_t b, int iters)
{
  uint32_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __umul24(acc[i] & 0xffffff, b + i) + a;
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_fqmul(fe* out, const fe* in, int iters)
{
  fe acc[ILP];
  fe m = in[threadIdx.x & 7];
  for (int i = 0; i < ILP; i++) acc[i] = in[(threadIdx.x + i) & 7];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = Fq::mul(acc[i], m);
  }
  fe s = acc[0];
  for (int i = 1; i < ILP; i++) s = Fq::add(s, acc[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
float timeit(F launch)
{
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main()
{
  void* buf;
  CHECK(hipMalloc(&buf, 256 * 2048 * 64 * 8));
  fe h[8];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) h[i].l[j] = 0x12345678u * (i + 1) + j * 0x9e3779b9u;
  for (int i = 0; i < 8; i++) h[i].l[7] &= 0x0fffffff;
  fe* din;
  CHECK(hipMalloc((void**)&din, sizeof h));
  CHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
  const int blocks = 256 * 8, threads = 256, iters = 2000;
  const double lanes = (double)blocks * threads;
#define RUN(name, ILP, kern, ...) { float ms = timeit([&] { hipLaunchKernelGGL((kern<ILP>), dim3(blocks), dim3(threads), 0, 0, __VA_ARGS__); }); \
    printf("%-12s ILP=%d: %8.3f ms  -> %8.2f Tops/s\n", name, ILP, ms, lanes * iters * ILP / (ms * 1e-3) / 1e12); }
  RUN("mad_u64_u32", 1, k_mad64, (uint64_t*)buf, 3u, 5u, iters)
  RUN("mad_u64_u32", 4, k_mad64, (uint64_t*)buf, 3u, 5u, iters)
  RUN("mad_u64_u32", 8, k_mad64, (uint64_t*)buf, 3u, 5u, iters)
  RUN("mul_lo_u32", 8, k_mullo, (uint32_t*)buf, 3u, 5u, iters)
  RUN("mul_hi_u32", 8, k_mulhi, (uint32_t*)buf, 3u, 5u, iters)
  RUN("fma_f64", 8, k_fma64, (double*)buf, 1.0000001, 0.5, iters)
  RUN("add+xor u32", 8, k_add32, (uint32_t*)buf, 3u, iters)
  RUN("mul24", 8, k_mad24, (uint32_t*)buf, 3u, 5u, iters)
  {
    const int it2 = 200;
    for (int b2 : {256 * 2, 256 * 4, 256 * 8}) {
      float ms = timeit([&] { hipLaunchKernelGGL((k_fqmul<1>), dim3(b2), dim3(256), 0, 0, (fe*)buf, din, it2); });
      printf("Fq::mul ILP=1 blocks=%d: %8.3f ms -> %7.2f Gmul/s\n", b2, ms, (double)b2 * 256 * it2 / (ms * 1e-3) / 1e9);
      ms = timeit([&] { hipLaunchKernelGGL((k_fqmul<2>), dim3(b2), dim3(256), 0, 0, (fe*)buf, din, it2); });
      printf("Fq::mul ILP=2 blocks=%d: %8.3f ms -> %7.2f Gmul/s\n", b2, ms, (double)b2 * 256 * it2 * 2 / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
