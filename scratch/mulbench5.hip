// radix-2^29 Montgomery multiply: compiler-scheduled columns (ff29.h today: one accumulator per column + a 64-bit add of the
// carry per column) against a SERIAL column chain with explicit v_mad_u64_u32 (the shifted carry is the addend of the next
// column's first multiply-add: no v_lshl_add_u64 per column).  Prints G mul/s of both and checks they agree.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "f29.h"
using namespace f29;

#define MAD(acc, x, y) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc")
#define MADS(acc, x, y) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(y) : "vcc")

template <int MODE>
__device__ __forceinline__ fe9 mul_serial(const fe9& a, const fe9& b)
{
  constexpr uint32_t P29c[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
  uint64_t acc = 0;
  uint32_t m[9];
  fe9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) MAD(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) {
      if (MODE == 0) MAD(acc, m[i], P29c[k - i]);
      else MADS(acc, m[i], P29c[k - i]);
    }
    m[k] = ((uint32_t)acc * NINV29) & MASK;
    if (MODE == 0) MAD(acc, m[k], P29c[0]);
    else MADS(acc, m[k], P29c[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) MAD(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 8; i < 9; i++) {
      if (MODE == 0) MAD(acc, m[i], P29c[k - i]);
      else MADS(acc, m[i], P29c[k - i]);
    }
    r.l[k - 9] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
// two interleaved chains: even columns' a·b products and m·p products in separate accumulators joined once per column
template <int V>
__global__ void k_mul(fe9* out, const fe9* in, int iters)
{
  fe9 acc = in[threadIdx.x & 7], m = in[(threadIdx.x + 3) & 7];
  for (int it = 0; it < iters; it++) {
    if (V == 0) acc = f29::mul(acc, m);
    else if (V == 1) acc = mul_serial<0>(acc, m);
    else acc = mul_serial<1>(acc, m);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// two independent multiplications per iteration (what a mixed addition offers: U2 and S2, PPP and Q, ZZ and ZZZ are independent)
template <int V>
__global__ void k_mul2(fe9* out, const fe9* in, int iters)
{
  fe9 a0 = in[threadIdx.x & 7], a1 = in[(threadIdx.x + 1) & 7], m = in[(threadIdx.x + 3) & 7];
  for (int it = 0; it < iters; it++) {
    if (V == 0) { a0 = f29::mul(a0, m); a1 = f29::mul(a1, m); }
    else { a0 = mul_serial<0>(a0, m); a1 = mul_serial<0>(a1, m); }
  }
  for (int j = 0; j < 9; j++) a0.l[j] ^= a1.l[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
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
  fe9 h9[8];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 9; j++) h9[i].l[j] = (0x12345678u * (i + 1) + j * 0x9e3779b9u) & (j == 8 ? 0x3fffffu : MASK);
  fe9* d9; (void)hipMalloc((void**)&d9, sizeof h9); (void)hipMemcpy(d9, h9, sizeof h9, hipMemcpyHostToDevice);
  fe9 *o0, *o1, *o2; (void)hipMalloc((void**)&o0, (size_t)blocks * threads * sizeof(fe9)); (void)hipMalloc((void**)&o1, (size_t)blocks * threads * sizeof(fe9)); (void)hipMalloc((void**)&o2, (size_t)blocks * threads * sizeof(fe9));
  const double n = (double)blocks * threads * iters;
  float m0 = timeit([&] { hipLaunchKernelGGL(k_mul<0>, dim3(blocks), dim3(threads), 0, 0, o0, d9, iters); });
  float m1 = timeit([&] { hipLaunchKernelGGL(k_mul<1>, dim3(blocks), dim3(threads), 0, 0, o1, d9, iters); });
  float m2 = timeit([&] { hipLaunchKernelGGL(k_mul<2>, dim3(blocks), dim3(threads), 0, 0, o2, d9, iters); });
  printf("one chain per thread : compiler columns %.1f Gmul/s | serial asm mads %.1f Gmul/s | serial, p in SGPR/literal %.1f Gmul/s\n", n / m0 / 1e6, n / m1 / 1e6, n / m2 / 1e6);
  float q0 = timeit([&] { hipLaunchKernelGGL(k_mul2<0>, dim3(blocks), dim3(threads), 0, 0, o0, d9, iters); });
  float q1 = timeit([&] { hipLaunchKernelGGL(k_mul2<1>, dim3(blocks), dim3(threads), 0, 0, o1, d9, iters); });
  printf("two chains per thread: compiler columns %.1f Gmul/s | serial asm mads %.1f Gmul/s\n", 2 * n / q0 / 1e6, 2 * n / q1 / 1e6);
  // agreement
  hipLaunchKernelGGL(k_mul<0>, dim3(1), dim3(64), 0, 0, o0, d9, 3);
  hipLaunchKernelGGL(k_mul<1>, dim3(1), dim3(64), 0, 0, o1, d9, 3);
  hipLaunchKernelGGL(k_mul<2>, dim3(1), dim3(64), 0, 0, o2, d9, 3);
  fe9 r0[64], r1[64], r2[64];
  (void)hipMemcpy(r0, o0, sizeof r0, hipMemcpyDeviceToHost); (void)hipMemcpy(r1, o1, sizeof r1, hipMemcpyDeviceToHost); (void)hipMemcpy(r2, o2, sizeof r2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; t++) for (int j = 0; j < 9; j++) if (r0[t].l[j] != r1[t].l[j] || r0[t].l[j] != r2[t].l[j]) bad++;
  printf("agreement: %s\n", bad ? "MISMATCH" : "ok");
  return bad != 0;
}
