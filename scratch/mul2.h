// candidate: product-scanning (FIPS) Montgomery multiply with v_mad_u64_u32 + v_addc carry capture
#pragma once
#include "../icicle-snark_amd/csrc/ff.h"
namespace bn254 {
// (top:acc) += a*b   — 96-bit accumulator, 2 instructions, no register moves
__host__ __device__ __forceinline__ void mac96(uint64_t& acc, uint32_t& top, uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(top) : "v"(a), "v"(b) : "vcc");
#else
  unsigned __int128 t = ((unsigned __int128)top << 64) + acc + (uint64_t)a * b;
  acc = (uint64_t)t;
  top = (uint32_t)(t >> 64);
#endif
}
__host__ __device__ __forceinline__ void mac96s(uint64_t& acc, uint32_t& top, uint32_t a, uint32_t b_const)
{
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(top) : "v"(a), "s"(b_const) : "vcc");
#else
  mac96(acc, top, a, b_const);
#endif
}
template <class P>
__host__ __device__ __forceinline__ fe mont_mul_fips(const fe& a, const fe& b)
{
  uint64_t acc = 0;
  uint32_t top = 0;
  uint32_t m[8], r[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mac96(acc, top, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) mac96s(acc, top, m[i], P::MOD[k - i]);
    m[k] = (uint32_t)acc * P::NINV;
    mac96s(acc, top, m[k], P::MOD[0]);
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; k++) {
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac96(acc, top, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac96s(acc, top, m[i], P::MOD[k - i]);
    r[k - 8] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
  fe out;
#pragma unroll
  for (int i = 0; i < 8; i++) out.l[i] = r[i];
  return Fp<P>::reduce_once(out);
}
} // namespace bn254
