// variant: two interleaved 96-bit accumulators per column (a·b chain and m·p chain) for ILP
#pragma once
#include "../icicle-snark_amd/csrc/ff.h"
namespace bn254 {
template <class P>
__device__ __forceinline__ fe mont_mul_fips2(const fe& a, const fe& b)
{
  typedef Fp<P> F;
  uint64_t accA = 0, accB = 0;
  uint32_t topA = 0, topB = 0;
  uint32_t m[8];
  fe r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      F::mac96(accA, topA, a.l[i], b.l[k - i]);
      if (i < k) F::mac96s(accB, topB, m[i], P::MOD[k - i]);
    }
    // merge: A += B
    {
      uint64_t s = accA + accB;
      uint32_t c = s < accA;
      accA = s;
      topA += topB + c;
    }
    m[k] = (uint32_t)accA * P::NINV;
    F::mac96s(accA, topA, m[k], P::MOD[0]);
    accA = (accA >> 32) | ((uint64_t)topA << 32);
    topA = 0; accB = 0; topB = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; k++) {
#pragma unroll
    for (int i = k - 7; i < 8; i++) {
      F::mac96(accA, topA, a.l[i], b.l[k - i]);
      F::mac96s(accB, topB, m[i], P::MOD[k - i]);
    }
    {
      uint64_t s = accA + accB;
      uint32_t c = s < accA;
      accA = s;
      topA += topB + c;
    }
    r.l[k - 8] = (uint32_t)accA;
    accA = (accA >> 32) | ((uint64_t)topA << 32);
    topA = 0; accB = 0; topB = 0;
  }
  return F::reduce_once(r);
}
} // namespace bn254
