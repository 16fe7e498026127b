// f29.h — experiment: BN254 Fq in radix 2^29 (9 limbs in u32), Montgomery with R = 2^261, lazy reduction.
// Column sums of 9+9 products of < 2^30-bit limbs fit a 64-bit accumulator, so v_mad_u64_u32 needs no carry capture.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace f29 {
constexpr uint32_t MASK = (1u << 29) - 1;
struct fe9 { uint32_t l[9]; };

// p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
// limbs of p in radix 2^29 and -p^-1 mod 2^29 (filled by gen below, checked by the host test)
__device__ __constant__ const uint32_t P29[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
constexpr uint32_t NINV29 = 0x4866389u;

#define F29_P(i) (P29c[i])

template <bool SQR = false>
__device__ __forceinline__ fe9 mul(const fe9& a, const fe9& b)
{
  constexpr uint32_t P29c[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
  uint64_t acc = 0;
  uint32_t m[9];
  fe9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P29c[k - i];
    m[k] = ((uint32_t)acc * NINV29) & MASK;
    acc += (uint64_t)m[k] * P29c[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * P29c[k - i];
    r.l[k - 9] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
} // namespace f29
