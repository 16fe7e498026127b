// fr29.h — BN254 SCALAR field in radix 2^29 (nine limbs in u32), Montgomery with R' = 2^261, lazy reduction: the arithmetic of
// the NTT passes (ntt.hip).  Same idea as ff29.h (the base field of the MSM bucket kernels): with 29-bit limbs a whole product
// column fits the 64-bit accumulator of v_mad_u64_u32, so a multiply is 81 + 81 multiply-adds and no carry instructions
// (the 8×32-bit form of ff.h: 136 + 128), and additions / subtractions are nine limb-wise instructions.
//
// (r = 2^28·k + 1, so r_0 = 2^28 + 1 and −r⁻¹ mod 2^29 = 2^28 − 1 in this radix; the shift-and-add forms of the quotient digit and
// of m·r_0 were measured against the plain multiplies and are not faster: a multiply-add is one instruction.)
//
// Conventions:  "N" = limbs l[0..7] < 2^29, l[8] holds the rest (< 2^32);  "< k" = value < k·r.  R'/r ≈ 169.28.
//   mul(a, w)     a: limbs < 2^31.5, l[8] < 2^32, any value < 2^264;  w: N, canonical (< 1)  →  N, value < a/169.28 + 1
//   add           limb-wise, no carry (limbs grow by one bit)
//   sub<K>(a, b)  a + K·r − b limb-wise; b must be N with value < (K − 1); K ≤ 1300
//   norm          carry propagation → N (value unchanged)
//   shrink        N, < 1354  →  N, < 0.244·a + 1   (subtracts ⌊a / 2^254⌋·r)
//   canon         N, < 1354  →  canonical N
// Data are kept in STANDARD form between the passes (as the 8×32-bit kernels do); the twiddles come in Montgomery-261 form
// (w·2^261 mod r, tables built once per domain), so a product x·w leaves in standard form again.
#pragma once
#include <stdint.h>

#include "ff29.h" // fe9, FF_HD

namespace bn254 {
namespace fr29 {

constexpr uint32_t MASK = (1u << 29) - 1;
constexpr uint32_t NINV = 0xfffffffu; // −r⁻¹ mod 2^29
#define FR29_R_LIMBS 0x10000001u, 0x1f0fac9fu, 0xe5c2450u, 0x7d090f3u, 0x1585d283u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu

struct Limbs9 {
  uint32_t v[9];
};
// K·r with limb i raised by j·2^29 and limb i + 1 lowered by j: every limb of a subtrahend whose limbs are ≤ j·(2^29 − 1) can be
// subtracted without going negative; the top limb is (K·r)_8 − j, so the subtrahend's value must be below (K − 1)·r.
// (constexpr, and cheap enough at run time for a wave-uniform K: the NTT passes compute their levels' constants in scalar registers)
FF_HD constexpr Limbs9 kr_borrow_proof(uint32_t k, uint32_t j = 1)
{
  constexpr uint32_t R[9] = {FR29_R_LIMBS};
  Limbs9 o{};
  uint64_t carry = 0;
  for (int i = 0; i < 9; i++) {
    const uint64_t t = (uint64_t)R[i] * k + carry;
    o.v[i] = i < 8 ? (uint32_t)(t & MASK) : (uint32_t)t;
    carry = t >> 29;
  }
  for (int i = 0; i < 8; i++) {
    o.v[i] += j << 29;
    o.v[i + 1] -= j;
  }
  return o;
}
constexpr Limbs9 kr_plain(uint32_t k)
{
  constexpr uint32_t R[9] = {FR29_R_LIMBS};
  Limbs9 o{};
  uint64_t carry = 0;
  for (int i = 0; i < 9; i++) {
    const uint64_t t = (uint64_t)R[i] * k + carry;
    o.v[i] = i < 8 ? (uint32_t)(t & MASK) : (uint32_t)t;
    carry = t >> 29;
  }
  return o;
}

// a·w·2^-261.  Column bound: Σ a_i·w_j (≤ 8 terms of < 2^31.5·2^29 and one with l[8] < 2^32) + Σ m_i·r_j (< 9·2^58) + carry < 2^64.
FF_HD fe9 mul(const fe9& a, const fe9& w)
{
  constexpr uint32_t R[9] = {FR29_R_LIMBS};
  uint64_t acc = 0;
  uint32_t m[9];
  fe9 o;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * w.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * R[k - i];
    m[k] = ((uint32_t)acc * NINV) & MASK; // −r⁻¹ mod 2^29 = 2^28 − 1
    acc += (uint64_t)m[k] * R[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * w.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * R[k - i];
    o.l[k - 9] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
  o.l[8] = (uint32_t)acc;
  return o;
}

FF_HD fe9 add(const fe9& a, const fe9& b)
{
  fe9 o;
#pragma unroll
  for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + b.l[i];
  return o;
}
// a + K·r − b;  b N, < K − 1
template <uint32_t K>
FF_HD fe9 sub(const fe9& a, const fe9& b)
{
  constexpr Limbs9 C = kr_borrow_proof(K);
  fe9 o;
#pragma unroll
  for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + (C.v[i] - b.l[i]);
  return o;
}
FF_HD fe9 norm(const fe9& a)
{
  fe9 o;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t t = a.l[i] + carry;
    o.l[i] = t & MASK;
    carry = t >> 29;
  }
  o.l[8] = a.l[8] + carry;
  return o;
}

// a − q·r for a small quotient q ≤ a / r (q < 2^11): one signed carry chain
FF_HD fe9 sub_qr(const fe9& a, uint32_t q)
{
  constexpr uint32_t R[9] = {FR29_R_LIMBS};
  fe9 o;
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int64_t t = (int64_t)a.l[i] - (int64_t)((uint64_t)q * R[i]) + carry;
    o.l[i] = (uint32_t)t & MASK;
    carry = t >> 29; // arithmetic shift: floor
  }
  o.l[8] = (uint32_t)((int64_t)a.l[8] - (int64_t)((uint64_t)q * R[8]) + carry);
  return o;
}
// N, < 1354 → N, < 0.244·a + 1:  q = ⌊a / 2^254⌋ ≤ a / r
FF_HD fe9 shrink(const fe9& a) { return sub_qr(a, a.l[8] >> 22); }

// conditional subtraction of K·r (a N)
template <uint32_t K>
FF_HD fe9 csub(const fe9& a)
{
  constexpr Limbs9 C = kr_plain(K);
  fe9 t;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t d = a.l[i] - C.v[i] - borrow;
    borrow = d >> 31;
    t.l[i] = i < 8 ? (d & MASK) : d;
  }
  fe9 o;
#pragma unroll
  for (int i = 0; i < 9; i++) o.l[i] = borrow ? a.l[i] : t.l[i];
  return o;
}
// N, < 4 → canonical
FF_HD fe9 canon4(const fe9& a) { return csub<1>(csub<2>(a)); }
// N, < 1354 → canonical:  q = ⌊l[8]·⌊2^53 / d⌋ / 2^53⌋ with d = (r >> 232) + 1 = 3171407 never exceeds a / r and falls short
// of ⌊a / r⌋ by at most 2 (a − q·r < 4·r; checked over random a < 1354·r: < 1.001·r)
FF_HD fe9 canon(const fe9& a)
{
  const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * 2840127191ull) >> 53);
  return canon4(sub_qr(a, q));
}

// 8×32-bit packed (value < 2^256) ↔ 9×29
FF_HD fe9 unpack(const fe& w)
{
  fe9 o;
  o.l[0] = w.l[0] & MASK;
  o.l[1] = ((w.l[0] >> 29) | (w.l[1] << 3)) & MASK;
  o.l[2] = ((w.l[1] >> 26) | (w.l[2] << 6)) & MASK;
  o.l[3] = ((w.l[2] >> 23) | (w.l[3] << 9)) & MASK;
  o.l[4] = ((w.l[3] >> 20) | (w.l[4] << 12)) & MASK;
  o.l[5] = ((w.l[4] >> 17) | (w.l[5] << 15)) & MASK;
  o.l[6] = ((w.l[5] >> 14) | (w.l[6] << 18)) & MASK;
  o.l[7] = ((w.l[6] >> 11) | (w.l[7] << 21)) & MASK;
  o.l[8] = w.l[7] >> 8;
  return o;
}
FF_HD fe pack(const fe9& a) // a N, value < 2^256
{
  fe w;
  w.l[0] = a.l[0] | (a.l[1] << 29);
  w.l[1] = (a.l[1] >> 3) | (a.l[2] << 26);
  w.l[2] = (a.l[2] >> 6) | (a.l[3] << 23);
  w.l[3] = (a.l[3] >> 9) | (a.l[4] << 20);
  w.l[4] = (a.l[4] >> 12) | (a.l[5] << 17);
  w.l[5] = (a.l[5] >> 15) | (a.l[6] << 14);
  w.l[6] = (a.l[6] >> 18) | (a.l[7] << 11);
  w.l[7] = (a.l[7] >> 21) | (a.l[8] << 8);
  return w;
}

#define FR29_CONST(name, ...)                                                                                          \
  FF_HD fe9 name()                                                                                                     \
  {                                                                                                                    \
    constexpr uint32_t c[9] = {__VA_ARGS__};                                                                           \
    fe9 o;                                                                                                             \
    for (int i = 0; i < 9; i++) o.l[i] = c[i];                                                                         \
    return o;                                                                                                          \
  }
// 2^266 mod r: x·2^256 (the Montgomery form of ff.h) → x·2^261:  mul(unpack(x256), c256_to_261())
FR29_CONST(c256_to_261, 0xfffead7u, 0x1d5444f4u, 0x4438aa5u, 0x3b4d096u, 0x134c84dau, 0xe92d304u, 0x14cb95b3u, 0x41b9d3du, 0x58003u)
// 2^522 mod r: standard → Montgomery-261 (and "standard-form product of two standard-form values": mul(mul(x, y), r2()))
FR29_CONST(r2, 0x5b69bd4u, 0x6170a5au, 0x20cddceu, 0x1db6310bu, 0xe54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x7d09161u, 0xa054au)

} // namespace fr29
} // namespace bn254
