// ff29.h — BN254 base field in radix 2^29 (nine 29-bit limbs in u32), Montgomery with R' = 2^261, LAZY reduction.
//
// Why: on gfx950 v_mad_u64_u32 issues at the same rate as any other VALU instruction (measured: 4 cycles per
// wave64, scratch/mulbench4.hip), so a multiply's cost is its instruction count.  With 32-bit limbs every
// multiply-add needs a second instruction to catch the carry out of the 64-bit accumulator (ff.h: 136 mad +
// 128 addc per product).  With 29-bit limbs a whole column — nine a·b products plus nine m·p products of
// < 2^30-bit operands — fits a 64-bit accumulator: 165 mads, no carry instructions, 175 G mul/s against 127 G
// (MI355X, same harness).  R'/p ≈ 147 leaves room to skip almost every modular reduction: values live in
// [0, k·p) for small k and limbs may exceed 29 bits between normalisations.  The price is that every formula
// states its bounds; `F29_CHECK` (host builds, tests/test_f29.py) turns them into assertions.
//
// Conventions used in comments:  "N" = limbs l[0..7] < 2^29 (l[8] takes the rest);  "< k·p" = value bound.
//   mul / sqr / mul2   inputs: per column  Σ limb products < 2^64 − 2^62 (e.g. both < 2^30, or 2^31 × 2^29),
//                      value product(s) < 147·p²;  output: N, value < (1 + Σ a·b / (147 p²))·p  ≤ 2p
//   add                limb-wise, no carry
//   sub<K,J>(a, b)     a + K·p − b  limb-wise, valid when b's limbs < J·2^29 and b < (K − J·2^-20)·p … see below
//   norm               carry propagation → N (value unchanged)
// Only mul/sqr/mul2 reduce values; only norm reduces limbs.
#pragma once
#include <stdint.h>

#include "ff.h"

#if defined(F29_CHECK)
// host-only checked build: every bound stated in the comments becomes a recorded failure (first one wins)
namespace bn254 { namespace f29 {
inline const char* g_check_failure = nullptr;
inline void check_fail(const char* what) { if (!g_check_failure) g_check_failure = what; }
} }
#define F29_ASSERT(cond, what) do { if (!(cond)) ::bn254::f29::check_fail(what); } while (0)
#else
#define F29_ASSERT(cond, what) (void)0
#endif

namespace bn254 {

struct fe9 {
  uint32_t l[9];
};

namespace f29 {

constexpr uint32_t MASK = (1u << 29) - 1;
// p in radix 2^29, −p⁻¹ mod 2^29, p⁻¹ mod 2^29
#define F29_P_LIMBS 0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu
constexpr uint32_t NINV = 0x4866389u;
constexpr uint32_t PINV = (0u - NINV) & MASK;

struct Limbs9 {
  uint32_t v[9];
};
// k·p in radix 2^29 with limb i raised by j·2^29 and limb i+1 lowered by j ("borrow-proof": every limb of a
// subtrahend whose limbs are ≤ j·(2^29 − 1) can be subtracted limb-wise without going negative).  The top limb is
// (k·p)_8 − j, so the subtrahend must satisfy  b_8 ≤ (k·p)_8 − j,  guaranteed when b < (k − 1)·p and j ≤ 2^20.
constexpr Limbs9 kp_borrow_proof(uint32_t k, uint32_t j)
{
  constexpr uint32_t P[9] = {F29_P_LIMBS};
  Limbs9 r{};
  uint64_t carry = 0;
  for (int i = 0; i < 9; i++) {
    uint64_t t = (uint64_t)P[i] * k + carry;
    r.v[i] = i < 8 ? (uint32_t)(t & MASK) : (uint32_t)t;
    carry = t >> 29;
  }
  for (int i = 0; i < 8; i++) {
    r.v[i] += j << 29;
    r.v[i + 1] -= j;
  }
  return r;
}
constexpr Limbs9 kp_plain(uint32_t k)
{
  constexpr uint32_t P[9] = {F29_P_LIMBS};
  Limbs9 r{};
  uint64_t carry = 0;
  for (int i = 0; i < 9; i++) {
    uint64_t t = (uint64_t)P[i] * k + carry;
    r.v[i] = i < 8 ? (uint32_t)(t & MASK) : (uint32_t)t;
    carry = t >> 29;
  }
  return r;
}

#if defined(F29_CHECK)
typedef unsigned __int128 u128;
inline void check_col(u128 acc) { F29_ASSERT(acc < ((u128)1 << 64), "f29: column accumulator overflow"); }
// value of a (possibly unnormalised) element as a pair (hi, lo) is awkward; use long double for bound checks only
inline long double approx_over_p(const fe9& a)
{
  long double v = 0, s = 1;
  for (int i = 0; i < 9; i++) {
    v += (long double)a.l[i] * s;
    s *= 536870912.0L;
  }
  return v / 2.18882428718392752222464057452572750886963111572978236626890378946452262e76L;
}
#define F29_ACC_T u128
#define F29_COL_CHECK(acc) check_col(acc)
#else
#define F29_ACC_T uint64_t
#define F29_COL_CHECK(acc) (void)0
#endif

// Σ_{q<NP} a_q·b_q · R'⁻¹ with ONE Montgomery reduction: product scanning over 17 columns; the compiler is free to
// reassociate the column sums (it keeps one 64-bit accumulator per column and interleaves the mads).
// Column bound: Σ_q 9·max(a_q limb)·max(b_q limb) + 9·2^58 + 2^35 < 2^64.
template <int NP>
FF_HD fe9 mul_core(const fe9& a0, const fe9& b0, const fe9& a1, const fe9& b1, const fe9& a2, const fe9& b2, const fe9& a3, const fe9& b3)
{
  constexpr uint32_t P[9] = {F29_P_LIMBS};
  F29_ACC_T acc = 0;
  uint32_t m[9];
  fe9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      acc += (uint64_t)a0.l[i] * b0.l[k - i];
      if (NP > 1) acc += (uint64_t)a1.l[i] * b1.l[k - i];
      if (NP > 2) acc += (uint64_t)a2.l[i] * b2.l[k - i];
      if (NP > 3) acc += (uint64_t)a3.l[i] * b3.l[k - i];
    }
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P[k - i];
    m[k] = ((uint32_t)acc * NINV) & MASK;
    acc += (uint64_t)m[k] * P[0];
    F29_COL_CHECK(acc);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) {
      acc += (uint64_t)a0.l[i] * b0.l[k - i];
      if (NP > 1) acc += (uint64_t)a1.l[i] * b1.l[k - i];
      if (NP > 2) acc += (uint64_t)a2.l[i] * b2.l[k - i];
      if (NP > 3) acc += (uint64_t)a3.l[i] * b3.l[k - i];
    }
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * P[k - i];
    F29_COL_CHECK(acc);
    r.l[k - 9] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
#if defined(F29_CHECK)
  F29_ASSERT(approx_over_p(r) < 2.0L, "f29: multiply output not below 2p (input value bound violated)");
#endif
  return r;
}
FF_HD fe9 mul(const fe9& a, const fe9& b) { return mul_core<1>(a, b, a, b, a, b, a, b); }
FF_HD fe9 mul2(const fe9& a, const fe9& b, const fe9& c, const fe9& d) { return mul_core<2>(a, b, c, d, a, b, a, b); } // a·b + c·d
FF_HD fe9 mul4(const fe9& a, const fe9& b, const fe9& c, const fe9& d, const fe9& e, const fe9& f, const fe9& g, const fe9& h)
{
  return mul_core<4>(a, b, c, d, e, f, g, h); // a·b + c·d + e·f + g·h
}
// a²: the off-diagonal products are taken once against the doubled operand (limbs of a must be < 2^29·√2 … N in practice)
FF_HD fe9 sqr(const fe9& a)
{
  constexpr uint32_t P[9] = {F29_P_LIMBS};
  F29_ACC_T acc = 0;
  uint32_t m[9], a2[9];
  fe9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a.l[i] * a2[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P[k - i];
    m[k] = ((uint32_t)acc * NINV) & MASK;
    acc += (uint64_t)m[k] * P[0];
    F29_COL_CHECK(acc);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; 2 * i < k; i++) acc += (uint64_t)a.l[i] * a2[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * P[k - i];
    F29_COL_CHECK(acc);
    r.l[k - 9] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
#if defined(F29_CHECK)
  F29_ASSERT(approx_over_p(r) < 2.0L, "f29: square output not below 2p");
#endif
  return r;
}

FF_HD fe9 add(const fe9& a, const fe9& b)
{
  fe9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
FF_HD fe9 dbl(const fe9& a)
{
  fe9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] << 1;
  return r;
}
// a + K·p − b.  b's limbs must be ≤ J·(2^29 − 1) and b_8 ≤ (K·p)_8 − J.
template <uint32_t K, uint32_t J>
FF_HD fe9 sub(const fe9& a, const fe9& b)
{
  constexpr Limbs9 C = kp_borrow_proof(K, J);
  fe9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
#if defined(F29_CHECK)
    F29_ASSERT(C.v[i] >= b.l[i], "f29: subtrahend limb exceeds the borrow-proof constant");
    F29_ASSERT((uint64_t)a.l[i] + C.v[i] - b.l[i] < (1ull << 32), "f29: limb overflow in sub");
#endif
    r.l[i] = a.l[i] + (C.v[i] - b.l[i]);
  }
  return r;
}
// K·p − b
template <uint32_t K, uint32_t J>
FF_HD fe9 neg(const fe9& b)
{
  constexpr Limbs9 C = kp_borrow_proof(K, J);
  fe9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
#if defined(F29_CHECK)
    F29_ASSERT(C.v[i] >= b.l[i], "f29: negated limb exceeds the borrow-proof constant");
#endif
    r.l[i] = C.v[i] - b.l[i];
  }
  return r;
}
FF_HD fe9 norm(const fe9& a)
{
  fe9 r;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t t = a.l[i] + carry; // limbs < 2^32 − 2^3: no wrap
#if defined(F29_CHECK)
    F29_ASSERT((uint64_t)a.l[i] + carry < (1ull << 32), "f29: limb wrap in norm");
#endif
    r.l[i] = t & MASK;
    carry = t >> 29;
  }
  r.l[8] = a.l[8] + carry;
  return r;
}

// value < 16·p, N  →  canonical [0, p), N
FF_HD fe9 canon(const fe9& a)
{
  constexpr Limbs9 KP[4] = {kp_plain(8), kp_plain(4), kp_plain(2), kp_plain(1)};
  fe9 v = a;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const Limbs9& C = KP[s];
    fe9 t;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const uint32_t d = v.l[i] - C.v[i] - borrow;
      borrow = d >> 31;
      t.l[i] = i < 8 ? (d & MASK) : d;
    }
    const bool keep = borrow != 0; // v < k·p
#pragma unroll
    for (int i = 0; i < 9; i++) v.l[i] = keep ? v.l[i] : t.l[i];
  }
  return v;
}
// value < 8·p, N  →  value < 2·p, N   (two conditional subtractions: 4p, 2p)
FF_HD fe9 reduce_lt2p(const fe9& a)
{
  constexpr Limbs9 KP[2] = {kp_plain(4), kp_plain(2)};
  fe9 v = a;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const Limbs9& C = KP[s];
    fe9 t;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const uint32_t d = v.l[i] - C.v[i] - borrow;
      borrow = d >> 31;
      t.l[i] = i < 8 ? (d & MASK) : d;
    }
    const bool keep = borrow != 0;
#pragma unroll
    for (int i = 0; i < 9; i++) v.l[i] = keep ? v.l[i] : t.l[i];
  }
#if defined(F29_CHECK)
  F29_ASSERT(approx_over_p(a) < 8.0L && approx_over_p(v) < 2.0L, "f29: reduce_lt2p bound");
#endif
  return v;
}
FF_HD bool is_zero_canon(const fe9& a)
{
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= a.l[i];
  return o == 0;
}
// cheap necessary condition for "a ≡ 0 (mod p)" when a is N and a < 16·p:  a = m·p ⇒ m ≡ a_0·p⁻¹ (mod 2^29), m < 16
FF_HD bool maybe_zero_mod_p(const fe9& a) { return ((a.l[0] * PINV) & MASK) < 16u; }

// 8×32-bit packed (value < 2^256) ↔ 9×29
FF_HD fe9 unpack(const fe& w)
{
  fe9 r;
  r.l[0] = w.l[0] & MASK;
  r.l[1] = ((w.l[0] >> 29) | (w.l[1] << 3)) & MASK;
  r.l[2] = ((w.l[1] >> 26) | (w.l[2] << 6)) & MASK;
  r.l[3] = ((w.l[2] >> 23) | (w.l[3] << 9)) & MASK;
  r.l[4] = ((w.l[3] >> 20) | (w.l[4] << 12)) & MASK;
  r.l[5] = ((w.l[4] >> 17) | (w.l[5] << 15)) & MASK;
  r.l[6] = ((w.l[5] >> 14) | (w.l[6] << 18)) & MASK;
  r.l[7] = ((w.l[6] >> 11) | (w.l[7] << 21)) & MASK;
  r.l[8] = w.l[7] >> 8;
  return r;
}
FF_HD fe pack(const fe9& a) // a must be N with value < 2^256
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

// constants as N-form elements (plain integers mod p); generated with Python big integers, verified by tests/test_f29.py
#define F29_ONE_M 0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x14c0419u, 0xaa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x52ac7a8u, 0xdc836u
#define F29_R2 0x59bac10u, 0xd1503a3u, 0x18016b8u, 0x10ab0ca8u, 0x2632639u, 0x2c0169fu, 0x169bfd53u, 0x11869d4cu, 0x2a11a6u
#define F29_C266 0x13349ca1u, 0x1a5d84a8u, 0xa3e5cacu, 0x100249e0u, 0x12b951e8u, 0xe92d304u, 0x14cb95b3u, 0x41b9d3du, 0x58003u
#define F29_C256 0x58f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0xe0a77u
#define F29_CONST(name, ...)                                                                                           \
  FF_HD fe9 name()                                                                                                     \
  {                                                                                                                    \
    constexpr uint32_t c[9] = {__VA_ARGS__};                                                                           \
    fe9 r;                                                                                                             \
    for (int i = 0; i < 9; i++) r.l[i] = c[i];                                                                         \
    return r;                                                                                                          \
  }
F29_CONST(one_m, F29_ONE_M)       // 2^261 mod p          (Montgomery one)
F29_CONST(r2_m, F29_R2)           // 2^522 mod p          (standard → Montgomery-261: mul(x, r2))
F29_CONST(c256_to_261, F29_C266)  // 2^266 mod p          (x·2^256 → x·2^261: mul(x256, c))
F29_CONST(c261_to_256, F29_C256)  // 2^256 mod p          (x·2^261 → x·2^256: mul(x261, c))
F29_CONST(one_std, 1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u) // x·2^261 → x: mul(x261, 1)

// a⁻¹ (Montgomery-261 in and out; a N, < 2p; 0 ↦ 0) by Fermat, a^(p − 2), two exponent bits per step: 254 squares and ≈ 95
// products on one lane (≈ 80 k instructions) — the reference against which inv_ds below is checked (tests/test_f29.py).
FF_HD fe9 inv(const fe9& a)
{
  constexpr uint32_t E[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}; // p − 2
  const fe9 a2 = sqr(a), a3 = mul(a2, a);
  fe9 r = one_m();
  for (int i = 127; i >= 0; i--) {
    r = sqr(sqr(r));
    const uint32_t d = (E[i >> 4] >> ((i & 15) * 2)) & 3u;
    if (d == 1) r = mul(r, a);
    else if (d == 2) r = mul(r, a2);
    else if (d == 3) r = mul(r, a3);
  }
  return r;
}

// The same inverse by 600 "divsteps" (Bernstein–Yang, https://gcd.cr.yp.to/safegcd-20190413.pdf, in the half-delta form: 590
// steps bound a 256-bit modulus) in 20 batches of 30: a batch runs on the low 30 bits of f and g alone and yields a 2×2
// transition matrix of 31-bit entries, which is then applied to the full-width pairs (f, g) — exact division by 2^30 — and
// (d, e) — division modulo p by adding the multiple of p that clears the low limb.  Invariants: d·x ≡ f, e·x ≡ g (mod p);
// at the end g = 0, f = ±1, so ±d = x⁻¹.  Signed limbs of 30 bits, the top one carrying the sign; everything is straight-line
// code with masks (no data-dependent branches: lanes of a wave stay together).  ≈ 14 k instructions against the ≈ 80 k of the
// Fermat ladder above (ff.h's 8×32 ladder: ≈ 100 k): what the batched projective → affine conversion of the fixed-base table
// builds spends per 32 points (msm_impl.h: batch_to_affine_kernel).  Checked against inv() and ff.h's Fermat inverse by
// tests/test_f29.py.
namespace ds30 {
constexpr int32_t M30 = (1 << 30) - 1;
constexpr int32_t P30[9] = {0x187cfd47, 0x3082305b, 0x71ca8d3, 0x205aa45a, 0x1585d97, 0x116da06, 0x1a029b85, 0x139cb84c, 0x3064};
constexpr uint32_t PINV30 = 0x1b799c77u; // p⁻¹ mod 2^30
struct S30 {
  int32_t v[9];
};
// 30 divsteps on the low limbs; returns the new zeta, t = (u, v, q, r) with 2^30·(f', g') = (u·f + v·g, q·f + r·g)
FF_HD int32_t divsteps30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t t[4])
{
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
  for (int i = 0; i < 30; i++) {
    uint32_t m1 = (uint32_t)(zeta >> 31);   // zeta < 0
    const uint32_t m2 = 0u - (g & 1u);      // g odd
    const uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1; // (f, u, v) negated when zeta < 0
    g += x & m2;
    q += y & m2;
    r += z & m2;
    m1 &= m2;                               // zeta < 0 and g odd: swap roles
    zeta = (int32_t)(((uint32_t)zeta ^ m1) - 1u);
    f += g & m1;
    u += q & m1;
    v += r & m1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t[0] = (int32_t)u;
  t[1] = (int32_t)v;
  t[2] = (int32_t)q;
  t[3] = (int32_t)r;
  return zeta;
}
FF_HD void update_de(S30& d, S30& e, const int32_t t[4])
{
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int32_t di = d.v[0], ei = e.v[0];
  int64_t cd = (int64_t)u * di + (int64_t)v * ei, ce = (int64_t)q * di + (int64_t)r * ei;
  md -= (int32_t)((PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)P30[0] * md;
  ce += (int64_t)P30[0] * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    di = d.v[i];
    ei = e.v[i];
    cd += (int64_t)u * di + (int64_t)v * ei;
    ce += (int64_t)q * di + (int64_t)r * ei;
    cd += (int64_t)P30[i] * md;
    ce += (int64_t)P30[i] * me;
    d.v[i - 1] = (int32_t)cd & M30;
    cd >>= 30;
    e.v[i - 1] = (int32_t)ce & M30;
    ce >>= 30;
  }
  d.v[8] = (int32_t)cd;
  e.v[8] = (int32_t)ce;
}
FF_HD void update_fg(S30& f, S30& g, const int32_t t[4])
{
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  int32_t fi = f.v[0], gi = g.v[0];
  int64_t cf = (int64_t)u * fi + (int64_t)v * gi, cg = (int64_t)q * fi + (int64_t)r * gi;
  cf >>= 30; // (the low 30 bits are zero by construction of t)
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    fi = f.v[i];
    gi = g.v[i];
    cf += (int64_t)u * fi + (int64_t)v * gi;
    cg += (int64_t)q * fi + (int64_t)r * gi;
    f.v[i - 1] = (int32_t)cf & M30;
    cf >>= 30;
    g.v[i - 1] = (int32_t)cg & M30;
    cg >>= 30;
  }
  f.v[8] = (int32_t)cf;
  g.v[8] = (int32_t)cg;
}
} // namespace ds30

// a (Montgomery-261, N, < 16p) → a⁻¹ (Montgomery-261, N, < 2p); 0 ↦ 0
FF_HD fe9 inv_ds(const fe9& a)
{
  using namespace ds30;
  const fe w = pack(canon(a)); // the integer A = a·R' mod p, 256 bits
  S30 d, e, f, g;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 30 * i, wd = bit >> 5, sh = bit & 31;
    uint32_t x = w.l[wd] >> sh;
    if (sh > 2 && wd + 1 < 8) x |= w.l[wd + 1] << (32 - sh);
    g.v[i] = (int32_t)(x & (uint32_t)M30);
    f.v[i] = P30[i];
    d.v[i] = 0;
    e.v[i] = i == 0 ? 1 : 0;
  }
  int32_t zeta = -1;
  for (int it = 0; it < 20; it++) {
    int32_t t[4];
    zeta = divsteps30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    update_de(d, e, t);
    update_fg(f, g, t);
  }
  // d ∈ (−2p, p), f = ±1:  result = sign(f)·d brought into [0, p)
  {
    const int32_t neg = f.v[8] >> 31;
    int32_t add = d.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d.v[i] = ((d.v[i] + (P30[i] & add)) ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      d.v[i + 1] += d.v[i] >> 30;
      d.v[i] &= M30;
    }
    add = d.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d.v[i] += P30[i] & add;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      d.v[i + 1] += d.v[i] >> 30;
      d.v[i] &= M30;
    }
  }
  // 30-bit limbs → 256-bit words → 29-bit limbs; A⁻¹ → a⁻¹·R' = A⁻¹·R'² = mul(A⁻¹, R'³)
  fe o;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int bit = 32 * k, li = bit / 30, sh = bit % 30;
    uint32_t x = (uint32_t)d.v[li] >> sh;
    if (li + 1 < 9) x |= (uint32_t)d.v[li + 1] << (30 - sh);
    if (sh > 28 && li + 2 < 9) x |= (uint32_t)d.v[li + 2] << (60 - sh);
    o.l[k] = x;
  }
  constexpr uint32_t R3[9] = {0xe2312b2u, 0x16c05ca2u, 0xbc84389u, 0x1cdf310bu, 0x11adafddu, 0x32e568eu, 0x1d6ae48cu, 0x10d4cd1fu, 0x26c2d2u}; // 2^783 mod p
  fe9 c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.l[i] = R3[i];
  return mul(unpack(o), c);
}

// packed Montgomery-256 / standard form (canonical) → Montgomery-261 (N, < 2p)
FF_HD fe9 from_mont256(const fe& x) { return mul(unpack(x), c256_to_261()); }
FF_HD fe9 from_std(const fe& x) { return mul(unpack(x), r2_m()); }
// Montgomery-261 (limbs fit the multiplier, value < 147·p/ … any lazy value) → packed canonical Montgomery-256
FF_HD fe to_mont256(const fe9& x) { return pack(canon(mul(x, c261_to_256()))); }

} // namespace f29
} // namespace bn254
