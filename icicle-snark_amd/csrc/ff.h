// ff.h — BN254 prime-field arithmetic for gfx950 (and the host side of the C-ABI).
//
// Layout at the ABI: 8×u32 little-endian limbs, STANDARD form, exactly the reference's
// `storage<8>` (icicle/include/icicle/math/storage.h; wrappers/rust/icicle-core/src/field.rs:10-15).
// Inside kernels values are kept in Montgomery form (R = 2^256): on CDNA4 the product and the
// reduction are both `v_mad_u64_u32` chains of the same shape, and the reduction needs no
// wide-multiply by a Barrett constant.  Every buffer that crosses the ABI is converted at the
// edge of the kernel that touches it (or, for NTT / vec-ops, never: montmul(x_std, w_mont) is
// already the standard-form product).
//
// One thread owns one field element (8 VGPRs); a 256-bit Montgomery multiply is 128+8 32×32→64
// multiply-adds.  Both moduli leave the top two bits of limb 7 clear, so the CIOS accumulator never
// needs a 10th word and sums of two residues never overflow 256 bits.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FF_HD __host__ __device__ __forceinline__
#define FF_HD_NOINLINE __host__ __device__ __noinline__
#else
#define FF_HD inline
#define FF_HD_NOINLINE
#endif

namespace bn254 {

struct alignas(16) fe {
  uint32_t l[8];
};

// Scalar field Fr — icicle/include/icicle/fields/snark_fields/bn254_scalar.h:9-10
struct FrP {
  static constexpr uint32_t MOD[8] = {0xf0000001, 0x43e1f593, 0x79b97091, 0x2833e848,
                                      0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72};
  static constexpr uint32_t NINV = 0xefffffff; // -MOD^-1 mod 2^32
  // R = 2^256 mod r, R2 = 2^512 mod r
  static constexpr uint32_t R[8] = {0x4ffffffb, 0xac96341c, 0x9f60cd29, 0x36fc7695,
                                    0x7879462e, 0x666ea36f, 0x9a07df2f, 0x0e0a77c1};
  static constexpr uint32_t R2[8] = {0xae216da7, 0x1bb8e645, 0xe35c59e3, 0x53fe3ab1,
                                     0x53bb8085, 0x8c49833d, 0x7f4e44a5, 0x0216d0b1};
};
// Base field Fq — icicle/include/icicle/fields/snark_fields/bn254_base.h:8-9
struct FqP {
  static constexpr uint32_t MOD[8] = {0xd87cfd47, 0x3c208c16, 0x6871ca8d, 0x97816a91,
                                      0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72};
  static constexpr uint32_t NINV = 0xe4866389;
  static constexpr uint32_t R[8] = {0xc58f0d9d, 0xd35d438d, 0xf5c70b3d, 0x0a78eb28,
                                    0x7879462c, 0x666ea36f, 0x9a07df2f, 0x0e0a77c1};
  static constexpr uint32_t R2[8] = {0x538afa89, 0xf32cfc5b, 0xd44501fb, 0xb5e71911,
                                     0x0a417ff6, 0x47ab1eff, 0xcab8351f, 0x06d89f71};
};

template <class P>
struct Fp {
  // ---------------------------------------------------------------- constants
  static FF_HD fe zero()
  {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
  }
  static FF_HD fe one_mont()
  {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = P::R[i];
    return r;
  }
  static FF_HD fe r2()
  {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = P::R2[i];
    return r;
  }
  static FF_HD fe one_std()
  {
    fe r = zero();
    r.l[0] = 1;
    return r;
  }
  static FF_HD fe modulus()
  {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = P::MOD[i];
    return r;
  }

  // ---------------------------------------------------------------- predicates
  static FF_HD bool is_zero(const fe& a)
  {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= a.l[i];
    return x == 0;
  }
  static FF_HD bool eq(const fe& a, const fe& b)
  {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= a.l[i] ^ b.l[i];
    return x == 0;
  }

  // ---------------------------------------------------------------- add / sub (same in both forms)
  // r = a - MOD, returns borrow
  static FF_HD uint32_t sub_mod_raw(fe& r, const fe& a)
  {
    uint64_t b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)a.l[i] - P::MOD[i] - b;
      r.l[i] = (uint32_t)d;
      b = (d >> 32) & 1;
    }
    return (uint32_t)b;
  }
  // a < MOD (a canonical residue)
  static FF_HD bool is_canonical(const fe& a)
  {
    fe t;
    return sub_mod_raw(t, a) != 0;
  }
  // conditional final subtraction: a in [0, 2p) -> [0, p)
  static FF_HD fe reduce_once(const fe& a)
  {
    fe t;
    uint32_t borrow = sub_mod_raw(t, a);
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : t.l[i];
    return r;
  }
  static FF_HD fe add(const fe& a, const fe& b)
  {
    fe s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      c += (uint64_t)a.l[i] + b.l[i];
      s.l[i] = (uint32_t)c;
      c >>= 32;
    }
    return reduce_once(s); // a+b < 2p < 2^255: no carry out
  }
  static FF_HD fe sub(const fe& a, const fe& b)
  {
    fe d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t t = (uint64_t)a.l[i] - b.l[i] - br;
      d.l[i] = (uint32_t)t;
      br = (t >> 32) & 1;
    }
    // add back p when the subtraction borrowed
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      c += (uint64_t)d.l[i] + (P::MOD[i] & mask);
      r.l[i] = (uint32_t)c;
      c >>= 32;
    }
    return r;
  }
  static FF_HD fe neg(const fe& a)
  {
    return sub(zero(), a);
  }
  static FF_HD fe dbl(const fe& a) { return add(a, a); }

  // ---------------------------------------------------------------- Montgomery multiply
  // returns a·b·R^-1 mod p, canonical, for a,b < p.
  //
  // Device (gfx950): product-scanning ("FIPS") form.  Column k of a·b + m·p is summed into a 96-bit
  // accumulator with  v_mad_u64_u32 acc, vcc, x, y, acc ; v_addc_co_u32 top, vcc, 0, top, vcc  — the
  // 64-bit accumulator is both addend and destination (no register moves, no zero-extension), the
  // carry-out of the 64-bit add is folded into `top`.  136 multiply-adds + 128 carry captures per
  // product; measured 125 G Fq-mul/s on MI355X vs 99 G for the compiler-scheduled CIOS below
  // (v_mad_u64_u32 issues at half rate, 4 cycles per wave64).
  // Host: operand-scanning CIOS in portable C++ (same results).
#if defined(__HIP_DEVICE_COMPILE__)
  static __device__ __forceinline__ void mac96(uint64_t& acc, uint32_t& top, uint32_t a, uint32_t b)
  {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(top) : "v"(a), "v"(b) : "vcc");
  }
  static __device__ __forceinline__ void mac96s(uint64_t& acc, uint32_t& top, uint32_t a, uint32_t b_const)
  {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(top) : "v"(a), "s"(b_const) : "vcc");
  }
  static __device__ __forceinline__ fe mul(const fe& a, const fe& b)
  {
    uint64_t acc = 0;
    uint32_t top = 0;
    uint32_t m[8];
    fe r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) mac96(acc, top, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = 0; i < k; i++) mac96s(acc, top, m[i], P::MOD[k - i]);
      m[k] = (uint32_t)acc * P::NINV;
      mac96s(acc, top, m[k], P::MOD[0]); // low word becomes 0
      acc = (acc >> 32) | ((uint64_t)top << 32);
      top = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96(acc, top, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96s(acc, top, m[i], P::MOD[k - i]);
      r.l[k - 8] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)top << 32);
      top = 0;
    }
    return reduce_once(r); // a·b·R^-1 < 2p < 2^255: nothing left in acc
  }
  // (a·b + c·d)·R^-1 mod p in ONE reduction: both products are summed into the same column accumulators
  // (a·b + c·d < 2p² < p·R/2, so the Montgomery quotient still leaves a value < 2p).  192 multiply-adds
  // instead of 272 for two multiplies and an addition; used by the Fq2 multiply (ec.h).
  static __device__ __forceinline__ fe mul2sum(const fe& a, const fe& b, const fe& c, const fe& d)
  {
    uint64_t acc = 0;
    uint32_t top = 0;
    uint32_t m[8];
    fe r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        mac96(acc, top, a.l[i], b.l[k - i]);
        mac96(acc, top, c.l[i], d.l[k - i]);
      }
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
      for (int i = k - 7; i < 8; i++) {
        mac96(acc, top, a.l[i], b.l[k - i]);
        mac96(acc, top, c.l[i], d.l[k - i]);
      }
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96s(acc, top, m[i], P::MOD[k - i]);
      r.l[k - 8] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)top << 32);
      top = 0;
    }
    return reduce_once(r);
  }
#else
  static FF_HD fe mul(const fe& a, const fe& b)
  {
    uint32_t t[8];
    for (int i = 0; i < 8; i++) t[i] = 0;
    uint32_t t8 = 0;
    for (int i = 0; i < 8; i++) {
      // t += a * b[i]
      uint64_t c = 0;
      const uint32_t bi = b.l[i];
      for (int j = 0; j < 8; j++) {
        uint64_t s = (uint64_t)a.l[j] * bi + t[j] + c;
        t[j] = (uint32_t)s;
        c = s >> 32;
      }
      uint32_t top = t8 + (uint32_t)c; // no overflow: modulus has two spare top bits
      // m = t[0] * (-p^-1) mod 2^32 ; t = (t + m*p) / 2^32
      const uint32_t m = t[0] * P::NINV;
      uint64_t s = (uint64_t)m * P::MOD[0] + t[0];
      c = s >> 32;
      for (int j = 1; j < 8; j++) {
        s = (uint64_t)m * P::MOD[j] + t[j] + c;
        t[j - 1] = (uint32_t)s;
        c = s >> 32;
      }
      s = (uint64_t)top + c;
      t[7] = (uint32_t)s;
      t8 = (uint32_t)(s >> 32);
    }
    fe r;
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return reduce_once(r);
  }
  static FF_HD fe mul2sum(const fe& a, const fe& b, const fe& c, const fe& d) { return add(mul(a, b), mul(c, d)); }
#endif
  static FF_HD fe sqr(const fe& a) { return mul(a, a); }

  // form conversions
  static FF_HD fe to_mont(const fe& a) { return mul(a, r2()); }
  static FF_HD fe from_mont(const fe& a) { return mul(a, one_std()); }

  // multiply by small constants via additions
  static FF_HD fe mul3(const fe& a) { return add(dbl(a), a); }

  // a^e for a 256-bit exponent given as limbs (host-side helper, also usable on device)
  static FF_HD fe pow(const fe& a, const fe& e)
  {
    fe acc = one_mont(), base = a;
    for (int i = 0; i < 256; i++) {
      if ((e.l[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base);
      base = sqr(base);
    }
    return acc;
  }
  // Montgomery-form inverse by Fermat; inverse(0) = 0 like the reference
  // (icicle/include/icicle/math/modular_arithmetic.h:601-603).
  static FF_HD fe inv(const fe& a)
  {
    fe e = modulus();
    e.l[0] -= 2; // both moduli have l[0] >= 2
    return pow(a, e);
  }
};

typedef Fp<FrP> Fr;
typedef Fp<FqP> Fq;

} // namespace bn254
