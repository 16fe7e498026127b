// ec.h — Fq2 and BN254 G1/G2 group law (host + gfx950 device), Montgomery-form coordinates.
//
// Two coordinate systems:
//  * XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity ⇔ ZZ = 0) for the MSM kernels: a mixed
//    add with an affine base is 8M + 2S and needs no field inversion.  Exceptional cases (identity
//    accumulator, P + P, P − P) are handled explicitly — repeated bases and 0/1 witnesses make
//    doublings common in real provers (cf. the skewed cases of wrappers/rust/icicle-core/src/msm/tests.rs:254-302).
//  * homogeneous projective (X:Y:Z), identity (0,1,0), with the reference's complete formulas
//    (icicle/include/icicle/curves/projective.h:54-169) for the host-side FFI (bn254_ecadd, …) so
//    that results carry the very same representative the reference returns.
//
// The ABI types are the reference's: Affine{x,y}, Projective{x,y,z}, coordinates standard form,
// affine identity = (0,0) (icicle/include/icicle/curves/affine.h; wrappers/rust/icicle-core/src/curve.rs:45-59,104-111).
#pragma once
#include "ff.h"

namespace bn254 {

struct alignas(16) fe2 {
  fe c0, c1;
};

// ------------------------------------------------------------------------------------------------
// uniform field interface used by the curve templates
// ------------------------------------------------------------------------------------------------
struct FqOps {
  typedef fe T;
  static constexpr int NFE = 1; // number of Fq limbs-blocks per element
  static FF_HD T zero() { return Fq::zero(); }
  static FF_HD T one() { return Fq::one_mont(); }
  static FF_HD T add(const T& a, const T& b) { return Fq::add(a, b); }
  static FF_HD T sub(const T& a, const T& b) { return Fq::sub(a, b); }
  static FF_HD T neg(const T& a) { return Fq::neg(a); }
  static FF_HD T dbl(const T& a) { return Fq::dbl(a); }
  static FF_HD T mul(const T& a, const T& b) { return Fq::mul(a, b); }
  static FF_HD T sqr(const T& a) { return Fq::sqr(a); }
  static FF_HD bool is_zero(const T& a) { return Fq::is_zero(a); }
  static FF_HD bool eq(const T& a, const T& b) { return Fq::eq(a, b); }
  static FF_HD T to_mont(const T& a) { return Fq::to_mont(a); }
  static FF_HD T from_mont(const T& a) { return Fq::from_mont(a); }
  static FF_HD T inv(const T& a) { return Fq::inv(a); }
  // a·b − c·d with a single Montgomery reduction
  static FF_HD T mul_sub_mul(const T& a, const T& b, const T& c, const T& d) { return Fq::mul2sum(a, b, Fq::neg(c), d); }
};

// Fq2 = Fq[u]/(u² + 1) — icicle/include/icicle/fields/complex_extension.h, nonresidue −1
// (icicle/include/icicle/fields/snark_fields/bn254_base.h:66-70)
// Fq2 multiply / square: out-of-line calls on the device by default (bounds code size and compile time of
// the many kernels that contain full XYZZ additions); the translation unit of the hot G2 bucket-accumulation
// kernel defines ISNARK_FQ2_INLINE and gets them inlined (no call-boundary register shuffles, no scratch).
#if defined(ISNARK_FQ2_INLINE) && defined(__HIP_DEVICE_COMPILE__)
#define FQ2_FN __device__ __forceinline__
#else
#define FQ2_FN FF_HD_NOINLINE
#endif

struct Fq2Ops {
  typedef fe2 T;
  static constexpr int NFE = 2;
  static FF_HD T zero() { return {Fq::zero(), Fq::zero()}; }
  static FF_HD T one() { return {Fq::one_mont(), Fq::zero()}; }
  static FF_HD T add(const T& a, const T& b) { return {Fq::add(a.c0, b.c0), Fq::add(a.c1, b.c1)}; }
  static FF_HD T sub(const T& a, const T& b) { return {Fq::sub(a.c0, b.c0), Fq::sub(a.c1, b.c1)}; }
  static FF_HD T neg(const T& a) { return {Fq::neg(a.c0), Fq::neg(a.c1)}; }
  static FF_HD T dbl(const T& a) { return {Fq::dbl(a.c0), Fq::dbl(a.c1)}; }
  // Out of line on the device: a G2 point addition inlines ~14 of these (3 Fq multiplies each);
  // keeping them as calls bounds code size and compile time at no measurable cost (the call
  // overhead is ~1 % of the ~1.6 k instructions inside).
  static FQ2_FN T mul(T a, T b) // operands BY VALUE: they travel in VGPRs; references would go through scratch
  {
    // (a0 b0 − a1 b1) + (a0 b1 + a1 b0) u, each component as ONE fused two-product Montgomery reduction
    // (Fq::mul2sum): 2 × 192 multiply-adds, against 3 × 136 + five modular add/subs for Karatsuba.
    return {Fq::mul2sum(a.c0, b.c0, a.c1, Fq::neg(b.c1)), Fq::mul2sum(a.c0, b.c1, a.c1, b.c0)};
  }
  static FQ2_FN T sqr(T a)
  {
    // (a0 + a1)(a0 − a1) + 2 a0 a1 u
    fe t = Fq::mul(a.c0, a.c1);
    fe r0 = Fq::mul(Fq::add(a.c0, a.c1), Fq::sub(a.c0, a.c1));
    return {r0, Fq::dbl(t)};
  }
  static FF_HD bool is_zero(const T& a) { return Fq::is_zero(a.c0) && Fq::is_zero(a.c1); }
  static FF_HD bool eq(const T& a, const T& b) { return Fq::eq(a.c0, b.c0) && Fq::eq(a.c1, b.c1); }
  static FF_HD T to_mont(const T& a) { return {Fq::to_mont(a.c0), Fq::to_mont(a.c1)}; }
  static FF_HD T from_mont(const T& a) { return {Fq::from_mont(a.c0), Fq::from_mont(a.c1)}; }
  static FF_HD T mul_sub_mul(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
  static FF_HD T inv(const T& a)
  {
    fe n = Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1));
    fe d = Fq::inv(n);
    return {Fq::mul(a.c0, d), Fq::neg(Fq::mul(a.c1, d))};
  }
};

// ------------------------------------------------------------------------------------------------
// point types (generic over the coordinate field)
// ------------------------------------------------------------------------------------------------
template <class F>
struct Affine {
  typename F::T x, y;
};
template <class F>
struct Projective {
  typename F::T x, y, z;
};
template <class F>
struct XYZZ {
  typename F::T x, y, zz, zzz;
};

template <class F>
struct Curve {
  typedef typename F::T T;
  typedef Affine<F> A;
  typedef Projective<F> P;
  typedef XYZZ<F> X;

  static FF_HD bool aff_is_zero(const A& a) { return F::is_zero(a.x) && F::is_zero(a.y); }
  static FF_HD A aff_to_mont(const A& a) { return {F::to_mont(a.x), F::to_mont(a.y)}; }
  static FF_HD A aff_neg(const A& a) { return {a.x, F::neg(a.y)}; }

  // ---------------------------------------------------------------- XYZZ
  static FF_HD X x_zero() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  static FF_HD bool x_is_zero(const X& p) { return F::is_zero(p.zz); }
  static FF_HD X x_from_affine(const A& a) // a != identity, Montgomery form
  {
    return {a.x, a.y, F::one(), F::one()};
  }
  static FF_HD X x_neg(const X& p) { return {p.x, F::neg(p.y), p.zz, p.zzz}; }

  // 2·(affine) — EFD mdbl-2008-s-1 (a = 0)
  static FF_HD X x_dbl_affine(const A& a)
  {
    T U = F::dbl(a.y);
    T V = F::sqr(U);
    T W = F::mul(U, V);
    T S = F::mul(a.x, V);
    T xx = F::sqr(a.x);
    T M = F::add(F::dbl(xx), xx);
    T X3 = F::sub(F::sqr(M), F::dbl(S));
    T Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, a.y));
    return {X3, Y3, V, W};
  }
  // 2·(XYZZ) — EFD dbl-2008-s-1 (a = 0)
  static FF_HD X x_dbl(const X& p)
  {
    if (x_is_zero(p)) return p;
    T U = F::dbl(p.y);
    T V = F::sqr(U);
    T W = F::mul(U, V);
    T S = F::mul(p.x, V);
    T xx = F::sqr(p.x);
    T M = F::add(F::dbl(xx), xx);
    T X3 = F::sub(F::sqr(M), F::dbl(S));
    T Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, p.y));
    return {X3, Y3, F::mul(V, p.zz), F::mul(W, p.zzz)};
  }
  // XYZZ += affine (Montgomery form, not the identity) — EFD madd-2008-s
  static FF_HD void x_madd(X& acc, const A& b)
  {
    if (x_is_zero(acc)) {
      acc = x_from_affine(b);
      return;
    }
    T U2 = F::mul(b.x, acc.zz);
    T S2 = F::mul(b.y, acc.zzz);
    T Pd = F::sub(U2, acc.x);
    T Rd = F::sub(S2, acc.y);
    if (F::is_zero(Pd)) {
      if (F::is_zero(Rd)) acc = x_dbl_affine(b);
      else acc = x_zero();
      return;
    }
    T PP = F::sqr(Pd);
    T PPP = F::mul(Pd, PP);
    T Q = F::mul(acc.x, PP);
    T X3 = F::sub(F::sub(F::sqr(Rd), PPP), F::dbl(Q));
    T Y3 = F::mul_sub_mul(Rd, F::sub(Q, X3), acc.y, PPP); // R·(Q − X3) − Y1·PPP, one reduction (G1)
    acc.x = X3;
    acc.y = Y3;
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
  }
  // XYZZ + XYZZ — EFD add-2008-s
  static FF_HD X x_add(const X& a, const X& b)
  {
    if (x_is_zero(a)) return b;
    if (x_is_zero(b)) return a;
    T U1 = F::mul(a.x, b.zz);
    T U2 = F::mul(b.x, a.zz);
    T S1 = F::mul(a.y, b.zzz);
    T S2 = F::mul(b.y, a.zzz);
    T Pd = F::sub(U2, U1);
    T Rd = F::sub(S2, S1);
    if (F::is_zero(Pd)) {
      if (F::is_zero(Rd)) return x_dbl(a);
      return x_zero();
    }
    T PP = F::sqr(Pd);
    T PPP = F::mul(Pd, PP);
    T Q = F::mul(U1, PP);
    T X3 = F::sub(F::sub(F::sqr(Rd), PPP), F::dbl(Q));
    T Y3 = F::sub(F::mul(Rd, F::sub(Q, X3)), F::mul(S1, PPP));
    T ZZ3 = F::mul(F::mul(a.zz, b.zz), PP);
    T ZZZ3 = F::mul(F::mul(a.zzz, b.zzz), PPP);
    return {X3, Y3, ZZ3, ZZZ3};
  }
  // XYZZ (Montgomery) → reference Projective (Montgomery): (X·ZZZ : Y·ZZ : ZZ·ZZZ); identity → (0,1,0)
  static FF_HD P x_to_projective(const X& p)
  {
    if (x_is_zero(p)) return {F::zero(), F::one(), F::zero()};
    return {F::mul(p.x, p.zzz), F::mul(p.y, p.zz), F::mul(p.zz, p.zzz)};
  }
  static FF_HD X x_from_projective(const P& p) // (X:Y:Z) → (X·Z, Y·Z², Z², Z³)
  {
    if (F::is_zero(p.z)) return x_zero();
    T z2 = F::sqr(p.z);
    return {F::mul(p.x, p.z), F::mul(p.y, z2), z2, F::mul(z2, p.z)};
  }

  // ---------------------------------------------------------------- reference projective (host FFI)
  static FF_HD P p_zero() { return {F::zero(), F::one(), F::zero()}; }
  static FF_HD P p_to_mont(const P& p) { return {F::to_mont(p.x), F::to_mont(p.y), F::to_mont(p.z)}; }
  static FF_HD P p_from_mont(const P& p) { return {F::from_mont(p.x), F::from_mont(p.y), F::from_mont(p.z)}; }
  static FF_HD P p_neg(const P& p) { return {p.x, F::neg(p.y), p.z}; }
  static FF_HD P p_from_affine(const A& a) // projective.h:34-37
  {
    if (aff_is_zero(a)) return p_zero();
    return {a.x, a.y, F::one()};
  }
  // projective.h:54-80 ; b3 = 3·b in Montgomery form
  static FF_HD P p_dbl(const P& p, const T& b3)
  {
    T t0 = F::sqr(p.y);
    T Z3 = F::dbl(F::dbl(F::dbl(t0)));
    T t1 = F::mul(p.y, p.z);
    T t2 = F::mul(b3, F::sqr(p.z));
    T X3 = F::mul(t2, Z3);
    T Y3 = F::add(t0, t2);
    Z3 = F::mul(t1, Z3);
    t1 = F::dbl(t2);
    t2 = F::add(t1, t2);
    t0 = F::sub(t0, t2);
    Y3 = F::add(X3, F::mul(t0, Y3));
    t1 = F::mul(p.x, p.y);
    X3 = F::dbl(F::mul(t0, t1));
    return {X3, Y3, Z3};
  }
  // projective.h:82-128
  static FF_HD P p_add(const P& p1, const P& p2, const T& b3)
  {
    T t00 = F::mul(p1.x, p2.x);
    T t01 = F::mul(p1.y, p2.y);
    T t02 = F::mul(p1.z, p2.z);
    T t07 = F::sub(F::mul(F::add(p1.x, p1.y), F::add(p2.x, p2.y)), F::add(t00, t01));
    T t12 = F::sub(F::mul(F::add(p1.y, p1.z), F::add(p2.y, p2.z)), F::add(t01, t02));
    T t17 = F::sub(F::mul(F::add(p1.x, p1.z), F::add(p2.x, p2.z)), F::add(t00, t02));
    T t19 = F::add(F::dbl(t00), t00);
    T t20 = F::mul(b3, t02);
    T t21 = F::add(t01, t20);
    T t22 = F::sub(t01, t20);
    T t23 = F::mul(b3, t17);
    T X3 = F::sub(F::mul(t07, t22), F::mul(t12, t23));
    T Y3 = F::add(F::mul(t22, t21), F::mul(t23, t19));
    T Z3 = F::add(F::mul(t21, t12), F::mul(t19, t07));
    return {X3, Y3, Z3};
  }
  // projective.h:210-213
  static FF_HD bool p_eq(const P& a, const P& b)
  {
    return F::eq(F::mul(a.x, b.z), F::mul(b.x, a.z)) && F::eq(F::mul(a.y, b.z), F::mul(b.y, a.z));
  }
  // projective.h:27-31 (inverse(0) = 0 ⇒ identity ↦ (0,0))
  static FF_HD A p_to_affine(const P& p)
  {
    T d = F::inv(p.z);
    return {F::mul(p.x, d), F::mul(p.y, d)};
  }
};

typedef Curve<FqOps> G1;
typedef Curve<Fq2Ops> G2;

} // namespace bn254
