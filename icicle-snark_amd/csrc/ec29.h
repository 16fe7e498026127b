// ec29.h — XYZZ bucket arithmetic on the lazy radix-2^29 field (ff29.h) for the MSM hot loops.
//
// Same formulas as ec.h (EFD madd-2008-s / add-2008-s / dbl-2008-s-1, a = 0), re-stated with explicit bounds.
// INVARIANT of every stored coordinate (accumulator X, Y, ZZ, ZZZ; per Fq2 component for G2):
//      limbs N (l[0..7] < 2^29),  value < 2·p (G1's X: < 7·p, see below),  Montgomery form with R' = 2^261.
// The identity is the all-zero tuple (a live ZZ is a product of non-zero field elements: never ≡ 0, so its limbs
// are never all zero).  Affine inputs obey the same invariant (canonical values from memory, or mul outputs).
//
// Bound bookkeeping for x_madd (per component; "<k" means value < k·p; out(m) = mul output bound 1 + m/147):
//   U2, S2            mul of two <2 values                                  → N, <2
//   P = U2 + 3p − X1  sub<3,1>: X1 N, X1_8 ≤ (2p)_8 ≤ (3p)_8 − 1           → limbs < 2^29 + 2^30, <5   → norm → Pn
//   R likewise                                                              → Rn N, <5
//   PP = Pn², PPP = Pn·PP, Q = X1·PP, RR = Rn²                              → N, <2 (tighter: RR<1.75, PPP<1.17, Q<1.07)
//   T = PPP + 2Q      limbs ≤ 3·(2^29 − 1), <3.4
//   X3 = RR + 5p − T  sub<5,3>: T_8 ≤ (3.4p)_8 ≤ (5p)_8 − 3                 → limbs < 2^32, <6.75 → norm → reduce_lt2p → N, <2
//   D = Q + 3p − X3   sub<3,1>                                              → limbs < 2^29 + 2^30, <5
//   Y3 = Rn·D + (3p − Y1)·PPP   one reduction (G1: mul2; G2: mul4 on normalised operands) → N, <2
//   ZZ3 = ZZ1·PP, ZZZ3 = ZZZ1·PPP                                           → N, <2
// G1 since round 5: X is the one coordinate that is not a product, and the two conditional subtractions that brought it back
// below 2p (reduce_lt2p: 90 of the ≈ 740 non-multiplier instructions of a mixed addition) are gone — an X is N and < 7·p
// (XBOUND), and whoever subtracts an X does it against 8p (subx):
//   P = U2 + 8p − X1  <10 → Pn;  PP = Pn² (100 p² < 147 p²), PPP = Pn·PP (20 p²), Q = X1·PP (14 p²)
//   X3 = RR + 5p − T  <6.75 → norm → N, <7          D = Q + 8p − X3  <10, limbs < 2^29 + 2^30
//   Y3 = Rn·D + (3p − Y1)·PPP: 50 p² + 3.5 p²; columns 9·(2^29·3·2^29 + 2^30·2^29 + 2^58) = 54·2^58 < 2^64 as before
// G2 keeps the reduction: its Fq2 square (a0 + a1)(a0 + 6p − a1) has no room for components beyond 5p.
// Fq2 products are formed as  c0 = a0·b0 + a1·(3p − b1),  c1 = a0·b1 + a1·b0  (mul2, one reduction each) and squares
// as  c0 = (a0 + a1)(a0 + 6p − a1),  c1 = a0·(2·a1)  with a <5 N:  (a0+a1) <10 normalised, (a0 + 6p − a1) <11 → 110 p² < 147 p².
// `F29_CHECK` host builds assert every limb/column/value condition (tests/test_f29.py drives them, incl. extremal inputs).
#pragma once
#include "ec.h"
#include "ff29.h"

namespace bn254 {

struct fe9x2 {
  fe9 c0, c1;
};

// ---------------------------------------------------------------------------------------------- field layers
struct Fq29 {
  typedef fe9 T;   // lazy element
  typedef fe PK;   // packed 32-byte element in memory
  typedef FqOps Old;
  static FF_HD T zero()
  {
    T r;
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
  }
  static FF_HD T one() { return f29::one_m(); }
  static FF_HD bool is_all_zero(const T& a) { return f29::is_zero_canon(a); }
  static FF_HD T mul(const T& a, const T& b) { return f29::mul(a, b); }
  static FF_HD T sqr_n(const T& a) { return f29::sqr(a); }                        // a N, <12
  static FF_HD T sub3(const T& a, const T& b) { return f29::sub<3, 1>(a, b); }    // a + 3p − b, b N <2
  static FF_HD T subx(const T& a, const T& x) { return f29::sub<8, 1>(a, x); }    // a + 8p − x, x an X coordinate: N, <7
  static FF_HD T norm(const T& a) { return f29::norm(a); }
  static FF_HD T dbl(const T& a) { return f29::dbl(a); }
  static FF_HD T tripled(const T& a) { return f29::add(f29::dbl(a), a); }         // a N → limbs < 3·2^29, <6
  static FF_HD T lt2p(const T& a) { return f29::reduce_lt2p(f29::norm(a)); }      // a <8 → N, <2
  static FF_HD T x3(const T& RR, const T& PPP, const T& Q)                         // RR − PPP − 2Q → N, <7 (RR <2 + 5p): no reduction
  {
    const T r = f29::norm(f29::sub<5, 3>(RR, f29::add(PPP, f29::dbl(Q))));
#if defined(F29_CHECK)
    F29_ASSERT(f29::approx_over_p(r) < 7.0L, "f29: X3 not below 7p");
#endif
    return r;
  }
  static FF_HD T y3(const T& Rn, const T& D, const T& Y1, const T& PPP) { return f29::mul2(Rn, D, f29::neg<3, 1>(Y1), PPP); }
  static FF_HD bool maybe_zero(const T& an) { return f29::maybe_zero_mod_p(an); }  // an N, <16
  static FF_HD bool is_zero_full(const T& an) { return f29::is_zero_canon(f29::canon(an)); }
  // exact (canonical) helpers for the rare paths; inputs N <2, outputs canonical N
  static FF_HD T c_add(const T& a, const T& b) { return f29::canon(f29::norm(f29::add(a, b))); }
  static FF_HD T c_sub(const T& a, const T& b) { return f29::canon(f29::norm(f29::sub<3, 1>(a, b))); }
  static FF_HD T c_mul(const T& a, const T& b) { return f29::canon(f29::mul(a, b)); }
  // memory forms
  static FF_HD T load_internal(const PK& x) { return f29::unpack(x); }           // packed canonical Montgomery-261
  static FF_HD T load_mont256(const PK& x) { return f29::from_mont256(x); }
  static FF_HD T load_std(const PK& x) { return f29::from_std(x); }
  static FF_HD PK store_mont256(const T& x) { return f29::to_mont256(x); }        // → packed canonical Montgomery-256
  static FF_HD PK store_internal(const T& x) { return f29::pack(f29::canon(x)); } // x N <16
  static FF_HD T neg_canon(const T& y) { return f29::neg<2, 1>(y); }              // y N <1 (canonical) → 2p − y, limbs < 2^30, <2
};

struct Fq2_29 {
  typedef fe9x2 T;
  typedef fe2 PK;
  typedef Fq2Ops Old;
  static FF_HD T zero() { return {Fq29::zero(), Fq29::zero()}; }
  static FF_HD T one() { return {f29::one_m(), Fq29::zero()}; }
  static FF_HD bool is_all_zero(const T& a) { return f29::is_zero_canon(a.c0) && f29::is_zero_canon(a.c1); }
  // a: limbs such that the mul2 column bound holds against N / <2^30 partners (see header); b N, <2
  static FF_HD T mul(const T& a, const T& b)
  {
    const fe9 nb1 = f29::neg<3, 1>(b.c1); // <3, limbs < 2^30
    return {f29::mul2(a.c0, b.c0, a.c1, nb1), f29::mul2(a.c0, b.c1, a.c1, b.c0)};
  }
  static FF_HD T sqr_n(const T& a) // a N, <5 per component
  {
    const fe9 s = f29::norm(f29::add(a.c0, a.c1));  // N, <10
    const fe9 d = f29::sub<6, 1>(a.c0, a.c1);        // <11, limbs < 2^29 + 2^30
    return {f29::mul(s, d), f29::mul(a.c0, f29::dbl(a.c1))};
  }
  static FF_HD T sub3(const T& a, const T& b) { return {f29::sub<3, 1>(a.c0, b.c0), f29::sub<3, 1>(a.c1, b.c1)}; }
  static FF_HD T subx(const T& a, const T& x) { return sub3(a, x); } // (an Fq2 X is reduced below 2p: see x3)
  static FF_HD T norm(const T& a) { return {f29::norm(a.c0), f29::norm(a.c1)}; }
  static FF_HD T dbl(const T& a) { return {f29::dbl(a.c0), f29::dbl(a.c1)}; }
  static FF_HD T tripled(const T& a) { return {Fq29::tripled(a.c0), Fq29::tripled(a.c1)}; }
  static FF_HD T lt2p(const T& a) { return {Fq29::lt2p(a.c0), Fq29::lt2p(a.c1)}; }
  static FF_HD T x3(const T& RR, const T& PPP, const T& Q) // per component RR − PPP − 2Q → N, <2
  {
    return {f29::reduce_lt2p(Fq29::x3(RR.c0, PPP.c0, Q.c0)), f29::reduce_lt2p(Fq29::x3(RR.c1, PPP.c1, Q.c1))};
  }
  // Y3 = R·D − Y1·PPP, four products per component in one reduction; D is normalised first (column bound 36·2^58)
  static FF_HD T y3(const T& Rn, const T& D, const T& Y1, const T& PPP)
  {
    const T Dn = norm(D);                                  // N, <5
    const fe9 nd1 = f29::norm(f29::neg<6, 1>(Dn.c1));      // N, <6
    const fe9 ny0 = f29::norm(f29::neg<3, 1>(Y1.c0));      // N, <3
    const fe9 ny1 = f29::norm(f29::neg<3, 1>(Y1.c1));      // N, <3
    // c0 = R0·D0 − R1·D1 − Y0·PPP0 + Y1·PPP1 ;  c1 = R0·D1 + R1·D0 − Y0·PPP1 − Y1·PPP0        (Σ < 25+30+6+4 = 65 p²)
    return {f29::mul4(Rn.c0, Dn.c0, Rn.c1, nd1, ny0, PPP.c0, Y1.c1, PPP.c1), f29::mul4(Rn.c0, Dn.c1, Rn.c1, Dn.c0, ny0, PPP.c1, ny1, PPP.c0)};
  }
  static FF_HD bool maybe_zero(const T& an) { return f29::maybe_zero_mod_p(an.c0) && f29::maybe_zero_mod_p(an.c1); }
  static FF_HD bool is_zero_full(const T& an) { return Fq29::is_zero_full(an.c0) && Fq29::is_zero_full(an.c1); }
  static FF_HD T c_add(const T& a, const T& b) { return {Fq29::c_add(a.c0, b.c0), Fq29::c_add(a.c1, b.c1)}; }
  static FF_HD T c_sub(const T& a, const T& b) { return {Fq29::c_sub(a.c0, b.c0), Fq29::c_sub(a.c1, b.c1)}; }
  static FF_HD T c_mul(const T& a, const T& b)
  {
    const T r = mul(a, b);
    return {f29::canon(r.c0), f29::canon(r.c1)};
  }
  static FF_HD T load_internal(const PK& x) { return {f29::unpack(x.c0), f29::unpack(x.c1)}; }
  static FF_HD T load_mont256(const PK& x) { return {f29::from_mont256(x.c0), f29::from_mont256(x.c1)}; }
  static FF_HD T load_std(const PK& x) { return {f29::from_std(x.c0), f29::from_std(x.c1)}; }
  static FF_HD PK store_mont256(const T& x) { return {f29::to_mont256(x.c0), f29::to_mont256(x.c1)}; }
  static FF_HD PK store_internal(const T& x) { return {f29::pack(f29::canon(x.c0)), f29::pack(f29::canon(x.c1))}; }
  static FF_HD T neg_canon(const T& y) { return {f29::neg<2, 1>(y.c0), f29::neg<2, 1>(y.c1)}; }
};

// ---------------------------------------------------------------------------------------------- curve layer
template <class F>
struct CurveL {
  typedef typename F::T T;
  typedef Curve<typename F::Old> Old; // packed / Montgomery-256 types of ec.h
  struct A {
    T x, y;
  };
  struct X {
    T x, y, zz, zzz;
  };
  enum Form { STD = 0, MONT256 = 1, INTERNAL = 2 }; // encodings of affine bases in memory

  static FF_HD X x_zero() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  static FF_HD bool x_is_zero(const X& p) { return F::is_all_zero(p.zz); }

  // packed affine point (not the identity) → lazy affine; `negate` flips the sign (y ← 2p − y: limbs < 2^30, <2)
  static FF_HD A load_affine(const typename Old::A& p, int form, bool negate)
  {
    A r;
    if (form == INTERNAL) {
      r.x = F::load_internal(p.x);
      r.y = F::load_internal(p.y);
      if (negate) r.y = F::neg_canon(r.y); // canonical (<1) → 2p − y: limbs < 2^30, <2
    } else {
      // standard / Montgomery-256 inputs are canonical in memory: negate there (p − y), then convert (one multiply each)
      const typename F::PK y = negate ? F::Old::neg(p.y) : p.y;
      r.x = form == MONT256 ? F::load_mont256(p.x) : F::load_std(p.x);
      r.y = form == MONT256 ? F::load_mont256(y) : F::load_std(y);
    }
    return r;
  }

  // exact formulas for the rare branches (canonical arithmetic; speed is irrelevant here)
  static FF_HD X x_dbl_affine_exact(const A& a)
  {
    const T x = F::c_add(F::norm(a.x), F::zero()), y = F::c_add(F::norm(a.y), F::zero());
    const T U = F::c_add(y, y);
    const T V = F::c_mul(U, U);
    const T W = F::c_mul(U, V);
    const T S = F::c_mul(x, V);
    const T xx = F::c_mul(x, x);
    const T M = F::c_add(F::c_add(xx, xx), xx);
    const T X3 = F::c_sub(F::c_mul(M, M), F::c_add(S, S));
    const T Y3 = F::c_sub(F::c_mul(M, F::c_sub(S, X3)), F::c_mul(W, y));
    return {X3, Y3, V, W};
  }

  // XYZZ += affine (not the identity)
  static FF_HD void x_madd(X& acc, const A& b)
  {
    if (x_is_zero(acc)) {
      acc = {F::norm(b.x), F::norm(b.y), F::one(), F::one()};
      return;
    }
    const T U2 = F::mul(b.x, acc.zz);
    const T S2 = F::mul(b.y, acc.zzz);
    const T Pn = F::norm(F::subx(U2, acc.x));
    const T Rn = F::norm(F::sub3(S2, acc.y));
    if (F::maybe_zero(Pn) && F::is_zero_full(Pn)) { // same x: doubling or cancellation
      if (F::is_zero_full(Rn)) acc = x_dbl_affine_exact(b);
      else acc = x_zero();
      return;
    }
    const T PP = F::sqr_n(Pn);
    const T PPP = F::mul(Pn, PP);
    const T Q = F::mul(acc.x, PP);
    const T RR = F::sqr_n(Rn);
    const T X3 = F::x3(RR, PPP, Q);
    const T D = F::subx(Q, X3);
    acc.y = F::y3(Rn, D, acc.y, PPP);
    acc.x = X3;
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
  }

  // 2·(XYZZ) — EFD dbl-2008-s-1 (a = 0).  U = 2Y (norm: <4), V = U², W = U·V, S = X·V, M = 3X² (reduced to <2),
  // X3 = M² − 2S, Y3 = M·(S − X3) − W·Y, ZZ3 = V·ZZ, ZZZ3 = W·ZZZ
  static FF_HD X x_dbl(const X& p)
  {
    if (x_is_zero(p)) return p;
    const T Un = F::norm(F::dbl(p.y));
    const T V = F::sqr_n(Un);
    const T W = F::mul(Un, V);
    const T S = F::mul(p.x, V);
    const T XX = F::sqr_n(p.x);
    const T Mn = F::lt2p(F::tripled(XX)); // 3·XX <6, normalised and reduced: N, <2
    const T MM = F::sqr_n(Mn);
    const T X3 = F::x3(MM, F::zero(), S);
    const T D = F::subx(S, X3);
    return {X3, F::y3(Mn, D, p.y, W), F::mul(V, p.zz), F::mul(W, p.zzz)};
  }

  // XYZZ + XYZZ — EFD add-2008-s
  static FF_HD X x_add(const X& a, const X& b)
  {
    if (x_is_zero(a)) return b;
    if (x_is_zero(b)) return a;
    const T U1 = F::mul(a.x, b.zz);
    const T U2 = F::mul(b.x, a.zz);
    const T S1 = F::mul(a.y, b.zzz);
    const T S2 = F::mul(b.y, a.zzz);
    const T Pn = F::norm(F::sub3(U2, U1));
    const T Rn = F::norm(F::sub3(S2, S1));
    if (F::maybe_zero(Pn) && F::is_zero_full(Pn)) {
      if (F::is_zero_full(Rn)) return x_dbl(a);
      return x_zero();
    }
    const T PP = F::sqr_n(Pn);
    const T PPP = F::mul(Pn, PP);
    const T Q = F::mul(U1, PP);
    const T RR = F::sqr_n(Rn);
    const T X3 = F::x3(RR, PPP, Q);
    const T D = F::subx(Q, X3);
    return {X3, F::y3(Rn, D, S1, PPP), F::mul(F::mul(a.zz, b.zz), PP), F::mul(F::mul(a.zzz, b.zzz), PPP)};
  }

  // bucket arrays between the accumulation and the reduction kernels hold XYZZ in the internal encoding
  // (packed canonical Montgomery-261, same 128 / 256 bytes as ec.h's XYZZ; identity = all zero)
  static FF_HD typename Old::X x_store_internal(const X& p)
  {
    if (x_is_zero(p)) return Old::x_zero();
    return {F::store_internal(p.x), F::store_internal(p.y), F::store_internal(p.zz), F::store_internal(p.zzz)};
  }
  static FF_HD X x_load_internal(const typename Old::X& p) { return {F::load_internal(p.x), F::load_internal(p.y), F::load_internal(p.zz), F::load_internal(p.zzz)}; }
  // ec.h XYZZ (canonical Montgomery-256) → lazy
  static FF_HD X x_from_old(const typename Old::X& p)
  {
    if (Old::x_is_zero(p)) return x_zero();
    return {F::load_mont256(p.x), F::load_mont256(p.y), F::load_mont256(p.zz), F::load_mont256(p.zzz)};
  }

  // lazy affine (N, <16) → packed internal encoding
  static FF_HD typename Old::A store_affine_internal(const A& a) { return {F::store_internal(F::norm(a.x)), F::store_internal(F::norm(a.y))}; }

  // lazy accumulator → packed XYZZ of ec.h (canonical, Montgomery R = 2^256)
  static FF_HD typename Old::X x_store(const X& p)
  {
    if (x_is_zero(p)) return Old::x_zero();
    return {F::store_mont256(p.x), F::store_mont256(p.y), F::store_mont256(p.zz), F::store_mont256(p.zzz)};
  }
};

typedef CurveL<Fq29> G1L;
typedef CurveL<Fq2_29> G2L;

} // namespace bn254
