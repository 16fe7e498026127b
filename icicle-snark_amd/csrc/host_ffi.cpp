// host_ffi.cpp — host-side (synchronous, CPU) field and curve FFI.
//
// In the reference these are host functions too (icicle/src/fields/ffi_extern.cpp,
// icicle/src/curves/ffi_extern.cpp): the Rust prover calls them a handful of times per proof
// for the r/s blinding and the final affine conversion (src/proof_helper.rs:274-316).  They are
// part of the drop-in boundary, not of the accelerated path.  Arithmetic: the same ff.h / ec.h
// templates the kernels use, Montgomery form inside, standard form at the ABI; projective
// results use the reference's complete formulas so the returned representative is identical.
#include <random>
#include <string.h>

#include "common.h"
#include "ec.h"

using namespace bn254;

namespace {

inline fe ld(const bn254_scalar_t* p)
{
  fe r;
  memcpy(r.l, p->limbs, 32);
  return r;
}
inline fe ldq(const bn254_fq_t* p)
{
  fe r;
  memcpy(r.l, p->limbs, 32);
  return r;
}
inline void st(bn254_scalar_t* p, const fe& v) { memcpy(p->limbs, v.l, 32); }
inline void stq(bn254_fq_t* p, const fe& v) { memcpy(p->limbs, v.l, 32); }

// Weierstrass 3·b in Montgomery form: G1 b = 3; G2 b' = 3/(9+u)
// (icicle/include/icicle/curves/params/bn254.h:25-27,40-43)
struct B3 {
  fe g1;
  fe2 g2;
  B3()
  {
    fe nine = Fq::zero();
    nine.l[0] = 9;
    g1 = Fq::to_mont(nine);
    fe2 xi = {Fq::to_mont(nine), Fq::one_mont()};
    fe three = Fq::zero();
    three.l[0] = 3;
    fe2 t3 = {Fq::to_mont(three), Fq::zero()};
    fe2 b = Fq2Ops::mul(t3, Fq2Ops::inv(xi));
    g2 = Fq2Ops::mul(t3, b);
  }
};
const B3& b3()
{
  static B3 v;
  return v;
}
template <class C>
struct B3Of;
template <>
struct B3Of<G1> {
  static const fe& get() { return b3().g1; }
};
template <>
struct B3Of<G2> {
  static const fe2& get() { return b3().g2; }
};

// scalar · point with the reference's fixed 4-bit window schedule (projective.h:176-208)
template <class C>
typename C::P mul_scalar(const typename C::P& point, const fe& scalar)
{
  const auto& B = B3Of<C>::get();
  typename C::P table[15];
  table[0] = point;
  for (int i = 1; i < 15; i++) table[i] = C::p_add(table[i - 1], point, B);
  typename C::P res = C::p_zero();
  bool nz = false;
  for (int w = 63; w >= 0; w--) {
    unsigned d = (scalar.l[w >> 3] >> ((w & 7) * 4)) & 15;
    for (int j = 0; nz && j < 4; j++) res = C::p_dbl(res, B);
    if (d) {
      res = C::p_add(res, table[d - 1], B);
      nz = true;
    }
  }
  return res;
}

template <class C, class PT>
typename C::P load_p(const PT* p)
{
  typename C::P r;
  static_assert(sizeof(r) == sizeof(PT), "layout");
  memcpy(&r, p, sizeof r);
  return C::p_to_mont(r);
}
template <class C, class PT>
void store_p(PT* p, const typename C::P& v)
{
  typename C::P r = C::p_from_mont(v);
  memcpy(p, &r, sizeof r);
}

template <class C, class PT>
bool is_raw_zero3(const PT* p)
{
  const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
  uint32_t x = 0;
  for (size_t i = 0; i < sizeof(PT) / 4; i++) x |= w[i];
  return x == 0;
}

// ffi_extern.cpp:9-16 — equality that rejects the all-zero (invalid) triple
template <class C, class PT>
bool ffi_eq(const PT* a, const PT* b)
{
  if (is_raw_zero3<C>(a) || is_raw_zero3<C>(b)) return false;
  return C::p_eq(load_p<C>(a), load_p<C>(b));
}

} // namespace

// ------------------------------------------------------------------------------------------------ Fr
ISNARK_API void bn254_add(const bn254_scalar_t* a, const bn254_scalar_t* b, bn254_scalar_t* r) { st(r, Fr::add(ld(a), ld(b))); }
ISNARK_API void bn254_sub(const bn254_scalar_t* a, const bn254_scalar_t* b, bn254_scalar_t* r) { st(r, Fr::sub(ld(a), ld(b))); }
ISNARK_API void bn254_mul(const bn254_scalar_t* a, const bn254_scalar_t* b, bn254_scalar_t* r)
{
  // montmul(a, b·R) = a·b
  st(r, Fr::mul(ld(a), Fr::to_mont(ld(b))));
}
ISNARK_API void bn254_inv(const bn254_scalar_t* a, bn254_scalar_t* r)
{
  st(r, Fr::from_mont(Fr::inv(Fr::to_mont(ld(a)))));
}
ISNARK_API void bn254_pow(const bn254_scalar_t* base, int exp, bn254_scalar_t* r)
{
  fe e = Fr::zero();
  e.l[0] = (uint32_t)exp;
  st(r, Fr::from_mont(Fr::pow(Fr::to_mont(ld(base)), e)));
}
ISNARK_API void bn254_from_u32(uint32_t val, bn254_scalar_t* r)
{
  fe v = Fr::zero();
  v.l[0] = val;
  st(r, v);
}
// unseeded randomness like the reference (icicle/include/icicle/utils/rand_gen.h:5)
ISNARK_API void bn254_generate_scalars(bn254_scalar_t* scalars, int size)
{
  static thread_local std::mt19937_64 gen{std::random_device{}()};
  for (int i = 0; i < size; i++) {
    fe v;
    for (int k = 0; k < 8; k += 2) {
      uint64_t x = gen();
      v.l[k] = (uint32_t)x;
      v.l[k + 1] = (uint32_t)(x >> 32);
    }
    v.l[7] &= 0x3fffffff; // < 2^254
    v = Fr::reduce_once(v);
    v = Fr::reduce_once(v); // 2^254 < 2r·… : two conditional subtractions bring it below r
    st(&scalars[i], v);
  }
}
ISNARK_API void bn254_base_field_from_u32(uint32_t val, bn254_fq_t* r)
{
  fe v = Fq::zero();
  v.l[0] = val;
  stq(r, v);
}
ISNARK_API void bn254_g2_base_field_from_u32(uint32_t val, bn254_fq2_t* r)
{
  memset(r, 0, sizeof *r);
  r->c0.limbs[0] = val;
}

// ------------------------------------------------------------------------------------------------ G1
ISNARK_API bool bn254_eq(const bn254_projective_t* a, const bn254_projective_t* b) { return ffi_eq<G1>(a, b); }
ISNARK_API void bn254_ecadd(const bn254_projective_t* a, const bn254_projective_t* b, bn254_projective_t* r)
{
  store_p<G1>(r, G1::p_add(load_p<G1>(a), load_p<G1>(b), b3().g1));
}
ISNARK_API void bn254_ecsub(const bn254_projective_t* a, const bn254_projective_t* b, bn254_projective_t* r)
{
  store_p<G1>(r, G1::p_add(load_p<G1>(a), G1::p_neg(load_p<G1>(b)), b3().g1));
}
ISNARK_API void bn254_mul_scalar(const bn254_projective_t* p, const bn254_scalar_t* s, bn254_projective_t* r)
{
  store_p<G1>(r, mul_scalar<G1>(load_p<G1>(p), ld(s)));
}
ISNARK_API void bn254_to_affine(const bn254_projective_t* p, bn254_affine_t* out)
{
  G1::A a = G1::p_to_affine(load_p<G1>(p));
  a = {Fq::from_mont(a.x), Fq::from_mont(a.y)};
  memcpy(out, &a, sizeof a);
}
ISNARK_API void bn254_from_affine(const bn254_affine_t* p, bn254_projective_t* out)
{
  G1::A a;
  memcpy(&a, p, sizeof a);
  G1::P r = G1::aff_is_zero(a) ? G1::P{Fq::zero(), Fq::one_std(), Fq::zero()} : G1::P{a.x, a.y, Fq::one_std()};
  memcpy(out, &r, sizeof r);
}
ISNARK_API void bn254_generator(bn254_projective_t* out)
{
  memset(out, 0, sizeof *out);
  out->x.limbs[0] = 1;
  out->y.limbs[0] = 2;
  out->z.limbs[0] = 1;
}
ISNARK_API bool bn254_is_on_curve(const bn254_projective_t* pp)
{
  G1::P p = load_p<G1>(pp);
  if (Fq::is_zero(p.x) && !Fq::is_zero(p.y) && Fq::is_zero(p.z)) return true;
  if (Fq::is_zero(p.z)) return false;
  // 3·(Z·Y² − X³) == b3·Z³
  fe z3 = Fq::mul(Fq::sqr(p.z), p.z), x3 = Fq::mul(Fq::sqr(p.x), p.x), zy2 = Fq::mul(Fq::sqr(p.y), p.z);
  return Fq::eq(Fq::mul3(Fq::sub(zy2, x3)), Fq::mul(b3().g1, z3));
}

// ------------------------------------------------------------------------------------------------ G2
ISNARK_API bool bn254_g2_eq(const bn254_g2_projective_t* a, const bn254_g2_projective_t* b) { return ffi_eq<G2>(a, b); }
ISNARK_API void bn254_g2_ecadd(const bn254_g2_projective_t* a, const bn254_g2_projective_t* b, bn254_g2_projective_t* r)
{
  store_p<G2>(r, G2::p_add(load_p<G2>(a), load_p<G2>(b), b3().g2));
}
ISNARK_API void bn254_g2_ecsub(const bn254_g2_projective_t* a, const bn254_g2_projective_t* b, bn254_g2_projective_t* r)
{
  store_p<G2>(r, G2::p_add(load_p<G2>(a), G2::p_neg(load_p<G2>(b)), b3().g2));
}
ISNARK_API void bn254_g2_mul_scalar(const bn254_g2_projective_t* p, const bn254_scalar_t* s, bn254_g2_projective_t* r)
{
  store_p<G2>(r, mul_scalar<G2>(load_p<G2>(p), ld(s)));
}
ISNARK_API void bn254_g2_to_affine(const bn254_g2_projective_t* p, bn254_g2_affine_t* out)
{
  G2::A a = G2::p_to_affine(load_p<G2>(p));
  a = {Fq2Ops::from_mont(a.x), Fq2Ops::from_mont(a.y)};
  memcpy(out, &a, sizeof a);
}
ISNARK_API void bn254_g2_from_affine(const bn254_g2_affine_t* p, bn254_g2_projective_t* out)
{
  G2::A a;
  memcpy(&a, p, sizeof a);
  fe2 one = {Fq::one_std(), Fq::zero()}, zero = Fq2Ops::zero();
  G2::P r = G2::aff_is_zero(a) ? G2::P{zero, one, zero} : G2::P{a.x, a.y, one};
  memcpy(out, &r, sizeof r);
}
ISNARK_API void bn254_g2_generator(bn254_g2_projective_t* out)
{
  // icicle/include/icicle/curves/params/bn254.h:32-39
  static const uint32_t xr[8] = {0xd992f6ed, 0x46debd5c, 0xf75edadd, 0x674322d4, 0x5e5c4479, 0x426a0066, 0x121f1e76, 0x1800deef};
  static const uint32_t xi[8] = {0xaef312c2, 0x97e485b7, 0x35a9e712, 0xf1aa4933, 0x31fb5d25, 0x7260bfb7, 0x920d483a, 0x198e9393};
  static const uint32_t yr[8] = {0x66fa7daa, 0x4ce6cc01, 0x0c43d37b, 0xe3d1e769, 0x8dcb408f, 0x4aab7180, 0xdb8c6deb, 0x12c85ea5};
  static const uint32_t yi[8] = {0xd122975b, 0x55acdadc, 0x70b38ef3, 0xbc4b3133, 0x690c3395, 0xec9e99ad, 0x585ff075, 0x090689d0};
  memset(out, 0, sizeof *out);
  memcpy(out->x.c0.limbs, xr, 32);
  memcpy(out->x.c1.limbs, xi, 32);
  memcpy(out->y.c0.limbs, yr, 32);
  memcpy(out->y.c1.limbs, yi, 32);
  out->z.c0.limbs[0] = 1;
}
ISNARK_API bool bn254_g2_is_on_curve(const bn254_g2_projective_t* pp)
{
  typedef Fq2Ops F;
  G2::P p = load_p<G2>(pp);
  if (F::is_zero(p.x) && !F::is_zero(p.y) && F::is_zero(p.z)) return true;
  if (F::is_zero(p.z)) return false;
  fe2 z3 = F::mul(F::sqr(p.z), p.z), x3 = F::mul(F::sqr(p.x), p.x), zy2 = F::mul(F::sqr(p.y), p.z);
  fe2 d = F::sub(zy2, x3);
  return F::eq(F::add(F::dbl(d), d), F::mul(b3().g2, z3));
}
