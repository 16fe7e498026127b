// f29_check.cpp — host-side checked build of the lazy radix-2^29 arithmetic (csrc/ff29.h, csrc/ec29.h).
// Test infrastructure: compiled with g++ -DF29_CHECK by tests/test_f29.py; every bound stated in those headers is
// asserted (128-bit column accumulators, limb domination in subtractions, value bounds) while the results are
// compared with the 8×32-bit arithmetic of ff.h / ec.h.  Returns 0 on success; f29_last_failure() names the
// first violated bound.
#include <stdint.h>
#include <string.h>

#include "../icicle-snark_amd/csrc/ec29.h"

using namespace bn254;

static const char* g_msg = "";
extern "C" const char* f29_last_failure() { return f29::g_check_failure ? f29::g_check_failure : g_msg; }
extern "C" void f29_reset() { f29::g_check_failure = nullptr; g_msg = ""; }

static bool same(const fe& a, const fe& b) { return memcmp(&a, &b, sizeof a) == 0; }
static int bad(const char* m) { g_msg = m; return 1; }

// field level: n triples (a, b, c) of canonical standard-form Fq elements
extern "C" int f29_check_field(const fe* v, int n)
{
  for (int i = 0; i + 3 < n; i++) {
    const fe a = v[i], b = v[i + 1], c = v[i + 2], d = v[i + 3];
    const fe am = Fq::to_mont(a), bm = Fq::to_mont(b), cm = Fq::to_mont(c), dm = Fq::to_mont(d);
    const fe9 a9 = f29::from_std(a), b9 = f29::from_std(b), c9 = f29::from_std(c), d9 = f29::from_std(d);
    if (!same(f29::to_mont256(a9), am)) return bad("from_std/to_mont256 round trip");
    if (!same(f29::to_mont256(f29::from_mont256(am)), am)) return bad("from_mont256");
    if (!same(f29::pack(f29::unpack(a)), a)) return bad("pack/unpack");
    if (!same(f29::to_mont256(f29::mul(a9, b9)), Fq::mul(am, bm))) return bad("mul");
    if (!same(f29::to_mont256(f29::sqr(a9)), Fq::sqr(am))) return bad("sqr");
    if (!same(f29::to_mont256(f29::mul2(a9, b9, c9, d9)), Fq::add(Fq::mul(am, bm), Fq::mul(cm, dm)))) return bad("mul2");
    if (!same(f29::to_mont256(f29::mul4(a9, b9, c9, d9, a9, c9, b9, d9)),
              Fq::add(Fq::add(Fq::mul(am, bm), Fq::mul(cm, dm)), Fq::add(Fq::mul(am, cm), Fq::mul(bm, dm)))))
      return bad("mul4");
    if (!same(f29::to_mont256(f29::norm(f29::sub<3, 1>(a9, b9))), Fq::sub(am, bm))) return bad("sub<3,1>");
    if (!same(f29::to_mont256(f29::norm(f29::add(a9, b9))), Fq::add(am, bm))) return bad("add");
    const fe9 t = f29::norm(f29::sub<5, 3>(a9, f29::add(b9, f29::dbl(c9))));
    if (!same(f29::to_mont256(f29::reduce_lt2p(t)), Fq::sub(Fq::sub(am, bm), Fq::dbl(cm)))) return bad("sub<5,3>/reduce_lt2p");
    if (!same(f29::pack(f29::canon(f29::unpack(a))), a)) return bad("canon of canonical");
    // lazy squares at the stated input bound (<5p, N)
    const fe9 w = f29::norm(f29::sub<3, 1>(a9, b9));
    if (!same(f29::to_mont256(f29::sqr(w)), Fq::sqr(Fq::sub(am, bm)))) return bad("sqr of a lazy difference");
    if (i % 64 == 0 && !same(f29::to_mont256(f29::inv(a9)), Fq::inv(am))) return bad("inv");
    if (!same(f29::to_mont256(f29::inv_ds(a9)), Fq::inv(am))) return bad("inv_ds");
    if (!same(f29::to_mont256(f29::inv_ds(f29::norm(f29::add(a9, f29::mul(b9, f29::one_m()))))), Fq::inv(Fq::add(am, bm)))) return bad("inv_ds of a lazy sum");
    if (!same(f29::pack(f29::unpack(f29::pack(f29::mul(a9, b9)))), f29::pack(f29::mul(a9, b9)))) return bad("pack of a lazy (<2p) value");
    const bool z = f29::maybe_zero_mod_p(f29::norm(f29::sub<3, 1>(a9, a9)));
    if (!z || !f29::is_zero_canon(f29::canon(f29::norm(f29::sub<3, 1>(a9, a9))))) return bad("zero test");
    if (f29::g_check_failure) return 2;
  }
  return 0;
}

// curve level: accumulate n affine points (Montgomery-256 packed, signs in bit 0 of sg[i]) with both arithmetics
template <class CL>
static int chain(const typename CL::Old::A* pts, const uint8_t* sg, int n, int form, const typename CL::Old::A* pts_form)
{
  typedef typename CL::Old Old;
  typename Old::X ref = Old::x_zero();
  typename CL::X acc = CL::x_zero();
  for (int i = 0; i < n; i++) {
    typename Old::A p = pts[i];
    if (sg[i] & 1) p = Old::aff_neg(p);
    Old::x_madd(ref, p);
    CL::x_madd(acc, CL::load_affine(pts_form[i], form, sg[i] & 1));
    if (f29::g_check_failure) return 2;
    // compare as group elements AND as coordinates (same formulas ⇒ same field values)
    const typename Old::X got = CL::x_store(acc);
    if (memcmp(&got, &ref, sizeof got) != 0) {
      // identical points can differ in representation only after a doubling/cancellation branch; compare projectively
      typename Old::P a = Old::x_to_projective(got), b = Old::x_to_projective(ref);
      if (!Old::p_eq(a, b)) return bad("madd chain diverged from ec.h");
    }
  }
  return 0;
}
// reduction-style use: lazy x_add / x_dbl over partial sums built from the points, against ec.h
template <class CL>
static int addtree(const typename CL::Old::A* pts, int n)
{
  typedef typename CL::Old Old;
  if (n < 4) return 0;
  // partial sums s_i = P_0 + … + P_i in both arithmetics
  typename Old::X ro = Old::x_zero();
  typename CL::X rl = CL::x_zero();
  typename Old::X line_o = Old::x_zero(), tri_o = Old::x_zero();
  typename CL::X line_l = CL::x_zero(), tri_l = CL::x_zero();
  for (int i = 0; i < n; i++) {
    Old::x_madd(ro, pts[i]);
    CL::x_madd(rl, CL::load_affine(pts[i], 1, false));
    // running sums like msm_bucket_reduce_kernel, through the internal encoding like the bucket array
    const typename CL::X b = CL::x_load_internal(CL::x_store_internal(rl));
    line_o = Old::x_add(line_o, ro);
    tri_o = Old::x_add(tri_o, line_o);
    line_l = CL::x_add(line_l, b);
    tri_l = CL::x_add(tri_l, line_l);
    if (i % 5 == 4) { // doublings, incl. x_add(a, a) → doubling branch and a + (−a) → identity
      tri_o = Old::x_dbl(tri_o);
      tri_l = CL::x_dbl(tri_l);
      line_o = Old::x_add(line_o, line_o);
      line_l = CL::x_add(line_l, line_l);
      const typename Old::X z = Old::x_add(ro, Old::x_neg(ro));
      typename CL::X nl = CL::x_from_old(Old::x_neg(ro));
      const typename CL::X zl = CL::x_add(rl, nl);
      if (!Old::x_is_zero(z) || !CL::x_is_zero(zl)) return bad("a + (−a) is not the identity");
    }
    if (f29::g_check_failure) return 2;
    const typename Old::X got = CL::x_store(tri_l);
    if (!Old::p_eq(Old::x_to_projective(got), Old::x_to_projective(tri_o))) return bad("x_add/x_dbl diverged from ec.h");
    const typename Old::X got2 = CL::x_store(CL::x_from_old(tri_o));
    if (memcmp(&got2, &tri_o, sizeof got2) != 0) return bad("x_from_old/x_store round trip");
  }
  return 0;
}
extern "C" int f29_check_g1_addtree(const void* pts, int n) { return addtree<G1L>((const G1::A*)pts, n); }
extern "C" int f29_check_g2_addtree(const void* pts, int n) { return addtree<G2L>((const G2::A*)pts, n); }

// pts: Montgomery-256 packed affine; pts_form: the same points in the encoding `form` (0 std, 1 mont256, 2 internal)
extern "C" int f29_check_g1_chain(const void* pts, const uint8_t* sg, int n, int form, const void* pts_form)
{
  return chain<G1L>((const G1::A*)pts, sg, n, form, (const G1::A*)pts_form);
}
extern "C" int f29_check_g2_chain(const void* pts, const uint8_t* sg, int n, int form, const void* pts_form)
{
  return chain<G2L>((const G2::A*)pts, sg, n, form, (const G2::A*)pts_form);
}
// encode canonical Montgomery-256 elements into the internal form (packed canonical Montgomery-261)
extern "C" void f29_to_internal(const fe* in, fe* out, int n)
{
  for (int i = 0; i < n; i++) out[i] = f29::pack(f29::canon(f29::from_mont256(in[i])));
}
