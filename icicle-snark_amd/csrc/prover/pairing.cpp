// pairing.cpp — BN254 optimal-ate pairing on the host and groth16_verify (the first "next" row).
//
//   bn254_pairing    ← icicle/src/pairing.cpp:11-26  (prepare_q → miller_loop → final_exponentiation,
//                       icicle/include/icicle/pairing/models/bn.h)
//   groth16_verify   ← src/lib.rs:63-82, src/proof_helper.rs:319-372, VerificationKey src/cache.rs:74-108
//
// Host code, like the reference's (the pairing is not on the accelerated path; it is the oracle-independent
// accept/reject check of every proof and completes the CLI's `verify` command).  Written from the published
// algorithm: tower Fq2 = Fq[u]/(u²+1), Fq6 = Fq2[v]/(v³−ξ), Fq12 = Fq6[w]/(w²−v) with ξ = 9+u; optimal ate
// Miller loop over a signed-digit expansion of 6x+2 (x = 4965661367192848881) with D-type twist line
// functions in homogeneous projective coordinates (Costello–Lange–Naehrig), the two Frobenius correction
// steps, final exponentiation with the Fuentes-Castañeda–Knapp–Rodríguez-Henríquez hard part.  All constants
// (Frobenius coefficients, twist constants, digit expansions) are derived at start-up from p, ξ and x.
// The value returned is the canonical representative of e(P,Q) in the same basis as the reference's
// TargetField (12 Fq coefficients, standard form): tests compare it bit for bit with oracle/_ref.
#include <mutex>
#include <random>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/groth16_prover.h"
#include "../common.h"
#include "../ec.h"

using namespace bn254;

namespace {

typedef Fq2Ops F2;
typedef fe2 f2;
struct f6 { f2 c0, c1, c2; };
struct f12 { f6 c0, c1; };

// ---- Fq2 helpers ---------------------------------------------------------------------------------------
f2 f2_conj(const f2& a) { return {a.c0, Fq::neg(a.c1)}; }
f2 f2_mul_fq(const f2& a, const fe& s) { return {Fq::mul(a.c0, s), Fq::mul(a.c1, s)}; }
f2 f2_mul_xi(const f2& a) // × (9 + u)
{
  fe a8 = Fq::dbl(Fq::dbl(Fq::dbl(a.c0))), b8 = Fq::dbl(Fq::dbl(Fq::dbl(a.c1)));
  return {Fq::sub(Fq::add(a8, a.c0), a.c1), Fq::add(Fq::add(b8, a.c1), a.c0)};
}
f2 f2_pow(const f2& a, const uint32_t* e, int nwords)
{
  f2 acc = F2::one(), base = a;
  for (int i = 0; i < nwords * 32; i++) {
    if ((e[i >> 5] >> (i & 31)) & 1) acc = F2::mul(acc, base);
    base = F2::sqr(base);
  }
  return acc;
}

// ---- Fq6 -------------------------------------------------------------------------------------------------
f6 f6_zero() { return {F2::zero(), F2::zero(), F2::zero()}; }
f6 f6_one() { return {F2::one(), F2::zero(), F2::zero()}; }
f6 f6_add(const f6& a, const f6& b) { return {F2::add(a.c0, b.c0), F2::add(a.c1, b.c1), F2::add(a.c2, b.c2)}; }
f6 f6_sub(const f6& a, const f6& b) { return {F2::sub(a.c0, b.c0), F2::sub(a.c1, b.c1), F2::sub(a.c2, b.c2)}; }
f6 f6_neg(const f6& a) { return {F2::neg(a.c0), F2::neg(a.c1), F2::neg(a.c2)}; }
f6 f6_mul(const f6& a, const f6& b)
{
  f2 v0 = F2::mul(a.c0, b.c0), v1 = F2::mul(a.c1, b.c1), v2 = F2::mul(a.c2, b.c2);
  f2 t0 = F2::sub(F2::sub(F2::mul(F2::add(a.c1, a.c2), F2::add(b.c1, b.c2)), v1), v2);
  f2 t1 = F2::sub(F2::sub(F2::mul(F2::add(a.c0, a.c1), F2::add(b.c0, b.c1)), v0), v1);
  f2 t2 = F2::sub(F2::sub(F2::mul(F2::add(a.c0, a.c2), F2::add(b.c0, b.c2)), v0), v2);
  return {F2::add(v0, f2_mul_xi(t0)), F2::add(t1, f2_mul_xi(v2)), F2::add(t2, v1)};
}
f6 f6_mul_v(const f6& a) { return {f2_mul_xi(a.c2), a.c0, a.c1}; } // × v
f6 f6_inv(const f6& a)
{
  f2 c0 = F2::sub(F2::sqr(a.c0), f2_mul_xi(F2::mul(a.c1, a.c2)));
  f2 c1 = F2::sub(f2_mul_xi(F2::sqr(a.c2)), F2::mul(a.c0, a.c1));
  f2 c2 = F2::sub(F2::sqr(a.c1), F2::mul(a.c0, a.c2));
  f2 t = F2::add(f2_mul_xi(F2::add(F2::mul(a.c2, c1), F2::mul(a.c1, c2))), F2::mul(a.c0, c0));
  f2 ti = F2::inv(t);
  return {F2::mul(c0, ti), F2::mul(c1, ti), F2::mul(c2, ti)};
}

// ---- Fq12 ------------------------------------------------------------------------------------------------
f12 f12_one() { return {f6_one(), f6_zero()}; }
f12 f12_mul(const f12& a, const f12& b)
{
  f6 v0 = f6_mul(a.c0, b.c0), v1 = f6_mul(a.c1, b.c1);
  f6 c1 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), v0), v1);
  return {f6_add(v0, f6_mul_v(v1)), c1};
}
f12 f12_sqr(const f12& a) // complex squaring: (a0 + a1 w)² = (a0+a1)(a0+v·a1) − a0a1 − v·a0a1 + 2·a0a1 w
{
  f6 ab = f6_mul(a.c0, a.c1);
  f6 t = f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_v(a.c1)));
  return {f6_sub(f6_sub(t, ab), f6_mul_v(ab)), f6_add(ab, ab)};
}
f12 f12_conj(const f12& a) { return {a.c0, f6_neg(a.c1)}; } // a^(p^6)
f12 f12_inv(const f12& a)
{
  f6 t = f6_sub(f6_mul(a.c0, a.c0), f6_mul_v(f6_mul(a.c1, a.c1)));
  f6 ti = f6_inv(t);
  return {f6_mul(a.c0, ti), f6_neg(f6_mul(a.c1, ti))};
}
bool f12_eq(const f12& a, const f12& b) { return memcmp(&a, &b, sizeof a) == 0; }

// ---- constants derived at start-up -------------------------------------------------------------------------
struct Consts {
  f2 g1[6];   // ξ^(k(p−1)/6), k = 0..5     (Frobenius coefficients for one application)
  f2 g2[6];   // ξ^(k(p²−1)/6)
  f2 g3[6];   // ξ^(k(p³−1)/6)
  fe two_inv; // Montgomery
  f2 b_twist; // 3/ξ  (Montgomery)
  std::vector<int> ate;   // signed digits of 6x+2, little endian
  std::vector<int> znaf;  // signed digits of x
  Consts()
  {
    // (p − 1)/6 by long division of the modulus limbs
    uint32_t e[8];
    uint64_t rem = 0;
    fe pm = Fq::modulus();
    pm.l[0] -= 1;
    for (int i = 7; i >= 0; i--) {
      uint64_t cur = (rem << 32) | pm.l[i];
      e[i] = (uint32_t)(cur / 6);
      rem = cur % 6;
    }
    fe nine = Fq::zero();
    nine.l[0] = 9;
    const f2 xi = {Fq::to_mont(nine), Fq::one_mont()};
    const f2 gamma = f2_pow(xi, e, 8); // ξ^((p−1)/6)
    g1[0] = g2[0] = g3[0] = F2::one();
    // x^p = conj(x) on Fq2:  ξ^((p²−1)/6) = γ·conj(γ),  ξ^((p³−1)/6) = γ²·conj(γ)
    const f2 gamma2 = F2::mul(gamma, f2_conj(gamma)), gamma3 = F2::mul(F2::sqr(gamma), f2_conj(gamma));
    for (int k = 1; k < 6; k++) {
      g1[k] = F2::mul(g1[k - 1], gamma);
      g2[k] = F2::mul(g2[k - 1], gamma2);
      g3[k] = F2::mul(g3[k - 1], gamma3);
    }
    fe two = Fq::zero();
    two.l[0] = 2;
    two_inv = Fq::inv(Fq::to_mont(two));
    fe three = Fq::zero();
    three.l[0] = 3;
    b_twist = F2::mul(f2{Fq::to_mont(three), Fq::zero()}, F2::inv(xi));
    // signed-digit (NAF) expansions
    const unsigned __int128 x = 4965661367192848881ull;
    auto naf = [](unsigned __int128 v) {
      std::vector<int> d;
      while (v) {
        if (v & 1) {
          int z = 2 - (int)(v & 3);
          d.push_back(z);
          v = z > 0 ? v - 1 : v + 1;
        } else d.push_back(0);
        v >>= 1;
      }
      return d;
    };
    ate = naf(6 * x + 2);
    znaf = naf(x);
  }
};
const Consts& K()
{
  static Consts c;
  return c;
}

// Frobenius^k on Fq12 for k = 1, 2, 3
f12 f12_frob(const f12& a, int k)
{
  const f2* g = k == 1 ? K().g1 : k == 2 ? K().g2 : K().g3;
  auto fr = [&](const f2& x) { return (k & 1) ? f2_conj(x) : x; };
  // basis element v^i w^j = ω^(2i + j) with ω⁶ = ξ  →  coefficient of ω^m is multiplied by ξ^(m(p^k −1)/6)
  f12 r;
  r.c0.c0 = fr(a.c0.c0);
  r.c0.c1 = F2::mul(fr(a.c0.c1), g[2]);
  r.c0.c2 = F2::mul(fr(a.c0.c2), g[4]);
  r.c1.c0 = F2::mul(fr(a.c1.c0), g[1]);
  r.c1.c1 = F2::mul(fr(a.c1.c1), g[3]);
  r.c1.c2 = F2::mul(fr(a.c1.c2), g[5]);
  return r;
}

f12 f12_exp_x(const f12& f) // f^x for f in the cyclotomic subgroup (inverse = conjugate)
{
  const std::vector<int>& d = K().znaf;
  f12 res = f12_one(), finv = f12_conj(f);
  bool started = false;
  for (int i = (int)d.size() - 1; i >= 0; i--) {
    if (started) res = f12_sqr(res);
    if (d[i]) {
      started = true;
      res = f12_mul(res, d[i] > 0 ? f : finv);
    }
  }
  return res;
}

// ---- G2 line functions (homogeneous projective R = (X:Y:Z) on the twist), D-type twist ------------------------
struct Line { f2 a, b, c; }; // evaluated at P as  a·y_P + b·x_P·w + c·v·w   (sparse Fq12: slots 0, 3, 4)

Line line_double(f2& X, f2& Y, f2& Z)
{
  const Consts& k = K();
  f2 a = f2_mul_fq(F2::mul(X, Y), k.two_inv);
  f2 b = F2::sqr(Y), c = F2::sqr(Z);
  f2 e = F2::mul(k.b_twist, F2::add(F2::dbl(c), c));
  f2 f = F2::add(F2::dbl(e), e);
  f2 g = f2_mul_fq(F2::add(b, f), k.two_inv);
  f2 h = F2::sub(F2::sqr(F2::add(Y, Z)), F2::add(b, c));
  f2 i = F2::sub(e, b);
  f2 j = F2::sqr(X);
  f2 e2 = F2::sqr(e);
  X = F2::mul(a, F2::sub(b, f));
  Y = F2::sub(F2::sqr(g), F2::add(F2::dbl(e2), e2));
  Z = F2::mul(b, h);
  return {F2::neg(h), F2::add(F2::dbl(j), j), i};
}
Line line_add(f2& X, f2& Y, f2& Z, const f2& qx, const f2& qy)
{
  f2 theta = F2::sub(Y, F2::mul(qy, Z));
  f2 lambda = F2::sub(X, F2::mul(qx, Z));
  f2 c = F2::sqr(theta), d = F2::sqr(lambda);
  f2 e = F2::mul(lambda, d), f = F2::mul(Z, c), g = F2::mul(X, d);
  f2 h = F2::sub(F2::add(e, f), F2::dbl(g));
  X = F2::mul(lambda, h);
  Y = F2::sub(F2::mul(theta, F2::sub(g, h)), F2::mul(e, Y));
  Z = F2::mul(Z, e);
  f2 j = F2::sub(F2::mul(theta, qx), F2::mul(lambda, qy));
  return {lambda, F2::neg(theta), j};
}
// f ← f · (a·y_P + b·x_P·w + c·v·w)
void f12_mul_line(f12& f, const Line& l, const fe& px, const fe& py)
{
  f12 s;
  s.c0 = {f2_mul_fq(l.a, py), F2::zero(), F2::zero()};
  s.c1 = {f2_mul_fq(l.b, px), l.c, F2::zero()};
  f = f12_mul(f, s);
}

f12 pairing_mont(const G1::A& p, const G2::A& q) // inputs Montgomery form, neither is the identity
{
  const Consts& k = K();
  f2 X = q.x, Y = q.y, Z = F2::one();
  const f2 nqy = F2::neg(q.y);
  f12 f = f12_one();
  const std::vector<int>& d = k.ate;
  for (int i = (int)d.size() - 2; i >= 0; i--) {
    f = f12_sqr(f);
    f12_mul_line(f, line_double(X, Y, Z), p.x, p.y);
    if (d[i] == 1) f12_mul_line(f, line_add(X, Y, Z, q.x, q.y), p.x, p.y);
    else if (d[i] == -1) f12_mul_line(f, line_add(X, Y, Z, q.x, nqy), p.x, p.y);
  }
  // Q1 = π(Q), Q2 = −π²(Q)
  const f2 q1x = F2::mul(f2_conj(q.x), k.g1[2]), q1y = F2::mul(f2_conj(q.y), k.g1[3]);
  const f2 q2x = F2::mul(f2_conj(q1x), k.g1[2]), q2y = F2::neg(F2::mul(f2_conj(q1y), k.g1[3]));
  f12_mul_line(f, line_add(X, Y, Z, q1x, q1y), p.x, p.y);
  f12_mul_line(f, line_add(X, Y, Z, q2x, q2y), p.x, p.y);

  // final exponentiation: easy part f^((p⁶−1)(p²+1))
  f12 r = f12_mul(f12_conj(f), f12_inv(f));
  r = f12_mul(f12_frob(r, 2), r);
  // hard part (Fuentes-Castañeda et al.), exponent (p⁴ − p² + 1)/r
  auto expx_neg = [](const f12& a) { return f12_conj(f12_exp_x(a)); }; // a^(−x)
  f12 y0 = expx_neg(r);
  f12 y1 = f12_sqr(y0);
  f12 y2 = f12_sqr(y1);
  f12 y3 = f12_mul(y2, y1);
  f12 y4 = expx_neg(y3);
  f12 y5 = f12_sqr(y4);
  f12 y6 = expx_neg(y5);
  y3 = f12_conj(y3);
  y6 = f12_conj(y6);
  f12 y7 = f12_mul(y6, y4);
  f12 y8 = f12_mul(y7, y3);
  f12 y9 = f12_mul(y8, y1);
  f12 y10 = f12_mul(y8, y4);
  f12 y11 = f12_mul(y10, r);
  f12 y12 = f12_frob(y9, 1);
  f12 y13 = f12_mul(y12, y11);
  f12 y14 = f12_mul(f12_frob(y8, 2), y13);
  f12 y15 = f12_frob(f12_mul(f12_conj(r), y9), 3);
  return f12_mul(y15, y14);
}

void f12_store_std(const f12& v, void* out)
{
  const fe* src = reinterpret_cast<const fe*>(&v);
  fe* dst = reinterpret_cast<fe*>(out);
  for (int i = 0; i < 12; i++) dst[i] = Fq::from_mont(src[i]);
}

// ---- a very small JSON reader (objects, arrays, strings, numbers) ---------------------------------------------
struct JVal {
  enum { NUL, STR, NUM, ARR, OBJ } t = NUL;
  std::string s; // STR / NUM text
  std::vector<JVal> a;
  std::vector<std::pair<std::string, JVal>> o;
  const JVal* get(const char* key) const
  {
    for (auto& kv : o)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};
struct JParser {
  const char* p;
  const char* end;
  bool ok = true;
  int depth = 0; // nesting of the value being parsed; a proof / key nests 3 deep, hostile input must not exhaust the stack
  static constexpr int MAX_DEPTH = 32;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
  std::string str()
  {
    std::string r;
    p++; // opening quote
    while (p < end && *p != '"') {
      const unsigned char ch = (unsigned char)*p;
      // RFC 8259 strings: no raw control characters; the files this verifier reads (snarkjs JSON) are plain ASCII, so bytes
      // above 0x7f are refused as well instead of being checked for well-formed UTF-8
      if (ch < 0x20 || ch > 0x7e) { ok = false; break; }
      if (ch == '\\') {
        if (p + 1 >= end) { ok = false; break; }
        p++;
        const char e = *p;
        if (e == 'u') {
          if (end - p < 5) { ok = false; break; }
          for (int k = 1; k <= 4; k++) {
            const char h = p[k];
            if (!((h >= '0' && h <= '9') || (h >= 'a' && h <= 'f') || (h >= 'A' && h <= 'F'))) ok = false;
          }
          if (!ok) break;
          r.append(p - 1, 6); // kept verbatim: no field this verifier reads contains an escape
          p += 5;
          continue;
        }
        if (e != '"' && e != '\\' && e != '/' && e != 'b' && e != 'f' && e != 'n' && e != 'r' && e != 't') { ok = false; break; }
      }
      r.push_back(*p++);
    }
    if (p < end) p++;
    else ok = false;
    return r;
  }
  JVal val()
  {
    JVal v;
    if (depth >= MAX_DEPTH) { ok = false; return v; }
    struct Nest { int& d; Nest(int& x) : d(x) { d++; } ~Nest() { d--; } } nest(depth);
    ws();
    if (p >= end) { ok = false; return v; }
    if (*p == '"') { v.t = JVal::STR; v.s = str(); }
    else if (*p == '[') {
      v.t = JVal::ARR;
      p++;
      ws();
      if (p < end && *p == ']') { p++; return v; }
      while (ok) {
        v.a.push_back(val());
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == ']') { p++; break; }
        ok = false;
      }
    } else if (*p == '{') {
      v.t = JVal::OBJ;
      p++;
      ws();
      if (p < end && *p == '}') { p++; return v; }
      while (ok) {
        ws();
        if (p >= end || *p != '"') { ok = false; break; }
        std::string k = str();
        ws();
        if (p >= end || *p != ':') { ok = false; break; }
        p++;
        v.o.emplace_back(k, val());
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == '}') { p++; break; }
        ok = false;
      }
    } else {
      v.t = JVal::NUM;
      while (p < end && *p != ',' && *p != ']' && *p != '}' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') v.s.push_back(*p++);
      if (!bare_token_ok(v.s)) ok = false;
    }
    return v;
  }
  // a bare token is a JSON number or one of the three literals — anything else is not JSON (serde_json rejects it too)
  static bool bare_token_ok(const std::string& t)
  {
    if (t == "true" || t == "false" || t == "null") return true;
    size_t i = 0;
    const size_t n = t.size();
    auto digits = [&] { const size_t b = i; while (i < n && t[i] >= '0' && t[i] <= '9') i++; return i > b; };
    if (i < n && t[i] == '-') i++;
    if (!digits()) return false;
    if (i < n && t[i] == '.') { i++; if (!digits()) return false; }
    if (i < n && (t[i] == 'e' || t[i] == 'E')) { i++; if (i < n && (t[i] == '+' || t[i] == '-')) i++; if (!digits()) return false; }
    return i == n;
  }
  // the whole text must be ONE value: only white space may follow it
  JVal document()
  {
    JVal v = val();
    ws();
    if (p != end) ok = false;
    return v;
  }
};

thread_local char g_verr[256] = "";
int vfail(int code, const char* msg)
{
  snprintf(g_verr, sizeof g_verr, "%s", msg);
  return code;
}

bool dec_to_fe(const std::string& s, fe* out) // decimal string → 256-bit standard form (BigUint::parse_bytes)
{
  uint32_t w[8] = {0};
  if (s.empty()) return false;
  for (char ch : s) {
    if (ch < '0' || ch > '9') return false;
    uint64_t carry = (uint64_t)(ch - '0');
    for (int i = 0; i < 8; i++) {
      uint64_t cur = (uint64_t)w[i] * 10 + carry;
      w[i] = (uint32_t)cur;
      carry = cur >> 32;
    }
    if (carry) return false;
  }
  memcpy(out->l, w, 32);
  return true;
}
// The reference deserialises whatever it is given (src/conversions.rs:58-96).  A verifier that takes proofs from an
// untrusted prover has to validate: coordinates must be canonical residues (< q), points must lie on the curve — for G2
// also in the order-r subgroup (the twist has a cofactor) — and public signals must be < r (x and x + r would verify
// alike: the aliasing snarkjs guards against).  (0, 0) stays the identity encoding it is everywhere else in this ABI.
bool g1_valid(const G1::A& a) // Montgomery form
{
  if (G1::aff_is_zero(a)) return true;
  fe three = Fq::zero();
  three.l[0] = 3;
  const fe rhs = Fq::add(Fq::mul(Fq::sqr(a.x), a.x), Fq::to_mont(three));
  return Fq::eq(Fq::sqr(a.y), rhs);
}
bool g2_valid(const G2::A& a) // Montgomery form: on the twist y² = x³ + 3/ξ and killed by r
{
  if (G2::aff_is_zero(a)) return true;
  const f2 rhs = F2::add(F2::mul(F2::sqr(a.x), a.x), K().b_twist);
  if (!F2::eq(F2::sqr(a.y), rhs)) return false;
  const G2::A std_a = {Fq2Ops::from_mont(a.x), Fq2Ops::from_mont(a.y)};
  bn254_g2_projective_t P, Q;
  bn254_g2_from_affine((const bn254_g2_affine_t*)&std_a, &P);
  const fe r = Fr::modulus();
  bn254_g2_mul_scalar(&P, (const bn254_scalar_t*)&r, &Q); // plain windowed double-and-add over the 254 bits of r
  const fe2* z = reinterpret_cast<const fe2*>(&Q) + 2;
  return Fq2Ops::is_zero(*z);
}
bool read_g1(const JVal* v, G1::A* out) // deserialize_g1_affine — src/conversions.rs:58-70
{
  if (!v || v->t != JVal::ARR || v->a.size() < 2) return false;
  for (const JVal& e : v->a)
    if (e.t != JVal::STR) return false; // Vec<String> in the reference: a bare number is a type error there
  fe x, y;
  if (!dec_to_fe(v->a[0].s, &x) || !dec_to_fe(v->a[1].s, &y)) return false;
  if (!Fq::is_canonical(x) || !Fq::is_canonical(y)) return false;
  *out = {Fq::to_mont(x), Fq::to_mont(y)};
  return g1_valid(*out);
}
bool read_g2(const JVal* v, G2::A* out) // deserialize_g2_affine — src/conversions.rs:72-96
{
  if (!v || v->t != JVal::ARR || v->a.size() < 2 || v->a[0].a.size() < 2 || v->a[1].a.size() < 2) return false;
  for (const JVal& row : v->a) {
    if (row.t != JVal::ARR) return false; // Vec<Vec<String>>
    for (const JVal& e : row.a)
      if (e.t != JVal::STR) return false;
  }
  fe c[4];
  if (!dec_to_fe(v->a[0].a[0].s, &c[0]) || !dec_to_fe(v->a[0].a[1].s, &c[1]) || !dec_to_fe(v->a[1].a[0].s, &c[2]) || !dec_to_fe(v->a[1].a[1].s, &c[3])) return false;
  for (const fe& ci : c)
    if (!Fq::is_canonical(ci)) return false;
  *out = {{Fq::to_mont(c[0]), Fq::to_mont(c[1])}, {Fq::to_mont(c[2]), Fq::to_mont(c[3])}};
  return g2_valid(*out);
}
bool read_file(const char* path, std::string* out)
{
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}

void bn254_base_field_generate_scalars_one(bn254_fq_t* out) // unseeded, like utils/rand_gen.h:5
{
  static thread_local std::mt19937_64 gen{std::random_device{}()};
  fe v;
  for (int k = 0; k < 8; k += 2) {
    uint64_t x = gen();
    v.l[k] = (uint32_t)x;
    v.l[k + 1] = (uint32_t)(x >> 32);
  }
  v.l[7] &= 0x3fffffff; // < 2^254 < 6p
  for (int k = 0; k < 5; k++) v = Fq::reduce_once(v);
  memcpy(out, &v, 32);
}

} // namespace

extern "C" {

// bn254_pairing — icicle/src/pairing.cpp:22-26.  p, q affine standard form; out: 12 Fq coefficients (standard form)
// of e(P,Q) in the basis c0.c0.c0, c0.c0.c1, c0.c1.c0, …, c1.c2.c1.  e(O,·) = e(·,O) = 1.
// The reference's C function is void while its Rust binding reads an eIcicleError (wrappers/rust/icicle-core/src/
// pairing/mod.rs:38-43); returning SUCCESS satisfies both.
__attribute__((visibility("default"))) eIcicleError bn254_pairing(const bn254_affine_t* p, const bn254_g2_affine_t* q, bn254_fq12_t* out)
{
  if (!p || !q || !out) return ICICLE_INVALID_POINTER;
  G1::A P;
  G2::A Q;
  memcpy(&P, p, sizeof P);
  memcpy(&Q, q, sizeof Q);
  if (G1::aff_is_zero(P) || G2::aff_is_zero(Q)) {
    f12_store_std(f12_one(), out);
    return ICICLE_SUCCESS;
  }
  f12_store_std(pairing_mont(G1::aff_to_mont(P), G2::aff_to_mont(Q)), out);
  return ICICLE_SUCCESS;
}

// ---- TargetField (Fq12) host FFI: icicle/src/fields/ffi_extern_pairing_extension.cpp:6-52 (standard form in and out) ----
static f12 f12_load_std(const bn254_fq12_t* a)
{
  f12 r;
  const fe* src = reinterpret_cast<const fe*>(a);
  fe* dst = reinterpret_cast<fe*>(&r);
  for (int i = 0; i < 12; i++) dst[i] = Fq::to_mont(src[i]);
  return r;
}
static f12 f12_addsub(const f12& a, const f12& b, bool sub)
{
  f12 r;
  const fe* x = reinterpret_cast<const fe*>(&a);
  const fe* y = reinterpret_cast<const fe*>(&b);
  fe* z = reinterpret_cast<fe*>(&r);
  for (int i = 0; i < 12; i++) z[i] = sub ? Fq::sub(x[i], y[i]) : Fq::add(x[i], y[i]);
  return r;
}
__attribute__((visibility("default"))) void bn254_pairing_target_field_add(const bn254_fq12_t* a, const bn254_fq12_t* b, bn254_fq12_t* r)
{
  f12_store_std(f12_addsub(f12_load_std(a), f12_load_std(b), false), r);
}
__attribute__((visibility("default"))) void bn254_pairing_target_field_sub(const bn254_fq12_t* a, const bn254_fq12_t* b, bn254_fq12_t* r)
{
  f12_store_std(f12_addsub(f12_load_std(a), f12_load_std(b), true), r);
}
__attribute__((visibility("default"))) void bn254_pairing_target_field_mul(const bn254_fq12_t* a, const bn254_fq12_t* b, bn254_fq12_t* r)
{
  f12_store_std(f12_mul(f12_load_std(a), f12_load_std(b)), r);
}
__attribute__((visibility("default"))) void bn254_pairing_target_field_inv(const bn254_fq12_t* a, bn254_fq12_t* r)
{
  f12_store_std(f12_inv(f12_load_std(a)), r);
}
__attribute__((visibility("default"))) void bn254_pairing_target_field_pow(const bn254_fq12_t* base, int exp, bn254_fq12_t* r)
{
  f12 acc = f12_one(), b = f12_load_std(base);
  for (unsigned e = (unsigned)exp; e; e >>= 1) {
    if (e & 1) acc = f12_mul(acc, b);
    b = f12_sqr(b);
  }
  f12_store_std(acc, r);
}
__attribute__((visibility("default"))) void bn254_pairing_target_field_from_u32(uint32_t val, bn254_fq12_t* r)
{
  memset(r, 0, sizeof *r);
  r->c[0][0].c0.limbs[0] = val;
}
__attribute__((visibility("default"))) void bn254_pairing_target_field_generate_scalars(bn254_fq12_t* out, int size)
{
  for (int i = 0; i < size; i++)
    for (int j = 0; j < 12; j++) bn254_base_field_generate_scalars_one(reinterpret_cast<bn254_fq_t*>(&out[i]) + j);
}

__attribute__((visibility("default"))) const char* groth16_verify_last_error(void) { return g_verr; }

// groth16_verify on JSON texts: returns 1 (accepted), 0 (rejected) or a negative error code.
//   e(−A, B) · e(Σ pubᵢ·ICᵢ₊₁ + IC₀, γ₂) · e(C, δ₂) · e(α₁, β₂) = 1        — src/proof_helper.rs:345-369
__attribute__((visibility("default"))) int groth16_verify_json(const char* proof_json, const char* public_json, const char* vk_json)
{
  if (!proof_json || !public_json || !vk_json) return vfail(-3, "null argument");
  JParser pp{proof_json, proof_json + strlen(proof_json)}, pq{public_json, public_json + strlen(public_json)}, pv{vk_json, vk_json + strlen(vk_json)};
  JVal proof = pp.document(), pub = pq.document(), vk = pv.document();
  if (!pp.ok || !pq.ok || !pv.ok || proof.t != JVal::OBJ || pub.t != JVal::ARR || vk.t != JVal::OBJ) return vfail(-2, "malformed JSON");
  G1::A pi_a, pi_c, alpha1;
  G2::A pi_b, beta2, gamma2, delta2;
  (void)K(); // constants (twist coefficient) before the point checks
  if (!read_g1(proof.get("pi_a"), &pi_a) || !read_g2(proof.get("pi_b"), &pi_b) || !read_g1(proof.get("pi_c"), &pi_c))
    return vfail(-2, "proof: bad point (not canonical, not on the curve, or outside the r-torsion)");
  if (!read_g1(vk.get("vk_alpha_1"), &alpha1) || !read_g2(vk.get("vk_beta_2"), &beta2) || !read_g2(vk.get("vk_gamma_2"), &gamma2) || !read_g2(vk.get("vk_delta_2"), &delta2))
    return vfail(-2, "verification key: bad point");
  const JVal* ic = vk.get("IC");
  const JVal* np = vk.get("nPublic");
  if (!ic || ic->t != JVal::ARR || !np) return vfail(-2, "verification key: IC / nPublic missing");
  if ((np->t != JVal::NUM && np->t != JVal::STR) || np->s.empty() || np->s.size() > 9 || np->s.find_first_not_of("0123456789") != std::string::npos)
    return vfail(-2, "verification key: nPublic is not a non-negative integer");
  const size_t n_public = (size_t)strtoul(np->s.c_str(), nullptr, 10);
  if (ic->a.size() < n_public + 1 || pub.a.size() < n_public) return vfail(-2, "public inputs / IC length mismatch");
  for (size_t i = 0; i < n_public; i++)
    if (pub.a[i].t != JVal::STR) return vfail(-2, "public signals must be decimal strings");
  // cpub = IC₀ + Σ pubᵢ·ICᵢ₊₁  (projective host arithmetic through the FFI functions)
  bn254_projective_t cpub, t;
  {
    G1::A a0;
    if (!read_g1(&ic->a[0], &a0)) return vfail(-2, "IC: bad point");
    G1::A s = {Fq::from_mont(a0.x), Fq::from_mont(a0.y)};
    bn254_from_affine((const bn254_affine_t*)&s, &cpub);
  }
  for (size_t i = 0; i < n_public; i++) {
    G1::A ai;
    fe sc;
    if (!read_g1(&ic->a[i + 1], &ai) || !dec_to_fe(pub.a[i].s, &sc)) return vfail(-2, "IC / public: bad value");
    if (!Fr::is_canonical(sc)) return vfail(-2, "public signal is not below the scalar field modulus");
    G1::A s = {Fq::from_mont(ai.x), Fq::from_mont(ai.y)};
    bn254_projective_t pi;
    bn254_from_affine((const bn254_affine_t*)&s, &pi);
    bn254_mul_scalar(&pi, (const bn254_scalar_t*)&sc, &t);
    bn254_ecadd(&cpub, &t, &cpub);
  }
  bn254_affine_t cpub_aff;
  bn254_to_affine(&cpub, &cpub_aff);
  G1::A cp;
  memcpy(&cp, &cpub_aff, sizeof cp);
  cp = G1::aff_to_mont(cp);
  const G1::A neg_a = {pi_a.x, Fq::neg(pi_a.y)};
  // four pairings on four threads, like the reference (src/proof_helper.rs:351-362)
  f12 e[4];
  auto one_or = [](const G1::A& P, const G2::A& Q) { return (G1::aff_is_zero(P) || G2::aff_is_zero(Q)) ? f12_one() : pairing_mont(P, Q); };
  (void)K(); // constants before the threads start
  std::thread t1([&] { e[0] = one_or(neg_a, pi_b); });
  std::thread t2([&] { e[1] = one_or(cp, gamma2); });
  std::thread t3([&] { e[2] = one_or(pi_c, delta2); });
  e[3] = one_or(alpha1, beta2);
  t1.join(); t2.join(); t3.join();
  f12 prod = f12_mul(f12_mul(e[0], e[1]), f12_mul(e[2], e[3]));
  return f12_eq(prod, f12_one()) ? 1 : 0;
}

// groth16_verify — src/lib.rs:63-82 (files in).  Returns 0 when the proof is accepted (the reference asserts),
// 1 when it is rejected, negative on I/O / format errors.
__attribute__((visibility("default"))) int groth16_verify(const char* proof_path, const char* public_path, const char* vk_path)
{
  std::string a, b, c;
  if (!proof_path || !public_path || !vk_path) return vfail(-3, "null argument");
  if (!read_file(proof_path, &a) || !read_file(public_path, &b) || !read_file(vk_path, &c)) return vfail(-1, "cannot read input file");
  const int r = groth16_verify_json(a.c_str(), b.c_str(), c.c_str());
  if (r < 0) return r;
  if (r == 0) return vfail(1, "Verification failed");
  return 0;
}

} // extern "C"
