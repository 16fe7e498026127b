// host check of csrc/fr29.h against the 8×32-bit Fr of ff.h:  g++ -O2 -std=c++17 -Iicicle-snark_amd/csrc -o /tmp/fr29_check tests/fr29_check.cc  (tests/test_fr29.py does that)
#include <cstdio>
#include <cstring>
#include <random>
#include "fr29.h"
using namespace bn254;
static std::mt19937_64 rng(7);
static fe rand_fe()
{
  fe a;
  for (int i = 0; i < 8; i++) a.l[i] = (uint32_t)rng();
  a.l[7] &= 0x0fffffffu; // < 2^252 < r
  return a;
}
static bool eq(const fe& a, const fe& b) { return memcmp(&a, &b, sizeof a) == 0; }
int main()
{
  int bad = 0;
  for (int it = 0; it < 200000; it++) {
    const fe x = rand_fe(), y = rand_fe(), w = rand_fe();      // standard-form values
    const fe w256 = Fr::to_mont(w);
    const fe9 w261 = fr29::canon4(fr29::mul(fr29::unpack(w256), fr29::c256_to_261()));
    // x·w
    const fe want = Fr::mul(x, w256);
    const fe got = fr29::pack(fr29::canon(fr29::mul(fr29::unpack(x), w261)));
    if (!eq(want, got)) bad++;
    // lazy chain: ((x + y)·K-form, x − y + K r) through several levels, then multiply and canonicalise
    fe9 a = fr29::unpack(x), b = fr29::unpack(y);
    fe9 s = fr29::norm(fr29::add(a, b));             // < 2
    fe9 d = fr29::norm(fr29::sub<2>(a, b));          // < 3
    fe9 s2 = fr29::norm(fr29::add(s, d));            // < 5   = 2x + 2r
    fe9 d2 = fr29::norm(fr29::sub<4>(s, d));         // s + 4r − d = 2y + 2r  < 6
    fe9 s3 = fr29::norm(fr29::add(s2, d2));          // 2x + 2y + 4r
    for (int k = 0; k < 6; k++) s3 = fr29::norm(fr29::add(s3, s3)); // ×64: < 704
    const fe want_s3 = Fr::mul(Fr::add(Fr::add(x, y), Fr::add(x, y)), Fr::to_mont([] { fe t = Fr::zero(); t.l[0] = 64; return t; }()));
    if (!eq(want_s3, fr29::pack(fr29::canon(s3)))) bad++;
    const fe9 sh = fr29::shrink(s3);
    if (!eq(want_s3, fr29::pack(fr29::canon(sh)))) bad++;
    // product of a big lazy value with a twiddle
    const fe want_p = Fr::mul(want_s3, w256);
    if (!eq(want_p, fr29::pack(fr29::canon(fr29::mul(s3, w261))))) bad++;
    // subtraction against a big subtrahend
    const fe9 dd = fr29::norm(fr29::sub<706>(a, s3));
    if (!eq(Fr::sub(x, want_s3), fr29::pack(fr29::canon(dd)))) bad++;
    // standard-form product of two standard-form values
    const fe want_xy = Fr::mul(Fr::to_mont(x), y);
    if (!eq(want_xy, fr29::pack(fr29::canon(fr29::mul(fr29::mul(fr29::unpack(x), fr29::unpack(y)), fr29::r2()))))) bad++;
  }
  printf("fr29 check: %d mismatches\n", bad);
  return bad != 0;
}
