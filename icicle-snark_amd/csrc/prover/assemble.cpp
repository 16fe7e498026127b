// assemble.cpp — tail of groth16_prove_helper (src/proof_helper.rs:274-316): blinding with (r, s), affine conversion and the
// JSON texts (src/conversions.rs:30-56, src/file_wrapper.rs:105-113), plus the group sum of shard commitments.
#include <algorithm>
#include <fcntl.h>
#include <errno.h>
#include <sys/mman.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <unistd.h>

#include "prover_internal.h"

using namespace bn254;
using namespace isnark;
using namespace isnark::prover;

namespace isnark {
namespace prover {
namespace {
std::string to_decimal(const fe& v) // BigUint::to_str_radix(10) — src/conversions.rs:30-40
{
  uint32_t w[8];
  memcpy(w, v.l, 32);
  std::string out;
  bool nz = true;
  while (nz) {
    uint64_t rem = 0;
    nz = false;
    for (int i = 7; i >= 0; i--) {
      uint64_t cur = (rem << 32) | w[i];
      w[i] = (uint32_t)(cur / 1000000000u);
      rem = cur % 1000000000u;
      if (w[i]) nz = true;
    }
    char buf[16];
    snprintf(buf, sizeof buf, nz ? "%09u" : "%u", (unsigned)rem);
    out.insert(0, buf);
  }
  return out;
}

// Uniform scalar in [0, r) from the kernel's CSPRNG (getrandom(2), /dev/urandom as fallback): 254 random bits, rejected
// while ≥ r (acceptance 0.756).  The reference draws r, s from an unseeded mt19937 (ScalarCfg::generate_random →
// utils/rand_gen.h:5) — zero-knowledge must not rest on a 32-bit-seeded, predictable generator, so the production path
// does not restate that; bn254_generate_scalars (test data, like the reference's) keeps the Mersenne twister.
bool secure_scalar(bn254_scalar_t* out)
{
  for (int tries = 0; tries < 256; tries++) {
    fe v;
    size_t got = 0;
    while (got < 32) {
      const ssize_t k = getrandom((uint8_t*)v.l + got, 32 - got, 0);
      if (k < 0) {
        if (errno == EINTR) continue;
        break;
      }
      got += (size_t)k;
    }
    if (got < 32) {
      FILE* f = fopen("/dev/urandom", "rb");
      if (!f) return false;
      const size_t k = fread(v.l, 1, 32, f);
      fclose(f);
      if (k != 32) return false;
    }
    v.l[7] &= 0x3fffffffu;
    if (Fr::is_canonical(v)) {
      memcpy(out, v.l, 32);
      return true;
    }
  }
  return false;
}
} // namespace

int compute_blinding(const ZKeyCache* z, const uint8_t* r_in, const uint8_t* s_in, Blinding* b)
{
  bn254_scalar_t rs[2];
  if ((!r_in && !secure_scalar(&rs[0])) || (!s_in && !secure_scalar(&rs[1]))) return fail(ERR_IO, "no entropy source for the blinding scalars"); // src/proof_helper.rs:276
  if (r_in) memcpy(&rs[0], r_in, 32);
  if (s_in) memcpy(&rs[1], s_in, 32);
  b->r = rs[0];
  b->s = rs[1];
  const bn254_projective_t* delta1 = (const bn254_projective_t*)&z->vk_delta_1;
  const bn254_g2_projective_t* delta2 = (const bn254_g2_projective_t*)&z->vk_delta_2;
  bn254_mul_scalar(delta1, &b->r, &b->d1r);
  bn254_mul_scalar(delta1, &b->s, &b->d1s);
  bn254_mul_scalar(&b->d1r, &b->s, &b->d1rs);
  bn254_g2_mul_scalar(delta2, &b->s, &b->d2s);
  return 0;
}

static std::string dec32(const void* p)
{
  fe v;
  memcpy(v.l, p, 32);
  return to_decimal(v);
}
void early_pi_a(const ZKeyCache* z, const Blinding& bl, const bn254_projective_t* a_plus_alpha_d1r, EarlyTerms* et)
{
  (void)z;
  (void)bl;
  bn254_affine_t a_aff;
  bn254_to_affine(a_plus_alpha_d1r, &a_aff);
  et->a_dec[0] = dec32(&a_aff.x);
  et->a_dec[1] = dec32(&a_aff.y);
  et->a_ready.store(true, std::memory_order_release);
}
void early_pi_b(const ZKeyCache* z, const Blinding& bl, const bn254_g2_projective_t* b2_sum, EarlyTerms* et)
{
  bn254_g2_projective_t pi_b = *b2_sum;
  bn254_g2_ecadd(&pi_b, (const bn254_g2_projective_t*)&z->vk_beta_2, &pi_b);
  bn254_g2_ecadd(&pi_b, &bl.d2s, &pi_b); // pi_b = B2 + β2 + δ2·s — src/proof_helper.rs:281
  bn254_g2_affine_t b_aff;
  bn254_g2_to_affine(&pi_b, &b_aff);
  et->b_dec[0] = dec32(&b_aff.x.c0);
  et->b_dec[1] = dec32(&b_aff.x.c1);
  et->b_dec[2] = dec32(&b_aff.y.c0);
  et->b_dec[3] = dec32(&b_aff.y.c1);
  et->b_ready.store(true, std::memory_order_release);
}

int assemble_impl(const ZKeyCache* z, const void* wtns, size_t wtns_len, const uint8_t* points, const Blinding& bl, char* proof_json, size_t proof_cap, char* public_json, size_t public_cap, const EarlyTerms* et)
{
  Wtns w;
  if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
  if (w.n_witness != z->n_vars) return fail(ERR_FORMAT, "Invalid witness length");
  typedef bn254_projective_t P1;
  typedef bn254_g2_projective_t P2;
  P1 pi_a, pi_b1, pi_c, pi_h, t2;
  P2 pi_b;
  memcpy(&pi_a, points, 96);
  memcpy(&pi_b1, points + 96, 96);
  memcpy(&pi_b, points + 192, 192);
  memcpy(&pi_c, points + 384, 96);
  memcpy(&pi_h, points + 480, 96);
  const P1* alpha1 = (const P1*)&z->vk_alpha_1;
  const P1* beta1 = (const P1*)&z->vk_beta_1;
  const P2* beta2 = (const P2*)&z->vk_beta_2;
  // src/proof_helper.rs:280-283
  bn254_ecadd(&pi_a, alpha1, &pi_a);
  bn254_ecadd(&pi_a, &bl.d1r, &pi_a);           // pi_a = A + α1 + δ1·r
  bn254_g2_ecadd(&pi_b, beta2, &pi_b);
  bn254_g2_ecadd(&pi_b, &bl.d2s, &pi_b);        // pi_b = B2 + β2 + δ2·s
  bn254_ecadd(&pi_b1, beta1, &pi_b1);
  bn254_ecadd(&pi_b1, &bl.d1s, &pi_b1);         // pi_b1 = B1 + β1 + δ1·s
  bn254_ecadd(&pi_c, &pi_h, &pi_c);             // C + H
  if (et && et->done.load(std::memory_order_acquire) == 2) {
    bn254_ecadd(&pi_c, &et->ta, &pi_c); // computed by the tail threads of A and B1 while H was still running
    bn254_ecadd(&pi_c, &et->tb, &pi_c);
  } else {
    P1 ta, tb;
    std::thread th([&] { bn254_mul_scalar(&pi_a, &bl.s, &ta); }); // pi_a·s  ∥  pi_b1·r
    bn254_mul_scalar(&pi_b1, &bl.r, &tb);
    th.join();
    bn254_ecadd(&pi_c, &ta, &pi_c);
    bn254_ecadd(&pi_c, &tb, &pi_c);
  }
  bn254_ecsub(&pi_c, &bl.d1rs, &pi_c);          // − δ1·r·s
  (void)t2;
  bn254_affine_t c_aff;
  bn254_to_affine(&pi_c, &c_aff);
  auto dec = [](const void* p) { return dec32(p); };
  // pi_a and pi_b: from the tail threads of A and B2 when they got there (EarlyTerms), else here
  std::string a_dec[2], b_dec[4];
  if (et && et->a_ready.load(std::memory_order_acquire)) {
    a_dec[0] = et->a_dec[0];
    a_dec[1] = et->a_dec[1];
  } else {
    bn254_affine_t a_aff;
    bn254_to_affine(&pi_a, &a_aff);
    a_dec[0] = dec(&a_aff.x);
    a_dec[1] = dec(&a_aff.y);
  }
  if (et && et->b_ready.load(std::memory_order_acquire)) {
    for (int i = 0; i < 4; i++) b_dec[i] = et->b_dec[i];
  } else {
    bn254_g2_affine_t b_aff;
    bn254_g2_to_affine(&pi_b, &b_aff);
    b_dec[0] = dec(&b_aff.x.c0);
    b_dec[1] = dec(&b_aff.x.c1);
    b_dec[2] = dec(&b_aff.y.c0);
    b_dec[3] = dec(&b_aff.y.c1);
  }
  // serde_json::to_writer_pretty of a Value built with json!(proof): object keys sorted (BTreeMap), 2-space indent
  std::string pj = "{\n  \"curve\": \"bn128\",\n";
  pj += "  \"pi_a\": [\n    \"" + a_dec[0] + "\",\n    \"" + a_dec[1] + "\",\n    \"1\"\n  ],\n";
  pj += "  \"pi_b\": [\n    [\n      \"" + b_dec[0] + "\",\n      \"" + b_dec[1] + "\"\n    ],\n    [\n      \"" + b_dec[2] + "\",\n      \"" + b_dec[3] +
        "\"\n    ],\n    [\n      \"1\",\n      \"0\"\n    ]\n  ],\n";
  pj += "  \"pi_c\": [\n    \"" + dec(&c_aff.x) + "\",\n    \"" + dec(&c_aff.y) + "\",\n    \"1\"\n  ],\n";
  pj += "  \"protocol\": \"groth16\"\n}";
  // public signals: witness[1..=n_public] as decimal strings — src/proof_helper.rs:297-307
  std::string qj = z->n_public ? "[\n" : "[]";
  for (uint32_t i = 1; i <= z->n_public; i++) {
    qj += "  \"" + dec(w.values + (size_t)i * 32) + "\"";
    qj += i == z->n_public ? "\n]" : ",\n";
  }
  int need = 0;
  if (proof_json) {
    if (pj.size() + 1 > proof_cap) need = (int)pj.size() + 1;
    else memcpy(proof_json, pj.c_str(), pj.size() + 1);
  }
  if (public_json) {
    if (qj.size() + 1 > public_cap) need = need > (int)qj.size() + 1 ? need : (int)qj.size() + 1;
    else memcpy(public_json, qj.c_str(), qj.size() + 1);
  }
  if (need) return fail(need, "output buffer too small (need %d bytes)", need);
  return 0;
}

} // namespace prover
} // namespace isnark

extern "C" {

__attribute__((visibility("default"))) int groth16_sum_commitments(const uint8_t* blocks, int count, uint8_t out[GROTH16_COMMITMENTS_BYTES])
{
  if (!blocks || !out || count < 1) return fail(ERR_ARG, "bad argument");
  uint8_t acc[GROTH16_COMMITMENTS_BYTES];
  memcpy(acc, blocks, sizeof acc);
  static const int off[5] = {0, 96, 192, 384, 480};
  for (int k = 1; k < count; k++) {
    const uint8_t* b = blocks + (size_t)k * GROTH16_COMMITMENTS_BYTES;
    for (int j = 0; j < 5; j++) {
      if (j == 2) bn254_g2_ecadd((const bn254_g2_projective_t*)(acc + off[j]), (const bn254_g2_projective_t*)(b + off[j]), (bn254_g2_projective_t*)(acc + off[j]));
      else bn254_ecadd((const bn254_projective_t*)(acc + off[j]), (const bn254_projective_t*)(b + off[j]), (bn254_projective_t*)(acc + off[j]));
    }
  }
  memcpy(out, acc, sizeof acc);
  return 0;
}

__attribute__((visibility("default"))) int groth16_assemble_proof(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, const uint8_t points[GROTH16_COMMITMENTS_BYTES],
                                                                  const uint8_t* r_in, const uint8_t* s_in, char* proof_json, size_t proof_cap, char* public_json, size_t public_cap)
{
  if (!cm || !wtns || !points) return fail(ERR_ARG, "null argument");
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  ZKeyCache* z = zp.get();
  if (!z) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  Blinding bl;
  if (int rc = compute_blinding(z, r_in, s_in, &bl)) return rc;
  return assemble_impl(z, wtns, wtns_len, points, bl, proof_json, proof_cap, public_json, public_cap);
}

} // extern "C"
