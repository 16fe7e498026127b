// msm_g1.hip — G1 instantiation of the MSM pipeline (see msm_impl.h).
#include <mutex>

#include "msm_impl.h"

namespace isnark {
thread_local float g_last_msm_ms[4] = {0, 0, 0, 0};
MsmProfile g_msm_ring[MSM_PROFILE_RING];
uint64_t g_msm_seq = 0;
static std::mutex g_msm_prof_mu;
MsmProfile* msm_profile_next(uint64_t* seq)
{
  std::lock_guard<std::mutex> lk(g_msm_prof_mu);
  MsmProfile* p = &g_msm_ring[g_msm_seq % MSM_PROFILE_RING];
  if (!p->ev[0])
    for (auto& e : p->ev) (void)hipEventCreate(&e);
  p->valid = false;
  *seq = g_msm_seq++;
  return p;
}
} // namespace isnark

// Profile of the `back`-th most recent MSM (0 = latest) issued by this process.  The caller must have
// synchronised the MSM's stream.  out_ms = {recode+sort, bucket accumulation kernel, large buckets +
// reduction + tail, total}; geom = {L, nbuckets, c, W, is_g2}.
ISNARK_API eIcicleError icicle_snark_msm_profile(int back, float out_ms[4], uint32_t geom[5])
{
  if (!out_ms || !geom || back < 0 || back >= MSM_PROFILE_RING || (uint64_t)back >= g_msm_seq) return ICICLE_INVALID_ARGUMENT;
  const MsmProfile& p = g_msm_ring[(g_msm_seq - 1 - back) % MSM_PROFILE_RING];
  if (!p.valid) return ICICLE_INVALID_ARGUMENT;
  if (hipEventElapsedTime(&out_ms[0], p.ev[0], p.ev[1]) != hipSuccess) return ICICLE_UNKNOWN_ERROR;
  (void)hipEventElapsedTime(&out_ms[1], p.ev[1], p.ev[2]);
  (void)hipEventElapsedTime(&out_ms[2], p.ev[2], p.ev[3]);
  (void)hipEventElapsedTime(&out_ms[3], p.ev[0], p.ev[3]);
  geom[0] = p.L; geom[1] = p.nbuckets; geom[2] = (uint32_t)p.c; geom[3] = (uint32_t)p.W; geom[4] = (uint32_t)p.is_g2;
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError bn254_msm(const bn254_scalar_t* scalars, const bn254_affine_t* bases, int msm_size, const MSMConfig* cfg, bn254_projective_t* results)
{
  return msm_impl<G1>(scalars, bases, msm_size, cfg, results);
}
ISNARK_API eIcicleError icicle_snark_last_msm_timings(float out_ms[4])
{
  if (!out_ms) return ICICLE_INVALID_POINTER;
  for (int i = 0; i < 4; i++) out_ms[i] = isnark::g_last_msm_ms[i];
  return ICICLE_SUCCESS;
}
ISNARK_API eIcicleError icicle_snark_g1_generator_mul(const bn254_scalar_t* s, uint64_t n, icicleStreamHandle stream, bn254_affine_t* out)
{
  G1::A gen = {Fq::one_std(), Fq::zero()};
  gen.y.l[0] = 2; // icicle/include/icicle/curves/params/bn254.h:21-24
  return generator_mul_impl<G1, FqOps>(s, n, (hipStream_t)stream, out, gen);
}
