// msm_g1.hip — G1 instantiation of the MSM pipeline (see msm_impl.h).
#include "msm_impl.h"

namespace isnark {
thread_local float g_last_msm_ms[4] = {0, 0, 0, 0};
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
