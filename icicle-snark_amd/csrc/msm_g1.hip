// msm_g1.hip — G1 instantiation of the MSM pipeline (see msm_impl.h).
#include "msm_impl.h"

namespace isnark {
thread_local float g_last_msm_ms[4] = {0, 0, 0, 0};

size_t msm_partials_bytes(const SortPlan* pl, bool g2, uint32_t* W, uint32_t* M)
{
  const size_t xs = g2 ? sizeof(G2::X) : sizeof(G1::X);
  if (M) *M = 1;
  if (pl->g.tab) {
    // table mode: [T | S_0 … S_{t−1}] of the bit-plane tree (msm_impl.h); *W = t
    uint32_t t = 0;
    while ((1u << t) < pl->nbuckets) t++;
    if (W) *W = t;
    return (size_t)(1 + t) * xs;
  }
  if (W) *W = (uint32_t)pl->g.Wb; // classic: one folded sum per window
  return (size_t)pl->g.Wb * xs;
}
eIcicleError msm_g1_partials(const SortPlan* pl, const void* d_points, int points_mont, uint32_t skip_below, hipStream_t s, void* d_partials, MsmProfile* prof, uint32_t row_len, int ticket_slot, const LargeSide* side)
{
  return msm_buckets_run<G1>(pl, (const G1::A*)d_points, points_mont, skip_below, pl->g.tab ? row_len : 1, s, (G1::X*)d_partials, prof, ticket_slot, side);
}
size_t msm_bucket_bytes(const SortPlan* pl, bool g2) { return (size_t)(pl->nbuckets ? pl->nbuckets : 1) * (g2 ? sizeof(G2::X) : sizeof(G1::X)); }
eIcicleError msm_g1_accumulate(const SortPlan* pl, const void* d_points, int points_form, uint32_t skip_below, hipStream_t s, void* d_buckets, bool into, MsmProfile* prof, uint32_t row_len, bool resident, const LargeSide* side)
{
  return msm_accumulate_stage<G1>(pl, (const G1::A*)d_points, points_form, skip_below, pl->g.tab ? row_len : 1, s, (G1::X*)d_buckets, into, prof, resident, side);
}
eIcicleError msm_g1_reduce(const SortPlan* pl, hipStream_t s, const void* d_buckets, void* d_partials, int ticket_slot)
{
  return msm_reduce_stage<G1>(pl, s, (const G1::X*)d_buckets, (G1::X*)d_partials, ticket_slot);
}
eIcicleError msm_g1_build_table(const void* d_points, uint32_t n, int from_form, const MsmGeom& g, hipStream_t s, void** d_table)
{
  return build_table_run<G1, FqOps>(d_points, n, from_form, g, s, d_table);
}
eIcicleError msm_g1_build_table_sliced(const void* d_points, uint32_t n, int from_form, const MsmGeom& g, hipStream_t s, void** d_table, const std::atomic<bool>* cancel)
{
  return build_table_sliced_run<G1, FqOps>(d_points, n, from_form, g, s, d_table, 2 * table_slice_bases(), cancel);
}
void msm_g1_host_tail_tab(const void* h_partials, uint32_t nbits, bn254_projective_t* out)
{
  G1::P p = msm_host_tail_tab<G1>((const G1::X*)h_partials, nbits);
  memcpy(out, &p, sizeof p);
}
eIcicleError msm_g1_points_to_internal(void* d_points, uint32_t n, int from_form, hipStream_t s) { return points_to_internal_run<G1>(d_points, n, from_form, s); }
void msm_g1_host_tail(const void* h_partials, uint32_t W, uint32_t bpw, int c, int wide, bn254_projective_t* out)
{
  G1::P p = msm_host_tail<G1>((const G1::X*)h_partials, W, bpw, c, wide);
  memcpy(out, &p, sizeof p);
}
} // namespace isnark

ISNARK_API eIcicleError bn254_msm(const bn254_scalar_t* scalars, const bn254_affine_t* bases, int msm_size, const MSMConfig* cfg, bn254_projective_t* results)
{
  return msm_impl<G1, FqOps>(scalars, bases, msm_size, cfg, results);
}
ISNARK_API eIcicleError bn254_msm_precompute_bases(const bn254_affine_t* bases, int nof_bases, const MSMConfig* cfg, bn254_affine_t* out)
{
  return precompute_impl<G1, FqOps>(bases, nof_bases, cfg, out);
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

// first launch of a translation unit's code object loads it onto the device (milliseconds): prewarm_modules (runtime.cpp) does that ahead
// of the first prove of a process
namespace isnark {
__global__ void module_warm_g1_kernel() {}
void module_warm_g1(hipStream_t s) { hipLaunchKernelGGL(module_warm_g1_kernel, dim3(1), dim3(1), 0, s); }
} // namespace isnark
