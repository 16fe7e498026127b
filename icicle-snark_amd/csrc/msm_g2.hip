// msm_g2.hip — G2 (Fq2 coordinates) instantiation of the MSM pipeline (see msm_impl.h).
#define ISNARK_G2_ACC_EXTERN 1
#include "msm_impl.h"

namespace isnark {
eIcicleError msm_g2_partials(const SortPlan* pl, const void* d_points, int points_mont, uint32_t skip_below, hipStream_t s, void* d_partials, MsmProfile* prof, uint32_t row_len, int ticket_slot)
{
  return msm_buckets_run<G2>(pl, (const G2::A*)d_points, points_mont, skip_below, pl->g.tab ? row_len : 1, s, (G2::X*)d_partials, prof, ticket_slot);
}
eIcicleError msm_g2_accumulate(const SortPlan* pl, const void* d_points, int points_form, uint32_t skip_below, hipStream_t s, void* d_buckets, bool into, MsmProfile* prof, uint32_t row_len, bool resident, const LargeSide* side)
{
  return msm_accumulate_stage<G2>(pl, (const G2::A*)d_points, points_form, skip_below, pl->g.tab ? row_len : 1, s, (G2::X*)d_buckets, into, prof, resident, side);
}
eIcicleError msm_g2_reduce(const SortPlan* pl, hipStream_t s, const void* d_buckets, void* d_partials, int ticket_slot)
{
  return msm_reduce_stage<G2>(pl, s, (const G2::X*)d_buckets, (G2::X*)d_partials, ticket_slot);
}
eIcicleError msm_g2_build_table(const void* d_points, uint32_t n, int from_form, const MsmGeom& g, hipStream_t s, void** d_table)
{
  return build_table_run<G2, Fq2Ops>(d_points, n, from_form, g, s, d_table);
}
eIcicleError msm_g2_build_table_sliced(const void* d_points, uint32_t n, int from_form, const MsmGeom& g, hipStream_t s, void** d_table, const std::atomic<bool>* cancel)
{
  return build_table_sliced_run<G2, Fq2Ops>(d_points, n, from_form, g, s, d_table, table_slice_bases(), cancel);
}
void msm_g2_host_tail_tab(const void* h_partials, uint32_t nbits, bn254_g2_projective_t* out)
{
  G2::P p = msm_host_tail_tab<G2>((const G2::X*)h_partials, nbits);
  memcpy(out, &p, sizeof p);
}
eIcicleError msm_g2_points_to_internal(void* d_points, uint32_t n, int from_form, hipStream_t s) { return points_to_internal_run<G2>(d_points, n, from_form, s); }
void msm_g2_host_tail(const void* h_partials, uint32_t W, uint32_t bpw, int c, int wide, bn254_g2_projective_t* out)
{
  G2::P p = msm_host_tail<G2>((const G2::X*)h_partials, W, bpw, c, wide);
  memcpy(out, &p, sizeof p);
}
} // namespace isnark

ISNARK_API eIcicleError bn254_g2_msm(const bn254_scalar_t* scalars, const bn254_g2_affine_t* bases, int msm_size, const MSMConfig* cfg, bn254_g2_projective_t* results)
{
  return msm_impl<G2, Fq2Ops>(scalars, bases, msm_size, cfg, results);
}
ISNARK_API eIcicleError bn254_g2_msm_precompute_bases(const bn254_g2_affine_t* bases, int nof_bases, const MSMConfig* cfg, bn254_g2_affine_t* out)
{
  return precompute_impl<G2, Fq2Ops>(bases, nof_bases, cfg, out);
}
ISNARK_API eIcicleError icicle_snark_g2_generator_mul(const bn254_scalar_t* s, uint64_t n, icicleStreamHandle stream, bn254_g2_affine_t* out)
{
  // icicle/include/icicle/curves/params/bn254.h:32-39
  static const uint32_t xr[8] = {0xd992f6ed, 0x46debd5c, 0xf75edadd, 0x674322d4, 0x5e5c4479, 0x426a0066, 0x121f1e76, 0x1800deef};
  static const uint32_t xi[8] = {0xaef312c2, 0x97e485b7, 0x35a9e712, 0xf1aa4933, 0x31fb5d25, 0x7260bfb7, 0x920d483a, 0x198e9393};
  static const uint32_t yr[8] = {0x66fa7daa, 0x4ce6cc01, 0x0c43d37b, 0xe3d1e769, 0x8dcb408f, 0x4aab7180, 0xdb8c6deb, 0x12c85ea5};
  static const uint32_t yi[8] = {0xd122975b, 0x55acdadc, 0x70b38ef3, 0xbc4b3133, 0x690c3395, 0xec9e99ad, 0x585ff075, 0x090689d0};
  G2::A gen;
  memcpy(gen.x.c0.l, xr, 32);
  memcpy(gen.x.c1.l, xi, 32);
  memcpy(gen.y.c0.l, yr, 32);
  memcpy(gen.y.c1.l, yi, 32);
  return generator_mul_impl<G2, Fq2Ops>(s, n, (hipStream_t)stream, out, gen);
}

// first launch of a translation unit's code object loads it onto the device (milliseconds): prewarm_modules (runtime.cpp) does that ahead
// of the first prove of a process
namespace isnark {
__global__ void module_warm_g2_kernel() {}
void module_warm_g2(hipStream_t s) { hipLaunchKernelGGL(module_warm_g2_kernel, dim3(1), dim3(1), 0, s); }
} // namespace isnark
