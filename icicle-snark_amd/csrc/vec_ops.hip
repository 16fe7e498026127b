// vec_ops.hip — pointwise Fr vector ops and Montgomery conversions (gfx950).
//
// Replaces icicle/backend/cuda/src/field/cuda_vec_ops.cu (mul_kernel :183, sub_kernel :148,
// add_kernel :105) and icicle/backend/cuda/include/cuda_mont.cuh (MontgomeryKernel :12-17) behind
// bn254_vector_{add,sub,mul}, bn254_scalar_convert_montgomery and
// bn254_{g2_,}affine_convert_montgomery.
//
// All of these stream 32-byte elements once: 96 B/element for the binary ops, 64 B/element for the
// conversions — HBM-bound by byte count.  One thread owns one element (two global_load_dwordx4 per
// operand, 16-B aligned), a wave therefore touches 2 KiB contiguous per operand per instruction
// pair; the grid is capped at 256 CUs × 8 blocks and strides (cdna_hip_programming.md G11/G13).
//
// Standard-form product without conversions: montmul(a, b) = a·b·R⁻¹, so a·b = montmul(montmul(a,b), R²).
#include "common.h"
#include "ff.h"

using namespace bn254;
using namespace isnark;

namespace {

enum VecOp { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_TO_MONT = 3, OP_FROM_MONT = 4 };

__device__ __forceinline__ fe ld_fe(const fe* p)
{
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 lo = q[0], hi = q[1];
  fe r;
  r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
  r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
  return r;
}
__device__ __forceinline__ void st_fe(fe* p, const fe& v)
{
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

template <class F, int OP>
__global__ __launch_bounds__(256) void vec_kernel(const fe* a, const fe* b, fe* out, uint64_t n) // no __restrict__: in-place use is part of the API
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    fe x = ld_fe(a + i);
    fe r;
    if (OP == OP_ADD) r = F::add(x, ld_fe(b + i));
    else if (OP == OP_SUB) r = F::sub(x, ld_fe(b + i));
    else if (OP == OP_MUL) r = F::mul(F::mul(x, ld_fe(b + i)), F::r2());
    else if (OP == OP_TO_MONT) r = F::to_mont(x);
    else r = F::from_mont(x);
    st_fe(out + i, r);
  }
}

inline int grid_for(uint64_t n)
{
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  return (int)blocks;
}

template <class F, int OP>
eIcicleError run(const void* a, const void* b, uint64_t n, const VecOpsConfig* cfg, void* out, bool has_b)
{
  if (!cfg || (n && (!a || !out || (has_b && !b)))) return ICICLE_INVALID_POINTER;
  if (cfg->batch_size != 1 && cfg->batch_size != 0 && cfg->columns_batch) {
    set_last_error("vec ops: columns_batch is not supported");
    return ICICLE_API_NOT_IMPLEMENTED;
  }
  ICICLE_TRY(require_device());
  const uint64_t total = n * (uint64_t)(cfg->batch_size > 1 ? cfg->batch_size : 1);
  hipStream_t s = (hipStream_t)cfg->stream;
  const size_t bytes = total * sizeof(fe);
  Staged sa, sb, so;
  ICICLE_TRY(sa.in(a, bytes, cfg->is_a_on_device, s));
  if (has_b) ICICLE_TRY(sb.in(b, bytes, cfg->is_b_on_device, s));
  ICICLE_TRY(so.out(out, bytes, cfg->is_result_on_device, s));
  if (total) {
    hipLaunchKernelGGL((vec_kernel<F, OP>), dim3(grid_for(total)), dim3(256), 0, s, sa.ptr<fe>(), has_b ? sb.ptr<fe>() : nullptr, so.ptr<fe>(), total);
    ICICLE_TRY(check_launch("vec_kernel"));
  }
  ICICLE_TRY(so.finish());
  // a host-resident result must be complete when a synchronous call returns; with is_async the
  // caller's stream sync covers the D2H enqueued above.
  return end_call(s, cfg->is_async);
}

} // namespace

ISNARK_API eIcicleError bn254_vector_add(const bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out)
{
  return run<Fr, OP_ADD>(a, b, n, cfg, out, true);
}
ISNARK_API eIcicleError bn254_vector_sub(const bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out)
{
  return run<Fr, OP_SUB>(a, b, n, cfg, out, true);
}
ISNARK_API eIcicleError bn254_vector_mul(const bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out)
{
  return run<Fr, OP_MUL>(a, b, n, cfg, out, true);
}
ISNARK_API eIcicleError bn254_scalar_convert_montgomery(const bn254_scalar_t* in, uint64_t n, bool to_mont, const VecOpsConfig* cfg, bn254_scalar_t* out)
{
  return to_mont ? run<Fr, OP_TO_MONT>(in, nullptr, n, cfg, out, false) : run<Fr, OP_FROM_MONT>(in, nullptr, n, cfg, out, false);
}
// points: every Fq coordinate independently (icicle/backend/cuda/src/curve/cuda_mont.cu)
ISNARK_API eIcicleError bn254_affine_convert_montgomery(const bn254_affine_t* in, uint64_t n, bool is_into, const VecOpsConfig* cfg, bn254_affine_t* out)
{
  return is_into ? run<Fq, OP_TO_MONT>(in, nullptr, 2 * n, cfg, out, false) : run<Fq, OP_FROM_MONT>(in, nullptr, 2 * n, cfg, out, false);
}
ISNARK_API eIcicleError bn254_g2_affine_convert_montgomery(const bn254_g2_affine_t* in, size_t n, bool is_into, const VecOpsConfig* cfg, bn254_g2_affine_t* out)
{
  return is_into ? run<Fq, OP_TO_MONT>(in, nullptr, 4 * (uint64_t)n, cfg, out, false) : run<Fq, OP_FROM_MONT>(in, nullptr, 4 * (uint64_t)n, cfg, out, false);
}
