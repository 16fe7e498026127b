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

enum VecOp { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_TO_MONT = 3, OP_FROM_MONT = 4, OP_DIV = 5 };

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
    else if (OP == OP_DIV) r = F::mul(x, F::inv(F::to_mont(ld_fe(b + i)))); // x·(b·R)⁻¹·… : inv(bR) = b⁻¹R⁻¹·R² ⇒ montmul(x, b⁻¹R) = x·b⁻¹ ; inverse(0) = 0 like the reference
    else if (OP == OP_TO_MONT) r = F::to_mont(x);
    else r = F::from_mont(x);
    st_fe(out + i, r);
  }
}

// out(b, i) = s[b] (op) v(b, i);  element (b, i) lives at b·size + i, or at b + i·batch with columns_batch
// (icicle/backend/cpu/src/field/cpu_vec_ops.cpp:254-272)
template <int OP>
__global__ __launch_bounds__(256) void scalar_vec_kernel(const fe* sc, const fe* v, fe* out, uint64_t size, uint32_t batch, int columns)
{
  const uint64_t total = size * batch, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint32_t b = columns ? (uint32_t)(i % batch) : (uint32_t)(i / size);
    const fe a = ld_fe(sc + b), x = ld_fe(v + i);
    fe r;
    if (OP == OP_ADD) r = Fr::add(a, x);
    else if (OP == OP_SUB) r = Fr::sub(a, x);
    else r = Fr::mul(Fr::mul(a, x), Fr::r2());
    st_fe(out + i, r);
  }
}

// Σ / Π over each batch vector, two stages.  Stage 1: grid (chunks, batch), each workgroup folds its chunk of the
// vector (products in Montgomery form) and leaves one partial; stage 2: one workgroup per batch folds the partials.
template <bool PRODUCT>
__device__ __forceinline__ fe fold(const fe& a, const fe& b) { return PRODUCT ? Fr::mul(a, b) : Fr::add(a, b); }
template <bool PRODUCT>
__global__ __launch_bounds__(256) void reduce_kernel(const fe* v, uint64_t size, uint32_t batch, int columns, int stage, fe* out)
{
  __shared__ fe sh[256];
  const uint32_t b = blockIdx.y;
  const uint64_t estride = stage == 1 && columns ? batch : 1;                       // distance between consecutive elements
  const fe* base = stage == 1 ? (columns ? v + b : v + (uint64_t)b * size) : v + (uint64_t)b * size;
  fe acc = PRODUCT ? Fr::one_mont() : Fr::zero();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += (uint64_t)gridDim.x * blockDim.x) {
    fe x = ld_fe(base + i * estride);
    if (PRODUCT && stage == 1) x = Fr::to_mont(x);
    acc = fold<PRODUCT>(acc, x);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] = fold<PRODUCT>(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    fe r = sh[0];
    if (stage == 2 && PRODUCT) r = Fr::from_mont(r);
    st_fe(out + (uint64_t)b * gridDim.x + blockIdx.x, r);
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
  // element-wise over size × batch_size elements: the batch layout (rows or columns) is irrelevant here
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

template <int OP>
eIcicleError run_scalar(const void* sc, const void* v, uint64_t size, const VecOpsConfig* cfg, void* out)
{
  if (!cfg || (size && (!sc || !v || !out))) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  const uint32_t batch = cfg->batch_size > 1 ? (uint32_t)cfg->batch_size : 1;
  const uint64_t total = size * batch;
  hipStream_t s = (hipStream_t)cfg->stream;
  Staged sa, sb, so;
  ICICLE_TRY(sa.in(sc, (size_t)batch * sizeof(fe), cfg->is_a_on_device, s));
  ICICLE_TRY(sb.in(v, total * sizeof(fe), cfg->is_b_on_device, s));
  ICICLE_TRY(so.out(out, total * sizeof(fe), cfg->is_result_on_device, s));
  if (total) {
    hipLaunchKernelGGL((scalar_vec_kernel<OP>), dim3(grid_for(total)), dim3(256), 0, s, sa.ptr<fe>(), sb.ptr<fe>(), so.ptr<fe>(), size, batch, cfg->columns_batch ? 1 : 0);
    ICICLE_TRY(check_launch("scalar_vec_kernel"));
  }
  ICICLE_TRY(so.finish());
  return end_call(s, cfg->is_async);
}

template <bool PRODUCT>
eIcicleError run_reduce(const void* v, uint64_t size, const VecOpsConfig* cfg, void* out)
{
  if (!cfg || !out || (size && !v)) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  const uint32_t batch = cfg->batch_size > 1 ? (uint32_t)cfg->batch_size : 1;
  hipStream_t s = (hipStream_t)cfg->stream;
  Staged sa, so;
  ICICLE_TRY(sa.in(v, size * batch * sizeof(fe), cfg->is_a_on_device, s));
  ICICLE_TRY(so.out(out, (size_t)batch * sizeof(fe), cfg->is_result_on_device, s));
  uint32_t chunks = (uint32_t)((size + 4095) / 4096);
  if (chunks == 0) chunks = 1;
  if (chunks > 1024) chunks = 1024;
  WsScoped<fe> partials;
  HIP_TRY(partials.alloc((size_t)chunks * batch, s), ICICLE_ALLOCATION_FAILED);
  hipLaunchKernelGGL((reduce_kernel<PRODUCT>), dim3(chunks, batch), dim3(256), 0, s, sa.ptr<fe>(), size, batch, cfg->columns_batch ? 1 : 0, 1, partials.p);
  hipLaunchKernelGGL((reduce_kernel<PRODUCT>), dim3(1, batch), dim3(256), 0, s, partials.p, (uint64_t)chunks, batch, 0, 2, so.ptr<fe>());
  ICICLE_TRY(check_launch("reduce_kernel"));
  partials.release();
  ICICLE_TRY(so.finish());
  return end_call(s, cfg->is_async);
}

} // namespace

// icicle/src/vec_ops.cpp:102-113 (a·b⁻¹), :55-65 (a += b), :118-161 (scalar ∘ vector), :9-34 (Σ, Π)
ISNARK_API eIcicleError bn254_vector_div(const bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out)
{
  return run<Fr, OP_DIV>(a, b, n, cfg, out, true);
}
ISNARK_API eIcicleError bn254_vector_accumulate(bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg)
{
  if (!cfg) return ICICLE_INVALID_POINTER;
  VecOpsConfig c = *cfg;
  c.is_result_on_device = cfg->is_a_on_device; // in place
  return run<Fr, OP_ADD>(a, b, n, &c, a, true);
}
ISNARK_API eIcicleError bn254_scalar_add_vec(const bn254_scalar_t* sc, const bn254_scalar_t* v, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out) { return run_scalar<OP_ADD>(sc, v, n, cfg, out); }
ISNARK_API eIcicleError bn254_scalar_sub_vec(const bn254_scalar_t* sc, const bn254_scalar_t* v, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out) { return run_scalar<OP_SUB>(sc, v, n, cfg, out); }
ISNARK_API eIcicleError bn254_scalar_mul_vec(const bn254_scalar_t* sc, const bn254_scalar_t* v, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out) { return run_scalar<OP_MUL>(sc, v, n, cfg, out); }
ISNARK_API eIcicleError bn254_vector_sum(const bn254_scalar_t* v, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out) { return run_reduce<false>(v, n, cfg, out); }
ISNARK_API eIcicleError bn254_vector_product(const bn254_scalar_t* v, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out) { return run_reduce<true>(v, n, cfg, out); }
// icicle/src/curves/montgomery_conversion.cpp:47-74: every Fq coordinate of the projective points
ISNARK_API eIcicleError bn254_projective_convert_montgomery(const bn254_projective_t* in, size_t n, bool is_into, const VecOpsConfig* cfg, bn254_projective_t* out)
{
  return is_into ? run<Fq, OP_TO_MONT>(in, nullptr, 3 * (uint64_t)n, cfg, out, false) : run<Fq, OP_FROM_MONT>(in, nullptr, 3 * (uint64_t)n, cfg, out, false);
}
ISNARK_API eIcicleError bn254_g2_projective_convert_montgomery(const bn254_g2_projective_t* in, size_t n, bool is_into, const VecOpsConfig* cfg, bn254_g2_projective_t* out)
{
  return is_into ? run<Fq, OP_TO_MONT>(in, nullptr, 6 * (uint64_t)n, cfg, out, false) : run<Fq, OP_FROM_MONT>(in, nullptr, 6 * (uint64_t)n, cfg, out, false);
}

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

// first launch of a translation unit's code object loads it onto the device (milliseconds): prewarm_modules (runtime.cpp) does that ahead
// of the first prove of a process
namespace isnark {
__global__ void module_warm_vec_kernel() {}
void module_warm_vec(hipStream_t s) { hipLaunchKernelGGL(module_warm_vec_kernel, dim3(1), dim3(1), 0, s); }
} // namespace isnark
