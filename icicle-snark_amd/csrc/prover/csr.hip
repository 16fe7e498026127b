// csr.hip — cold path: zkey section 4 (coefficients) → device CSR, built on the device (see qap.h).
// Replaces the host loops of CacheManager::compute (src/cache.rs:126-166: parse, :214 from_mont) and the
// per-prove serial scatter-add they feed (src/proof_helper.rs:81-92).  The raw 44-byte records are uploaded
// once; streaming kernels (count → exclusive scan, msm_sort.hip's → scatter with the Montgomery conversion fused)
// produce rowptr/cols/vals.  Entries of one row land in arbitrary order — the row sum is exact field
// arithmetic, so the order is immaterial.
#include "../msm_plan.h"
#include "qap.h"

using namespace bn254;

namespace {

constexpr uint32_t REC_WORDS = 11; // {m:u32 c:u32 s:u32 value[8×u32]}

__global__ __launch_bounds__(256) void csr_count_kernel(const uint32_t* __restrict__ rec, uint32_t n_coef, uint32_t n, uint32_t n_vars, uint32_t* __restrict__ counts,
                                                         uint32_t* __restrict__ err)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_coef) return;
  const uint32_t* e = rec + (size_t)i * REC_WORDS;
  const uint32_t m = e[0] & 0xff, c = e[1], s = e[2]; // only byte 0 of m is read — src/cache.rs:159
  if (m > 1 || c >= n || s >= n_vars) {
    atomicMin(err, i); // first offending record
    return;
  }
  atomicAdd(counts + (size_t)m * n + c, 1u);
}

__global__ __launch_bounds__(256) void csr_scatter_kernel(const uint32_t* __restrict__ rec, uint32_t n_coef, uint32_t n, uint32_t* __restrict__ cursor, uint32_t* __restrict__ cols,
                                                           fe* __restrict__ vals)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_coef) return;
  const uint32_t* e = rec + (size_t)i * REC_WORDS;
  const uint32_t m = e[0] & 0xff, c = e[1];
  const uint32_t pos = atomicAdd(cursor + (size_t)m * n + c, 1u);
  cols[pos] = e[2];
  fe v;
#pragma unroll
  for (int k = 0; k < 8; k++) v.l[k] = e[3 + k];
  // the file stores value·R²; one from_mont (src/cache.rs:214) leaves value·R = Montgomery form of the coefficient
  v = Fr::from_mont(v);
  uint4* q = reinterpret_cast<uint4*>(vals + pos);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__global__ void csr_fill_kernel(uint32_t* p, size_t n, uint32_t v)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

} // namespace

namespace isnark {

hipError_t qap_build_csr(const uint32_t* d_records, uint32_t n_coef, uint32_t n, uint32_t n_vars, uint32_t* d_rowptr, uint32_t* d_cols, fe* d_vals, uint32_t* first_bad,
                         hipStream_t s)
{
  *first_bad = 0xffffffffu;
  const size_t rows = 2 * (size_t)n + 1;
  uint32_t *counts = nullptr, *err = nullptr, *tmp = nullptr;
  hipError_t e;
  if (rows >= (1ull << 32)) return hipErrorInvalidValue;
  if ((e = hipMalloc((void**)&counts, (rows + 1) * 4)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&tmp, exclusive_scan_u32_scratch_words((uint32_t)rows) * 4)) != hipSuccess) {
    (void)hipFree(counts);
    return e;
  }
  err = counts + rows;
  hipLaunchKernelGGL(csr_fill_kernel, dim3(1024), dim3(256), 0, s, counts, rows, 0u);
  hipLaunchKernelGGL(csr_fill_kernel, dim3(1), dim3(1), 0, s, err, (size_t)1, 0xffffffffu);
  if (n_coef) hipLaunchKernelGGL(csr_count_kernel, dim3((n_coef + 255) / 256), dim3(256), 0, s, d_records, n_coef, n, n_vars, counts, err);
  e = exclusive_scan_u32(counts, (uint32_t)rows, d_rowptr, tmp, s); // (msm_sort.hip: this library's own scan kernels)
  if (e == hipSuccess) e = hipMemcpyAsync(first_bad, err, 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess && *first_bad == 0xffffffffu && n_coef) {
    // cursor = copy of rowptr (the counts buffer is free again)
    e = hipMemcpyAsync(counts, d_rowptr, (rows - 1) * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(csr_scatter_kernel, dim3((n_coef + 255) / 256), dim3(256), 0, s, d_records, n_coef, n, counts, d_cols, d_vals);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  (void)hipFree(tmp);
  (void)hipFree(counts);
  return e;
}

} // namespace isnark

// first launch of a translation unit's code object loads it onto the device (milliseconds): prewarm_modules (runtime.cpp) does that ahead
// of the first prove of a process
namespace isnark {
__global__ void module_warm_csr_kernel() {}
void module_warm_csr(hipStream_t s) { hipLaunchKernelGGL(module_warm_csr_kernel, dim3(1), dim3(1), 0, s); }
} // namespace isnark
