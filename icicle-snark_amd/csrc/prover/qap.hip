// qap.hip — QAP front end on the device (see qap.h).  All three kernels stream 32-byte elements:
//   spmv : reads 36 B per coefficient (value + column) + one gathered 32-B witness element, writes 96 B per row
//   coset: 3·64 B + 32 B per column;  final: 96 B read, 32 B written per element.
#include "qap.h"

using namespace bn254;

namespace {

__device__ __forceinline__ fe ld(const fe* p)
{
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  fe r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void st(fe* p, const fe& v)
{
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__global__ __launch_bounds__(256) void qap_spmv_kernel(const fe* __restrict__ w, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ cols,
                                                        const fe* __restrict__ vals, uint32_t n, fe* __restrict__ d_vec)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  // rows c (A) and n + c (B) advance in lockstep so that the dependent loads (rowptr → column → witness) of the
  // two rows are in flight together
  uint32_t ka = rowptr[c], kb = rowptr[n + c];
  const uint32_t ha = rowptr[c + 1], hb = rowptr[n + c + 1];
  fe a = Fr::zero(), b = Fr::zero();
  while (ka < ha || kb < hb) {
    const bool da = ka < ha, db = kb < hb;
    const uint32_t ia = da ? ka : 0u, ib = db ? kb : 0u;
    const uint32_t ca = cols[ia], cb = cols[ib];
    const fe va = ld(vals + ia), vb = ld(vals + ib);
    const fe wa = ld(w + ca), wb = ld(w + cb);
    if (da) a = Fr::add(a, Fr::mul(va, wa)); // coef·R ⊗ w = coef·w
    if (db) b = Fr::add(b, Fr::mul(vb, wb));
    ka++;
    kb++;
  }
  st(d_vec + c, b);                  // slot 0 = B   (src/proof_helper.rs:94-96)
  st(d_vec + (size_t)n + c, a);      // slot 1 = A   (:97-99)
  st(d_vec + 2 * (size_t)n + c, Fr::mul(Fr::mul(a, b), Fr::r2())); // slot 2 = A∘B (:108-114)
}

__global__ __launch_bounds__(256) void qap_coset_mul3_kernel(fe* d_vec, const fe* __restrict__ keys, uint32_t key_stride, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const fe k = ld(keys + (size_t)i * key_stride);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    fe* p = d_vec + (size_t)r * n + i;
    st(p, Fr::mul(ld(p), k));
  }
}

__global__ __launch_bounds__(256) void qap_final_kernel(fe* d_vec, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe a = ld(d_vec + i), b = ld(d_vec + (size_t)n + i), c = ld(d_vec + 2 * (size_t)n + i);
  st(d_vec + (size_t)n + i, Fr::sub(Fr::mul(Fr::mul(a, b), Fr::r2()), c));
}

} // namespace

namespace isnark {

hipError_t qap_spmv(const fe* witness, const uint32_t* rowptr, const uint32_t* cols, const fe* vals, uint32_t n, fe* d_vec, hipStream_t s)
{
  hipLaunchKernelGGL(qap_spmv_kernel, dim3((n + 255) / 256), dim3(256), 0, s, witness, rowptr, cols, vals, n, d_vec);
  return hipGetLastError();
}
hipError_t qap_coset_mul3(fe* d_vec, const fe* keys, uint32_t key_stride, uint32_t n, hipStream_t s)
{
  hipLaunchKernelGGL(qap_coset_mul3_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_vec, keys, key_stride, n);
  return hipGetLastError();
}
hipError_t qap_final(fe* d_vec, uint32_t n, hipStream_t s)
{
  hipLaunchKernelGGL(qap_final_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_vec, n);
  return hipGetLastError();
}

} // namespace isnark
