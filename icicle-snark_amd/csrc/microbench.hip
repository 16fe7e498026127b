// microbench.hip — the two machine constants SURVEY.md §8d asks the bench to MEASURE rather than assume:
// device-to-device copy bandwidth (HBM ceiling for streaming kernels) and the issue rate of v_mad_u64_u32 (the
// ceiling of the modular-arithmetic kernels).  Used by bench.py as denominators next to the nominal peaks.
#include "common.h"

using namespace isnark;

namespace {
__global__ __launch_bounds__(256) void mad_rate_kernel(uint64_t* out, const uint32_t* in, int iters)
{
  const uint32_t a = in[threadIdx.x & 7], b = in[(threadIdx.x + 1) & 7];
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      c0 += (uint64_t)a * b; c1 += (uint64_t)a * b; c2 += (uint64_t)a * b; c3 += (uint64_t)a * b;
      c4 += (uint64_t)a * b; c5 += (uint64_t)a * b; c6 += (uint64_t)a * b; c7 += (uint64_t)a * b;
      asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)); // eight independent chains
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
} // namespace

// out[0] = device-to-device copy rate in GB/s counting read + write bytes; out[1] = v_mad_u64_u32 lane-operations per
// second in units of 10^12.  Takes ≈20 ms.
ISNARK_API eIcicleError icicle_snark_microbench(double out[2])
{
  if (!out) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0), ICICLE_UNKNOWN_ERROR);
  HIP_TRY(hipEventCreate(&e1), ICICLE_UNKNOWN_ERROR);
  const size_t bytes = (size_t)1 << 30;
  void *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc(&a, bytes), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(hipMalloc(&b, bytes), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(hipMemsetAsync(a, 1, bytes, nullptr), ICICLE_UNKNOWN_ERROR);
  HIP_TRY(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, nullptr), ICICLE_COPY_FAILED); // warm-up
  HIP_TRY(hipEventRecord(e0, nullptr), ICICLE_UNKNOWN_ERROR);
  for (int i = 0; i < 4; i++) HIP_TRY(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, nullptr), ICICLE_COPY_FAILED);
  HIP_TRY(hipEventRecord(e1, nullptr), ICICLE_UNKNOWN_ERROR);
  HIP_TRY(hipEventSynchronize(e1), ICICLE_SYNCHRONIZATION_FAILED);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  out[0] = 4.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
  const int blocks = 4096, iters = 500;
  hipLaunchKernelGGL(mad_rate_kernel, dim3(blocks), dim3(256), 0, nullptr, (uint64_t*)b, (const uint32_t*)a, iters);
  HIP_TRY(hipEventRecord(e0, nullptr), ICICLE_UNKNOWN_ERROR);
  hipLaunchKernelGGL(mad_rate_kernel, dim3(blocks), dim3(256), 0, nullptr, (uint64_t*)b, (const uint32_t*)a, iters);
  HIP_TRY(hipEventRecord(e1, nullptr), ICICLE_UNKNOWN_ERROR);
  HIP_TRY(hipEventSynchronize(e1), ICICLE_SYNCHRONIZATION_FAILED);
  (void)hipEventElapsedTime(&ms, e0, e1);
  out[1] = (double)blocks * 256 * iters * 64.0 / (ms * 1e-3) / 1e12;
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ICICLE_SUCCESS;
}
