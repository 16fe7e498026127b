// exchange.hip — device side of the in-process multi-GPU exchanges (prover/multi.cpp).
//
// xGMI is a point-to-point fabric (7 links per GPU): with peer access enabled a kernel on GPU r reads the memory of every
// other GPU directly, all links at once.  The three exchanges of a sharded prove are therefore PULL kernels — one launch per
// rank and exchange on the rank's own stream, ordered behind the producers by events:
//   * witness all-gather   rank r copies slice p of rank p's witness buffer into its own, for every p ≠ r;
//   * all-to-all (twice)   rank r fills recv[q][p][·] from rank p's send[q][r][·] for the three rows q of the distributed
//                          QAP front end (the RCCL form of the same exchange: comm/rccl_comm.cpp, grouped ncclSend/ncclRecv).
// With several shards on ONE device (how a 1-GPU box exercises the 8-way path) the same kernels are plain device copies.
#include "exchange.h"

namespace isnark {
namespace {

// 16-byte words; every slice / chunk is a multiple of 32 bytes (field elements)
__global__ __launch_bounds__(256) void xchg_allgather_pull_kernel(PeerPtrs src, uint4* __restrict__ dst, uint32_t me, uint64_t slice_u4)
{
  const uint32_t p = blockIdx.y;
  if (p == me) return;
  const uint4* __restrict__ s = reinterpret_cast<const uint4*>(src.p[p]) + (uint64_t)p * slice_u4;
  uint4* __restrict__ d = dst + (uint64_t)p * slice_u4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slice_u4; i += (uint64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

__global__ __launch_bounds__(256) void xchg_alltoall_pull_kernel(PeerPtrs send, uint4* __restrict__ recv, uint32_t G, uint32_t me, uint64_t row_u4, uint64_t chunk_u4)
{
  const uint32_t q = blockIdx.y / G, p = blockIdx.y % G;
  // what rank p addressed to this rank: chunk `me` of its row q → chunk p of this rank's row q
  const uint4* __restrict__ s = reinterpret_cast<const uint4*>(send.p[p]) + (uint64_t)q * row_u4 + (uint64_t)me * chunk_u4;
  uint4* __restrict__ d = recv + (uint64_t)q * row_u4 + (uint64_t)p * chunk_u4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunk_u4; i += (uint64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// one lane spins for `ticks` of the constant-rate wall clock (self-test of the exchanges' ordering)
__global__ void xchg_delay_kernel(uint64_t ticks)
{
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

} // namespace

hipError_t xchg_delay(double ms, hipStream_t s)
{
  int dev = 0, khz = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
    (void)hipGetLastError();
    khz = 100000; // 100 MHz on CDNA3/4
  }
  hipLaunchKernelGGL(xchg_delay_kernel, dim3(1), dim3(1), 0, s, (uint64_t)(ms * (double)khz));
  return hipGetLastError();
}

hipError_t xchg_allgather_pull(const PeerPtrs& bufs, uint32_t G, uint32_t me, size_t slice_bytes, hipStream_t s)
{
  if (G < 2 || slice_bytes == 0) return hipSuccess;
  if (G > XCHG_MAX_PEERS || slice_bytes % 16) return hipErrorInvalidValue;
  const uint64_t u4 = slice_bytes / 16;
  unsigned bx = (unsigned)((u4 + 256 * 8 - 1) / (256 * 8)); // ≈ 8 words per thread
  if (bx > 512) bx = 512;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(xchg_allgather_pull_kernel, dim3(bx, G), dim3(256), 0, s, bufs, reinterpret_cast<uint4*>(const_cast<void*>(bufs.p[me])), me, u4);
  return hipGetLastError();
}

hipError_t xchg_alltoall_pull(const PeerPtrs& sends, void* recv, uint32_t G, uint32_t me, uint32_t rows, size_t row_bytes, size_t chunk_bytes, hipStream_t s)
{
  if (G < 1 || rows == 0 || chunk_bytes == 0) return hipSuccess;
  if (G > XCHG_MAX_PEERS || row_bytes % 16 || chunk_bytes % 16) return hipErrorInvalidValue;
  const uint64_t u4 = chunk_bytes / 16;
  unsigned bx = (unsigned)((u4 + 256 * 8 - 1) / (256 * 8));
  if (bx > 128) bx = 128;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(xchg_alltoall_pull_kernel, dim3(bx, rows * G), dim3(256), 0, s, sends, reinterpret_cast<uint4*>(recv), G, me, (uint64_t)(row_bytes / 16), u4);
  return hipGetLastError();
}

} // namespace isnark
