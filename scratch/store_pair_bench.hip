// does it matter on MI355X whether a 32-byte element leaves a lane as two 16-byte stores at p and p + 16 (every store
// instruction then covers HALF of each 32-byte sector), or whether neighbouring lanes exchange halves first so that each
// store instruction covers whole sectors?   hipcc --offload-arch=gfx950 -O3 store_pair_bench.hip -o store_pair_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
struct fe { uint4 a, b; };
__global__ void k_plain(fe* p, size_t n, uint32_t seed)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 lo = make_uint4(seed + i, 1, 2, 3), hi = make_uint4(4, 5, 6, seed ^ (uint32_t)i);
  uint4* q = reinterpret_cast<uint4*>(p + i);
  q[0] = lo; q[1] = hi;
}
__device__ __forceinline__ uint32_t swap1(uint32_t v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); } // quad_perm [1,0,3,2]
__global__ void k_paired(fe* p, size_t n, uint32_t seed)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return; // (n even: pairs stay together)
  uint4 lo = make_uint4(seed + i, 1, 2, 3), hi = make_uint4(4, 5, 6, seed ^ (uint32_t)i);
  const bool odd = threadIdx.x & 1;
  // even lane sends its high half and receives the partner's low half; the odd lane the other way round
  uint4 send = odd ? lo : hi, recv;
  recv.x = swap1(send.x); recv.y = swap1(send.y); recv.z = swap1(send.z); recv.w = swap1(send.w);
  uint4* qe = reinterpret_cast<uint4*>(p + (i & ~(size_t)1)); // the even element of the pair: 4 uint4 = E.lo E.hi O.lo O.hi
  // instruction 1 covers E (even lane: E.lo at +0, odd lane: E.hi at +1); instruction 2 covers O (even: O.lo at +2, odd: O.hi at +3)
  qe[odd ? 1 : 0] = odd ? recv : lo;
  qe[odd ? 3 : 2] = odd ? hi : recv;
}
// the whole wave's 64 elements (2 KiB, contiguous) leave as two fully contiguous 1 KiB store instructions: lane l writes half (l & 1)
// of element (l >> 1) and of element 32 + (l >> 1)
__device__ __forceinline__ uint32_t perm(uint32_t v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__global__ void k_wave(fe* p, size_t n, uint32_t seed)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 lo = make_uint4(seed + i, 1, 2, 3), hi = make_uint4(4, 5, 6, seed ^ (uint32_t)i);
  const int l = threadIdx.x & 63, h = l & 1, s0 = l >> 1, s1 = 32 + (l >> 1);
  uint4 a0, a1, b0, b1;
  a0.x = perm(lo.x, s0); a0.y = perm(lo.y, s0); a0.z = perm(lo.z, s0); a0.w = perm(lo.w, s0);
  a1.x = perm(hi.x, s0); a1.y = perm(hi.y, s0); a1.z = perm(hi.z, s0); a1.w = perm(hi.w, s0);
  b0.x = perm(lo.x, s1); b0.y = perm(lo.y, s1); b0.z = perm(lo.z, s1); b0.w = perm(lo.w, s1);
  b1.x = perm(hi.x, s1); b1.y = perm(hi.y, s1); b1.z = perm(hi.z, s1); b1.w = perm(hi.w, s1);
  uint4* q = reinterpret_cast<uint4*>(p + (i & ~(size_t)63));
  q[l] = h ? a1 : a0;
  q[64 + l] = h ? b1 : b0;
}
// runs of 8 elements (256 B) at scattered places — the output pattern of an NTT pass with 8 columns: plain vs halves gathered
// inside the group of 8 lanes so that each instruction writes one whole 128-byte line of the run
__device__ __forceinline__ size_t run_base(size_t g) { return ((g * 2654435761u) & ((1u << 22) - 1)) * 8; } // 2^22 runs of 8
__global__ void k_runs_plain(fe* p, size_t n, uint32_t seed)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 lo = make_uint4(seed + i, 1, 2, 3), hi = make_uint4(4, 5, 6, seed ^ (uint32_t)i);
  uint4* q = reinterpret_cast<uint4*>(p + run_base(i >> 3) + (i & 7));
  q[0] = lo; q[1] = hi;
}
__global__ void k_runs_grouped(fe* p, size_t n, uint32_t seed)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 lo = make_uint4(seed + i, 1, 2, 3), hi = make_uint4(4, 5, 6, seed ^ (uint32_t)i);
  const int l = threadIdx.x & 63, g0 = l & ~7, k = l & 7, h = k & 1, s0 = g0 + (k >> 1), s1 = g0 + 4 + (k >> 1);
  uint4 a0, a1, b0, b1;
  a0.x = perm(lo.x, s0); a0.y = perm(lo.y, s0); a0.z = perm(lo.z, s0); a0.w = perm(lo.w, s0);
  a1.x = perm(hi.x, s0); a1.y = perm(hi.y, s0); a1.z = perm(hi.z, s0); a1.w = perm(hi.w, s0);
  b0.x = perm(lo.x, s1); b0.y = perm(lo.y, s1); b0.z = perm(lo.z, s1); b0.w = perm(lo.w, s1);
  b1.x = perm(hi.x, s1); b1.y = perm(hi.y, s1); b1.z = perm(hi.z, s1); b1.w = perm(hi.w, s1);
  uint4* q = reinterpret_cast<uint4*>(p + run_base(i >> 3));
  q[k] = h ? a1 : a0;
  q[8 + k] = h ? b1 : b0;
}
__global__ void k_f4(uint4* p, size_t n4, uint32_t seed)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) p[i] = make_uint4(seed + i, 1, 2, 3);
}
int main()
{
  const size_t n = (size_t)1 << 25; // 1 GiB of 32-byte elements
  fe* d; hipMalloc(&d, n * sizeof(fe));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 6; which++) {
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_plain, dim3(n / 256), dim3(256), 0, 0, d, n, rep);
      if (which == 1) hipLaunchKernelGGL(k_paired, dim3(n / 256), dim3(256), 0, 0, d, n, rep);
      if (which == 2) hipLaunchKernelGGL(k_f4, dim3(2 * n / 256), dim3(256), 0, 0, (uint4*)d, 2 * n, rep);
      if (which == 3) hipLaunchKernelGGL(k_wave, dim3(n / 256), dim3(256), 0, 0, d, n, rep);
      if (which == 4) hipLaunchKernelGGL(k_runs_plain, dim3(n / 256), dim3(256), 0, 0, d, n, rep);
      if (which == 5) hipLaunchKernelGGL(k_runs_grouped, dim3(n / 256), dim3(256), 0, 0, d, n, rep);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const char* names[6] = {"two 16-byte stores per lane (p, p+16)", "halves exchanged between lane pairs    ", "float4 stream (16 B per lane)          ", "wave transposed: two contiguous 1 KiB  ",
                            "256-B runs scattered, plain            ", "256-B runs scattered, lines per instr. "};
    printf("%s: %.3f ms for 1 GiB written = %.0f GB/s\n", names[which], best, 1.0737 / (best * 1e-3));
  }
  // check k_paired writes what k_plain writes
  fe* h0 = (fe*)malloc(1 << 20); fe* h1 = (fe*)malloc(1 << 20);
  hipLaunchKernelGGL(k_plain, dim3(n / 256), dim3(256), 0, 0, d, n, 77); hipMemcpy(h0, d, 1 << 20, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_paired, dim3(n / 256), dim3(256), 0, 0, d, n, 77); hipMemcpy(h1, d, 1 << 20, hipMemcpyDeviceToHost);
  printf("paired == plain: %s\n", memcmp(h0, h1, 1 << 20) == 0 ? "yes" : "NO");
  hipLaunchKernelGGL(k_wave, dim3(n / 256), dim3(256), 0, 0, d, n, 77); hipMemcpy(h1, d, 1 << 20, hipMemcpyDeviceToHost);
  printf("wave == plain: %s\n", memcmp(h0, h1, 1 << 20) == 0 ? "yes" : "NO");
  hipMemset(d, 0, n * sizeof(fe));
  hipLaunchKernelGGL(k_runs_plain, dim3(n / 256), dim3(256), 0, 0, d, n, 77); hipMemcpy(h0, d, 1 << 20, hipMemcpyDeviceToHost);
  hipMemset(d, 0, n * sizeof(fe));
  hipLaunchKernelGGL(k_runs_grouped, dim3(n / 256), dim3(256), 0, 0, d, n, 77); hipMemcpy(h1, d, 1 << 20, hipMemcpyDeviceToHost);
  printf("runs grouped == runs plain: %s\n", memcmp(h0, h1, 1 << 20) == 0 ? "yes" : "NO");
  return 0;
}
