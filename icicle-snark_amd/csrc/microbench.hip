// microbench.hip — the two machine constants SURVEY.md §8d asks the bench to MEASURE rather than assume:
// device-to-device copy bandwidth (HBM ceiling for streaming kernels) and the issue rate of v_mad_u64_u32 (the
// ceiling of the modular-arithmetic kernels).  Used by bench.py as denominators next to the nominal peaks.
#include <map>
#include <string.h>
#include <mutex>

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
// PMC calibration probes (MI355X guide: FETCH_SIZE / WRITE_SIZE are calibrated for wide coalesced streams only — "calibrate on a
// known byte count in your own access pattern"): known numbers of (a) 64-byte gathers at random 64-byte-aligned addresses of a
// table far larger than the caches — the access pattern of the bucket accumulation —, (b) 128-byte gathers (G2 bases),
// (c) coalesced 16-byte-per-lane streaming reads, (d) scattered 4-byte stores, (e) coalesced 4-byte stores.
__device__ __forceinline__ uint32_t probe_hash(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void probe_gather_kernel(const uint4* __restrict__ table, uint32_t n_slots, uint32_t per_thread, int quads, uint4* __restrict__ out)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t k = 0; k < per_thread; k++) {
    const uint32_t slot = probe_hash(t * per_thread + k) % n_slots; // slot of `quads` uint4
    for (int q = 0; q < quads; q++) {
      const uint4 v = table[(size_t)slot * quads + q];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  out[t] = acc;
}
__global__ __launch_bounds__(256) void probe_stream_kernel(const uint4* __restrict__ in, uint64_t n, uint4* __restrict__ out)
{
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = in[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void probe_store_kernel(uint32_t* __restrict__ dst, uint32_t n_words, uint32_t per_thread, int scattered)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t k = 0; k < per_thread; k++) {
    const uint32_t i = t * per_thread + k;
    const uint32_t pos = scattered ? probe_hash(i) % n_words : (k * gridDim.x * blockDim.x + t) % n_words;
    dst[pos] = i;
  }
}
} // namespace

// runs the five probes once each (kernel names probe_*): meant to be executed under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`;
// out[0..4] = the bytes each probe is known to move {gather64, gather128, stream, scattered 4-B stores, coalesced 4-B stores}
ISNARK_API eIcicleError icicle_snark_pmc_probes(double out[5])
{
  if (!out) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  const size_t table_bytes = (size_t)2 << 30; // 2 GiB: far beyond L2 (32 MiB) and the Infinity Cache (256 MiB)
  uint4 *table = nullptr, *o = nullptr;
  HIP_TRY(hipMalloc((void**)&table, table_bytes), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(hipMalloc((void**)&o, (size_t)4096 * 256 * 16), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(hipMemsetAsync(table, 1, table_bytes, nullptr), ICICLE_UNKNOWN_ERROR);
  const uint32_t blocks = 4096, per = 26;
  hipLaunchKernelGGL(probe_gather_kernel, dim3(blocks), dim3(256), 0, nullptr, table, (uint32_t)(table_bytes / 64), per, 4, o);
  out[0] = (double)blocks * 256 * per * 64;
  hipLaunchKernelGGL(probe_gather_kernel, dim3(blocks), dim3(256), 0, nullptr, table, (uint32_t)(table_bytes / 128), per, 8, o);
  out[1] = (double)blocks * 256 * per * 128;
  hipLaunchKernelGGL(probe_stream_kernel, dim3(blocks), dim3(256), 0, nullptr, table, (uint64_t)(table_bytes / 16), o);
  out[2] = (double)table_bytes;
  hipLaunchKernelGGL(probe_store_kernel, dim3(blocks), dim3(256), 0, nullptr, (uint32_t*)table, (uint32_t)(table_bytes / 4), 20u, 1);
  out[3] = (double)blocks * 256 * 20 * 4;
  hipLaunchKernelGGL(probe_store_kernel, dim3(blocks), dim3(256), 0, nullptr, (uint32_t*)table, (uint32_t)(table_bytes / 4), 20u, 0);
  out[4] = (double)blocks * 256 * 20 * 4;
  ICICLE_TRY(check_launch("pmc probes"));
  HIP_TRY(hipDeviceSynchronize(), ICICLE_SYNCHRONIZATION_FAILED);
  (void)hipFree(table);
  (void)hipFree(o);
  return ICICLE_SUCCESS;
}

// The five probes above, timed with HIP events: out[i] = GB/s of {64-byte gathers, 128-byte gathers, coalesced 16-byte reads,
// scattered 4-byte stores, coalesced 4-byte stores} over a 2 GiB buffer.  The boxes of one pool differ most in the scattered
// patterns (round 5: digit sort and table build 2–3× slower on some boxes at the same copy rate): bench.py prints these next to
// its line so that a reader can tell what kind of box a number comes from.  Takes ≈ 30 ms.
ISNARK_API eIcicleError icicle_snark_access_probes(double out[5])
{
  if (!out) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  const size_t table_bytes = (size_t)2 << 30;
  uint4 *table = nullptr, *o = nullptr;
  HIP_TRY(hipMalloc((void**)&table, table_bytes), ICICLE_ALLOCATION_FAILED);
  if (hipMalloc((void**)&o, (size_t)4096 * 256 * 16) != hipSuccess) {
    (void)hipFree(table);
    return ICICLE_ALLOCATION_FAILED;
  }
  hipEvent_t ev[6];
  for (auto& e : ev) (void)hipEventCreate(&e);
  (void)hipMemsetAsync(table, 1, table_bytes, nullptr);
  const uint32_t blocks = 4096, per = 26;
  double bytes[5];
  for (int rep = 0; rep < 2; rep++) { // (the first round warms the kernels' code object and the page tables)
    (void)hipEventRecord(ev[0], nullptr);
    hipLaunchKernelGGL(probe_gather_kernel, dim3(blocks), dim3(256), 0, nullptr, table, (uint32_t)(table_bytes / 64), per, 4, o);
    bytes[0] = (double)blocks * 256 * per * 64;
    (void)hipEventRecord(ev[1], nullptr);
    hipLaunchKernelGGL(probe_gather_kernel, dim3(blocks), dim3(256), 0, nullptr, table, (uint32_t)(table_bytes / 128), per, 8, o);
    bytes[1] = (double)blocks * 256 * per * 128;
    (void)hipEventRecord(ev[2], nullptr);
    hipLaunchKernelGGL(probe_stream_kernel, dim3(blocks), dim3(256), 0, nullptr, table, (uint64_t)(table_bytes / 16), o);
    bytes[2] = (double)table_bytes;
    (void)hipEventRecord(ev[3], nullptr);
    hipLaunchKernelGGL(probe_store_kernel, dim3(blocks), dim3(256), 0, nullptr, (uint32_t*)table, (uint32_t)(table_bytes / 4), 20u, 1);
    bytes[3] = (double)blocks * 256 * 20 * 4;
    (void)hipEventRecord(ev[4], nullptr);
    hipLaunchKernelGGL(probe_store_kernel, dim3(blocks), dim3(256), 0, nullptr, (uint32_t*)table, (uint32_t)(table_bytes / 4), 20u, 0);
    bytes[4] = (double)blocks * 256 * 20 * 4;
    (void)hipEventRecord(ev[5], nullptr);
  }
  const eIcicleError le = check_launch("access probes");
  const hipError_t se = hipEventSynchronize(ev[5]);
  for (int i = 0; i < 5; i++) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
    out[i] = ms > 0 ? bytes[i] / (ms * 1e-3) / 1e9 : 0;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  (void)hipFree(table);
  (void)hipFree(o);
  if (le != ICICLE_SUCCESS) return le;
  return se == hipSuccess ? ICICLE_SUCCESS : ICICLE_SYNCHRONIZATION_FAILED;
}

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

// ---- which streams share a hardware pipe --------------------------------------------------------------------------------------------
// The HIP runtime puts streams on hardware queues and queue k sits on pipe k mod 4 (profiles/r05_pipe_probe.txt).  A kernel with more
// workgroups than the GPU holds keeps its pipe busy dispatching for as long as it has workgroups to place, and the packets of every
// other queue on that pipe — kernels, event records — wait meanwhile (0.5–1 ms behind a 3 ms kernel).  The prover's transform passes and
// accumulations are such kernels, so WHICH of a key's six streams share a pipe decides whether the witness digit sort waits behind the
// front end (its slow mode: 2.2 instead of 0.9 ms) and whether two accumulation chains take turns.  The mapping is an accident of the
// order in which a process's streams were first used; this probe measures it: per round one oversubscribed kernel on stream X and a
// one-workgroup kernel + event on every other stream — those that complete late are X's pipe-mates.
namespace {
__global__ __launch_bounds__(256) void pipe_probe_big_kernel(unsigned long long* out, int iters)
{
  unsigned long long x = threadIdx.x + blockIdx.x;
  for (int i = 0; i < iters; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
  if (x == 42) out[0] = x;
}
__global__ void pipe_probe_tiny_kernel(unsigned long long* out)
{
  if (threadIdx.x == 999) out[1] = 1;
}
} // namespace

namespace isnark {
// cls[i] = pipe class of st[i] (0 … 3 in order of first appearance), or −1 everywhere when the measurement is not clean (another key
// proving beside it, an odd runtime): the caller then keeps the streams' roles as they are.  ≈ 3.3 ms per round, at most four rounds.
bool probe_stream_pipes(hipStream_t* st, int n, int* cls)
{
  for (int i = 0; i < n; i++) cls[i] = -1;
  if (n < 2 || n > 16) return false;
  unsigned long long* d = nullptr;
  if (hipMalloc((void**)&d, 64) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  hipEvent_t ref = nullptr, big_end = nullptr, stop[16] = {};
  bool ok = hipEventCreate(&ref) == hipSuccess && hipEventCreate(&big_end) == hipSuccess;
  for (int i = 0; i < n && ok; i++) ok = hipEventCreate(&stop[i]) == hipSuccess;
  // first use of every stream in a fixed order (a fresh stream gets its queue now), and a warm code object
  for (int i = 0; i < n && ok; i++) {
    hipLaunchKernelGGL(pipe_probe_tiny_kernel, dim3(1), dim3(64), 0, st[i], d);
    ok = hipStreamSynchronize(st[i]) == hipSuccess;
  }
  if (ok) {
    hipLaunchKernelGGL(pipe_probe_big_kernel, dim3(64), dim3(256), 0, st[0], d, 8);
    ok = hipStreamSynchronize(st[0]) == hipSuccess;
  }
  int next_class = 0;
  for (int x = 0; x < n && ok; x++) {
    if (cls[x] >= 0) continue;
    if (next_class >= 4) { // more than four classes cannot be: the measurement was disturbed
      ok = false;
      break;
    }
    if (next_class == 3) { // three pipes are known: whatever is left sits on the fourth (no round needed)
      for (int j = x; j < n; j++)
        if (cls[j] < 0) cls[j] = 3;
      break;
    }
    cls[x] = next_class;
    ok = hipEventRecord(ref, st[x]) == hipSuccess;
    hipLaunchKernelGGL(pipe_probe_big_kernel, dim3(40000), dim3(256), 0, st[x], d, 3000); // ≈ 3 ms: pipe-mates wait 0.5–1 ms, the others ≈ 0.1
    ok = ok && hipEventRecord(big_end, st[x]) == hipSuccess;
    for (int j = 0; j < n && ok; j++) {
      if (j == x) continue;
      hipLaunchKernelGGL(pipe_probe_tiny_kernel, dim3(1), dim3(64), 0, st[j], d);
      ok = hipEventRecord(stop[j], st[j]) == hipSuccess;
    }
    for (int j = 0; j < n && ok; j++) ok = hipStreamSynchronize(st[j]) == hipSuccess;
    float big_ms = 0;
    ok = ok && hipEventElapsedTime(&big_ms, ref, big_end) == hipSuccess;
    for (int j = 0; j < n && ok; j++) {
      if (j == x) continue;
      float ms = 0;
      ok = hipEventElapsedTime(&ms, ref, stop[j]) == hipSuccess;
      const bool mate = ms > 0.35f && ms > 0.1f * big_ms;
      if (mate) {
        if (cls[j] >= 0 && cls[j] != next_class) ok = false; // a stream on two pipes: disturbed
        cls[j] = next_class;
      }
    }
    next_class++;
  }
  (void)hipGetLastError();
  if (ref) (void)hipEventDestroy(ref);
  if (big_end) (void)hipEventDestroy(big_end);
  for (int i = 0; i < n; i++)
    if (stop[i]) (void)hipEventDestroy(stop[i]);
  (void)hipFree(d);
  if (!ok)
    for (int i = 0; i < n; i++) cls[i] = -1;
  return ok;
}

// Classes once measured are kept per stream for the life of the process (a stream keeps its queue; the library's streams are pooled and
// come back): the prewarm thread of a CacheManager measures the pool's streams in the background, a key built later looks its six
// up — and measures them itself (≈ 10 ms) only when one of them is new.  Labels are comparable within one measurement (`epoch`) only.
namespace {
std::mutex g_pipe_mu;
std::map<hipStream_t, std::pair<int, int>> g_pipe_cls; // stream → (epoch, class)
int g_pipe_epoch = 0;
} // namespace
void stream_pipes_measure(hipStream_t* st, int n)
{
  int cls[16], prev[16];
  if (n > 16) n = 16;
  // the tiny kernels are timed from the host's enqueue of the big one: a launching thread that is descheduled in between (a busy
  // host) makes strangers look like pipe-mates — accept a measurement only when two in a row agree (at most four)
  bool have_prev = false, agreed = false;
  for (int attempt = 0; attempt < 4 && !agreed; attempt++) {
    if (!probe_stream_pipes(st, n, cls)) {
      have_prev = false;
      continue;
    }
    agreed = have_prev && memcmp(cls, prev, sizeof(int) * n) == 0;
    memcpy(prev, cls, sizeof(int) * n);
    have_prev = true;
  }
  if (!agreed) return;
  std::lock_guard<std::mutex> lk(g_pipe_mu);
  const int ep = ++g_pipe_epoch;
  for (int i = 0; i < n; i++) g_pipe_cls[st[i]] = {ep, cls[i]};
}
bool stream_pipe_classes(hipStream_t* st, int n, int* cls, bool allow_measure)
{
  for (int attempt = 0; attempt < (allow_measure ? 2 : 1); attempt++) {
    {
      std::lock_guard<std::mutex> lk(g_pipe_mu);
      int ep = -1;
      bool all = true;
      for (int i = 0; i < n && all; i++) {
        const auto it = g_pipe_cls.find(st[i]);
        if (it == g_pipe_cls.end() || (ep >= 0 && it->second.first != ep)) all = false;
        else {
          ep = it->second.first;
          cls[i] = it->second.second;
        }
      }
      if (all) return true;
    }
    if (attempt == 0) stream_pipes_measure(st, n);
  }
  for (int i = 0; i < n; i++) cls[i] = -1;
  return false;
}
void stream_pipes_forget(hipStream_t st)
{
  std::lock_guard<std::mutex> lk(g_pipe_mu);
  g_pipe_cls.erase(st);
}
} // namespace isnark
