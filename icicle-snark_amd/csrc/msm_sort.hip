// msm_sort.hip — stage 1-3 of the MSM pipeline (see msm_impl.h): signed-digit recoding of the scalars and
// a counting sort of (point index | sign) by (window, bucket).
//
//  recode      s ↦ digits d_w ∈ [−2^(c−1), 2^(c−1)) with Σ d_w 2^(cw) = s: scalars above (r − 1)/2 are
//              negated first (s ← r − s, sign flipped), then t = s + H with H = Σ_w 2^(cw+c−1) makes
//              d_w = ((t >> cw) & (2^c − 1)) − 2^(c−1) independent per window (W = ⌊254/c⌋ + 1 windows, the
//              top one cannot overflow; table mode narrows the top windows by one bit so that they tile the 254 bits
//              exactly, MsmGeom.wide).  Signed digits halve the bucket count of the reference's unsigned
//              digits (cuda_msm.cuh:166-203).
//  histogram   one global atomic per non-zero digit into counts[w·NB + |d| − 1]            (2 MiB of counters at c = 16: L2-resident)
//  scan        exclusive prefix sum, three small kernels (block sums / top / finish) + list of large buckets
//  scatter     second pass over the scalars; returning atomic on a cursor copy gives each entry its slot.
//              The order inside a bucket is arbitrary — addition is commutative, results are not affected.
// HBM traffic: 2 × 32 B per scalar read, 4 B per digit written (the key is implicit in the position): the
// reference moves 8-byte (key, value) pairs through three CUB radix sorts + RLE + scan (cuda_msm.cuh:401-485, :561-636).
#include <atomic>
#include <mutex>
#include <string.h>

#include "msm_plan.h"

using namespace bn254;

namespace isnark {

// one ring per device: the events are recorded on streams of the device whose thread issues the MSM, and an event only
// records on a stream of the device it was created on (a process that drives several GPUs has one prover thread per device)
constexpr int MSM_PROFILE_DEVICES = 16;
struct MsmProfileRing {
  MsmProfile slots[MSM_PROFILE_RING];
  uint64_t seq = 0;
};
static MsmProfileRing g_msm_rings[MSM_PROFILE_DEVICES];
static std::mutex g_msm_prof_mu;
static MsmProfileRing& msm_ring_of_active_device()
{
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MSM_PROFILE_DEVICES) d = 0;
  return g_msm_rings[d];
}

MsmProfile* msm_profile_next()
{
  std::lock_guard<std::mutex> lk(g_msm_prof_mu);
  MsmProfileRing& r = msm_ring_of_active_device();
  MsmProfile* p = &r.slots[r.seq % MSM_PROFILE_RING];
  if (!p->ev[0])
    for (auto& e : p->ev) (void)hipEventCreate(&e);
  p->valid = false;
  p->has_sort_end = false;
  p->resolved = false;
  r.seq++;
  return p;
}

void msm_profile_own_init(MsmProfile* p)
{
  if (!p->ev[0])
    for (auto& e : p->ev) (void)hipEventCreate(&e);
  p->valid = false;
  p->has_sort_end = false;
  p->resolved = false;
}
void msm_profile_own_destroy(MsmProfile* p)
{
  for (auto& e : p->ev) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
}
static void msm_profile_resolve(const MsmProfile& p, float out_ms[5])
{
  for (int k = 0; k < 5; k++) out_ms[k] = 0;
  (void)hipEventElapsedTime(&out_ms[0], p.ev[0], p.ev[1]);
  (void)hipEventElapsedTime(&out_ms[1], p.ev[1], p.ev[2]);
  (void)hipEventElapsedTime(&out_ms[2], p.ev[2], p.ev[3]);
  (void)hipEventElapsedTime(&out_ms[3], p.ev[0], p.ev[3]);
  if (p.has_sort_end) (void)hipEventElapsedTime(&out_ms[4], p.ev[0], p.ev[4]);
  (void)hipGetLastError();
}
// the caller's streams are synchronised: the ring of the active device receives n resolved copies, in order (the last one is
// "back = 0" of icicle_snark_msm_profile); the ring's own events of those slots are left alone
void msm_profile_publish(const MsmProfile* src, int n)
{
  std::lock_guard<std::mutex> lk(g_msm_prof_mu);
  MsmProfileRing& r = msm_ring_of_active_device();
  for (int k = 0; k < n; k++) {
    MsmProfile* p = &r.slots[r.seq % MSM_PROFILE_RING];
    hipEvent_t keep[5];
    memcpy(keep, p->ev, sizeof keep);
    *p = src[k];
    memcpy(p->ev, keep, sizeof keep);
    p->resolved = true;
    for (float& m : p->ms) m = 0;
    if (src[k].valid && src[k].L) msm_profile_resolve(src[k], p->ms); // (L = 0: a slot that stands for "did not run", e.g. the head sort of an unsplit witness)
    r.seq++;
  }
}

namespace {

__device__ __forceinline__ fe ld_fe(const fe* p)
{
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  fe r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}

// scalar → t = s' + H (9 limbs), neg = (s was replaced by r − s)
__device__ __forceinline__ void recode(const fe* scalars, uint32_t i, const MsmGeom& g, int mont, uint32_t t[9], uint32_t& neg)
{
  fe s = ld_fe(scalars + i);
  if (mont) s = Fr::from_mont(s);
  // s > (r − 1)/2 is replaced by r − s: the value recoded is then ≤ (r − 1)/2 < 0.756·2^253, which leaves the top window —
  // exactly the scalar's last bits in table mode — room for its half-range offset and a carry
  constexpr uint32_t HALF[8] = {0xf8000000u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};
  neg = 0;
  bool decided = false;
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    if (!decided && s.l[k] != HALF[k]) {
      neg = s.l[k] > HALF[k] ? 1u : 0u;
      decided = true;
    }
  }
  if (neg) s = Fr::neg(s);
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    c += (uint64_t)s.l[k] + g.H[k];
    t[k] = (uint32_t)c;
    c >>= 32;
  }
  t[8] = (uint32_t)c + g.H[8];
}
// signed digit of window w: 0 for a zero digit, else |d| with the sign in bit 31
__device__ __forceinline__ uint32_t digit(const uint32_t t[9], int w, const MsmGeom& g)
{
  const int cw = w < g.wide ? g.c : g.c - 1;
  const int bit = w < g.wide ? w * g.c : g.wide * g.c + (w - g.wide) * (g.c - 1);
  const int limb = bit >> 5, off = bit & 31;
  uint64_t v = t[limb];
  if (limb < 8) v |= (uint64_t)t[limb + 1] << 32;
  const uint32_t raw = (uint32_t)(v >> off) & ((1u << cw) - 1);
  const int32_t d = (int32_t)raw - (int32_t)(1u << (cw - 1));
  if (d == 0) return 0;
  return d < 0 ? ((uint32_t)(-d) | 0x80000000u) : (uint32_t)d;
}

// bucket of digit magnitude bk+1 of window w, and the entry that names the point
// (classic layout with precomputed bases: window w uses multiple j = w / nbms of the base and the bucket set w mod nbms)
__device__ __forceinline__ uint32_t bucket_id(const MsmGeom& g, int w, uint32_t bk)
{
  if (g.tab) return bk;
  const int wm = g.pf > 1 ? w % g.nbms : w;
  return (uint32_t)wm * g.NB + bk;
}
__device__ __forceinline__ uint32_t entry_idx(const MsmGeom& g, int w, uint32_t i)
{
  if (g.tab) return i | ((uint32_t)w << g.IB);
  return g.pf > 1 ? i * (uint32_t)g.pf + (uint32_t)(w / g.nbms) : i;
}

// zero `n` u32 words (a kernel instead of hipMemsetAsync keeps every dependency on the compute queue)
__global__ __launch_bounds__(256) void msm_zero_kernel(uint32_t* __restrict__ p, uint32_t n)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0;
}

// Small digits of the two lowest windows are first counted in LDS: real witnesses are dominated by 0/1 and other
// small values, i.e. by a handful of buckets of windows 0/1; without this, a third of all global atomics of a
// witness-like scalar set hit ONE counter (measured 4.6 ms instead of 1 ms at L = 2^20).
constexpr uint32_t HIST_HOT = 512; // cached magnitudes per window, windows 0 and 1
__global__ __launch_bounds__(256) void msm_hist_kernel(const fe* __restrict__ scalars, uint32_t L, MsmGeom g, int mont, uint32_t* __restrict__ counts)
{
  __shared__ uint32_t hot[2 * HIST_HOT];
  for (uint32_t k = threadIdx.x; k < 2 * HIST_HOT; k += blockDim.x) hot[k] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < L) {
    uint32_t t[9], neg;
    recode(scalars, i, g, mont, t, neg);
    for (int w = 0; w < g.W; w++) {
      const uint32_t d = digit(t, w, g);
      if (d) {
        const uint32_t bk = (d & 0x7fffffffu) - 1;
        if ((g.tab || w < 2) && bk < HIST_HOT && bk < g.NB) atomicAdd(&hot[(g.tab ? 0 : w) * HIST_HOT + bk], 1u);
        else atomicAdd(&counts[bucket_id(g, w, bk)], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < 2 * HIST_HOT; k += blockDim.x) {
    const uint32_t v = hot[k];
    if (v) atomicAdd(&counts[bucket_id(g, (int)(k / HIST_HOT), k % HIST_HOT)], v);
  }
}

__global__ __launch_bounds__(256) void msm_scatter_kernel(const fe* __restrict__ scalars, uint32_t L, MsmGeom g, int mont, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  uint32_t t[9], neg;
  recode(scalars, i, g, mont, t, neg);
  for (int w = 0; w < g.W; w++) {
    const uint32_t d = digit(t, w, g);
    if (d) {
      const uint32_t pos = atomicAdd(&cursor[bucket_id(g, w, (d & 0x7fffffffu) - 1)], 1u);
      const uint32_t sign = (d >> 31) ^ neg;
      sorted[pos] = entry_idx(g, w, i) | (sign << 31);
    }
  }
}

// ---- exclusive scan of m counters: 2048 per workgroup (256 threads × 8) --------------------------
constexpr int SCAN_T = 256, SCAN_E = 8, SCAN_B = SCAN_T * SCAN_E;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* sh, uint32_t* total)
{
  const uint32_t tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (uint32_t d = 1; d < SCAN_T; d <<= 1) {
    uint32_t x = tid >= d ? sh[tid - d] : 0;
    __syncthreads();
    sh[tid] += x;
    __syncthreads();
  }
  if (total) *total = sh[SCAN_T - 1];
  return sh[tid] - v;
}

__global__ __launch_bounds__(SCAN_T) void msm_scan_sums_kernel(const uint32_t* __restrict__ counts, uint32_t m, uint32_t* __restrict__ bsum)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t sh[SCAN_T];
  const uint32_t base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_E;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_E; k++)
    if (base + k < m) s += counts[base + k];
  uint32_t total;
  block_exclusive_scan(s, sh, &total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(SCAN_T) void msm_scan_top_kernel(uint32_t* bsum, uint32_t nblocks)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t sh[SCAN_T];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nblocks; base += SCAN_T) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblocks ? bsum[i] : 0;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, sh, &total);
    if (i < nblocks) bsum[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
}
__global__ __launch_bounds__(SCAN_T) void msm_scan_finish_kernel(const uint32_t* __restrict__ counts, uint32_t m, const uint32_t* __restrict__ bsum, uint32_t* __restrict__ offsets,
                                                                  uint32_t* __restrict__ cursor, uint32_t thr, uint32_t* __restrict__ n_large, uint32_t* __restrict__ large_list,
                                                                  uint32_t* __restrict__ large_first, uint2* __restrict__ large_items, uint32_t item_cap)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t sh[SCAN_T];
  const uint32_t base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_E;
  uint32_t v[SCAN_E], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_E; k++) {
    v[k] = base + k < m ? counts[base + k] : 0;
    s += v[k];
  }
  uint32_t run = bsum[blockIdx.x] + block_exclusive_scan(s, sh, nullptr);
#pragma unroll
  for (int k = 0; k < SCAN_E; k++) {
    if (base + k < m) {
      offsets[base + k] = run;
      cursor[base + k] = run;
      if (v[k] > thr) {
        // a large bucket becomes ⌈count / MSM_LARGE_CHUNK⌉ work items, each summed by one workgroup
        const uint32_t li = atomicAdd(n_large, 1u);
        const uint32_t nch = (v[k] + MSM_LARGE_CHUNK - 1) / MSM_LARGE_CHUNK;
        const uint32_t first = atomicAdd(n_large + 2, nch);
        large_list[li] = base + k;
        large_first[li] = first;
        for (uint32_t q = 0; q < nch && first + q < item_cap; q++) large_items[first + q] = make_uint2(base + k, q);
      }
      run += v[k];
    }
  }
}

// generic finish of the 3-kernel exclusive scan: out[i] = bsum[block] + local exclusive prefix
__global__ __launch_bounds__(SCAN_T) void msm_scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t m, const uint32_t* __restrict__ bsum, uint32_t* __restrict__ out)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t sh[SCAN_T];
  const uint32_t base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_E;
  uint32_t v[SCAN_E], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_E; k++) {
    v[k] = base + k < m ? in[base + k] : 0;
    s += v[k];
  }
  uint32_t run = bsum[blockIdx.x] + block_exclusive_scan(s, sh, nullptr);
#pragma unroll
  for (int k = 0; k < SCAN_E; k++) {
    if (base + k < m) out[base + k] = run;
    run += v[k];
  }
}

// ---- order buckets by decreasing size (counting sort on min(count, ORDER_BINS-1)) ----------------
// One thread accumulates one bucket; buckets of a wave should hold the same number of points or the
// wave runs for its largest bucket (random digits: Poisson sizes, a quarter of the lanes idle —
// SQ_THREAD_CYCLES_VALU / (64·SQ_ACTIVE_INST_VALU) = 0.77 before this ordering, 0.99 after).  The
// reference sorts bucket sizes with two more CUB radix sorts (cuda_msm.cuh:606-630); sizes are small
// integers, so a counting sort is enough: per-workgroup LDS histograms, one scan, LDS ranks — no global
// atomics (a first version with global atomics on ~100 hot counters cost 1.1 ms at 2^19 buckets).
constexpr int ORDER_BINS = 256;
__device__ __forceinline__ uint32_t order_key(uint32_t cnt) { return (uint32_t)ORDER_BINS - 1 - min(cnt, (uint32_t)ORDER_BINS - 1); } // ascending key = descending size
__global__ __launch_bounds__(ORDER_BINS) void msm_order_hist_kernel(const uint32_t* __restrict__ counts, uint32_t m, uint32_t nblk, uint32_t* __restrict__ blockhist)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t h[ORDER_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * ORDER_BINS + threadIdx.x;
  if (b < m) atomicAdd(&h[order_key(counts[b])], 1u);
  __syncthreads();
  blockhist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x]; // key-major
}
__global__ __launch_bounds__(ORDER_BINS) void msm_order_scatter_kernel(const uint32_t* __restrict__ counts, uint32_t m, uint32_t nblk, const uint32_t* __restrict__ scanned, uint32_t* __restrict__ order)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t h[ORDER_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * ORDER_BINS + threadIdx.x;
  if (b < m) {
    const uint32_t key = order_key(counts[b]);
    const uint32_t r = atomicAdd(&h[key], 1u);
    order[scanned[key * nblk + blockIdx.x] + r] = b;
  }
}

// ---- two-level scatter ----------------------------------------------------------------------------
// The one-level scatter writes 4 bytes at a random position of a 134 MB array per digit (PMC: WRITE_SIZE 2.0 GB for
// 134 MB of payload at L = 2^21 — every store is its own partial-line write).  Two levels keep every store inside
// a small, densely written region:
//  A  (msm_partition_kernel)   a workgroup takes 4096 scalars × all windows, counts its digits per PARTITION
//     (the top ≤8 bits of the bucket index; LDS atomics), reserves a run per partition with one global atomic,
//     and writes (index | low bucket bits | sign) into that run — ≈16 consecutive entries (64 B) per partition.
//  B  (msm_fine_count_kernel, msm_fine_place_kernel) 8 workgroups per partition (≈8 K entries, 32 KB, L2-resident)
//     count the ≤128 buckets of the partition in LDS, derive their offsets and place the entries at offset + rank.
constexpr int PA_THREADS = 256, PA_PER_THREAD = 16, PA_SCALARS = PA_THREADS * PA_PER_THREAD;

__global__ __launch_bounds__(PA_THREADS) void msm_partition_kernel(const fe* __restrict__ scalars, uint32_t L, MsmGeom g, int mont, int low_bits, uint32_t NP, uint32_t nparts,
                                                                    uint32_t* __restrict__ part_cursor, uint32_t* __restrict__ tmp)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  extern __shared__ uint32_t sh[];
  uint32_t* hist = sh;
  uint32_t* cur = sh + nparts;
  for (uint32_t p = threadIdx.x; p < nparts; p += PA_THREADS) hist[p] = 0;
  __syncthreads();
  const uint32_t first = blockIdx.x * PA_SCALARS;
  for (int u = 0; u < PA_PER_THREAD; u++) {
    const uint32_t i = first + u * PA_THREADS + threadIdx.x;
    if (i < L) {
      uint32_t t[9], neg;
      recode(scalars, i, g, mont, t, neg);
      for (int w = 0; w < g.W; w++) {
        const uint32_t d = digit(t, w, g);
        if (d) atomicAdd(&hist[bucket_id(g, w, (d & 0x7fffffffu) - 1) >> low_bits], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < nparts; p += PA_THREADS) {
    const uint32_t h = hist[p];
    cur[p] = h ? atomicAdd(&part_cursor[p], h) : 0;
  }
  __syncthreads();
  const uint32_t low_mask = (1u << low_bits) - 1;
  const int fs = 31 - low_bits; // entry = index bits [0, fs) | low bucket bits [fs, 31) | sign
  for (int u = 0; u < PA_PER_THREAD; u++) {
    const uint32_t i = first + u * PA_THREADS + threadIdx.x;
    if (i < L) {
      uint32_t t[9], neg;
      recode(scalars, i, g, mont, t, neg);
      for (int w = 0; w < g.W; w++) {
        const uint32_t d = digit(t, w, g);
        if (d) {
          const uint32_t bk = bucket_id(g, w, (d & 0x7fffffffu) - 1);
          const uint32_t pos = atomicAdd(&cur[bk >> low_bits], 1u);
          tmp[pos] = entry_idx(g, w, i) | ((bk & low_mask) << fs) | (((d >> 31) ^ neg) << 31);
        }
      }
    }
  }
}


// ---- counts and offsets from the partitions (no global histogram) ------------------------------------------------
// The per-digit global atomics of msm_hist_kernel (25.6 M at 1.6 M constraints, 1.1 ms, and the main source of
// interference with the kernels running beside the sort) are only needed by the one-level scatter.  The two-level
// path counts digits per PARTITION in LDS (coarse histogram: one global atomic per workgroup and partition), scans
// the ≤ 8192 partition totals in one workgroup, partitions, and then derives the per-bucket counts and offsets
// inside each partition (≤ 128 buckets, LDS).
__global__ __launch_bounds__(PA_THREADS) void msm_coarse_hist_kernel(const fe* __restrict__ scalars, uint32_t L, MsmGeom g, int mont, int low_bits, uint32_t nparts,
                                                                      uint32_t* __restrict__ part_count)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  extern __shared__ uint32_t sh[];
  for (uint32_t p = threadIdx.x; p < nparts; p += PA_THREADS) sh[p] = 0;
  __syncthreads();
  const uint32_t first = blockIdx.x * PA_SCALARS;
  for (int u = 0; u < PA_PER_THREAD; u++) {
    const uint32_t i = first + u * PA_THREADS + threadIdx.x;
    if (i < L) {
      uint32_t t[9], neg;
      recode(scalars, i, g, mont, t, neg);
      for (int w = 0; w < g.W; w++) {
        const uint32_t d = digit(t, w, g);
        if (d) atomicAdd(&sh[bucket_id(g, w, (d & 0x7fffffffu) - 1) >> low_bits], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < nparts; p += PA_THREADS) {
    const uint32_t h = sh[p];
    if (h) atomicAdd(&part_count[p], h);
  }
}
// exclusive scan of the partition totals, one workgroup (nparts ≤ 8192 = 256 threads × 32)
__global__ __launch_bounds__(SCAN_T) void msm_part_scan_kernel(const uint32_t* __restrict__ part_count, uint32_t nparts, uint32_t* __restrict__ part_start, uint32_t* __restrict__ part_cursor)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t sh[SCAN_T];
  const uint32_t per = (nparts + SCAN_T - 1) / SCAN_T, lo = threadIdx.x * per;
  uint32_t s = 0;
  for (uint32_t k = 0; k < per; k++)
    if (lo + k < nparts) s += part_count[lo + k];
  uint32_t total;
  uint32_t run = block_exclusive_scan(s, sh, &total);
  for (uint32_t k = 0; k < per; k++)
    if (lo + k < nparts) {
      part_start[lo + k] = run;
      part_cursor[lo + k] = run;
      run += part_count[lo + k];
    }
  if (threadIdx.x == 0) part_start[nparts] = total;
}
constexpr int PB_SPLIT = 8; // workgroups per partition (a witness-like scalar set puts a third of all entries into ONE partition)
__global__ __launch_bounds__(256) void msm_fine_count_kernel(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ part_start, int low_bits, uint32_t* __restrict__ counts)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t hist[128];
  const uint32_t part = blockIdx.x, b0 = part << low_bits, nbk = 1u << low_bits;
  const int fs = 31 - low_bits;
  const uint32_t fmask = nbk - 1;
  const uint32_t start = part_start[part], end = part_start[part + 1];
  const uint32_t stride = blockDim.x * gridDim.y, first = start + blockIdx.y * blockDim.x + threadIdx.x;
  if (first - threadIdx.x >= end) return;
  if (threadIdx.x < 128) hist[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t e = first; e < end; e += stride) atomicAdd(&hist[(tmp[e] >> fs) & fmask], 1u);
  __syncthreads();
  if (threadIdx.x < nbk) {
    const uint32_t h = hist[threadIdx.x];
    if (h) atomicAdd(&counts[b0 + threadIdx.x], h);
  }
}
// offsets of the partition's buckets (every workgroup recomputes them from the counts; split 0 publishes them and
// registers large buckets), then this workgroup's stripe of entries is placed.  cursor[] starts at zero.
__global__ __launch_bounds__(256) void msm_fine_place_kernel(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ part_start, const uint32_t* __restrict__ counts,
                                                             uint32_t* __restrict__ cursor, int low_bits, uint32_t thr, uint32_t* __restrict__ offsets, uint32_t* __restrict__ n_large,
                                                             uint32_t* __restrict__ large_list, uint32_t* __restrict__ large_first, uint2* __restrict__ large_items, uint32_t item_cap,
                                                             uint32_t* __restrict__ sorted)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t offs[128], hist[128], cur[128];
  const uint32_t part = blockIdx.x, b0 = part << low_bits, nbk = 1u << low_bits;
  const int fs = 31 - low_bits;
  const uint32_t fmask = nbk - 1;
  const uint32_t start = part_start[part], end = part_start[part + 1];
  const uint32_t mine = threadIdx.x < nbk ? counts[b0 + threadIdx.x] : 0;
  if (threadIdx.x < 128) {
    offs[threadIdx.x] = mine;
    hist[threadIdx.x] = 0;
  }
  __syncthreads();
  for (uint32_t d = 1; d < 128; d <<= 1) { // inclusive scan over ≤ 128 counters
    uint32_t x = 0;
    if (threadIdx.x < 128 && threadIdx.x >= d) x = offs[threadIdx.x - d];
    __syncthreads();
    if (threadIdx.x < 128) offs[threadIdx.x] += x;
    __syncthreads();
  }
  if (threadIdx.x < nbk) {
    const uint32_t o = start + offs[threadIdx.x] - mine;
    cur[threadIdx.x] = o;
    if (blockIdx.y == 0) {
      offsets[b0 + threadIdx.x] = o;
      if (mine > thr) {
        // a large bucket becomes ⌈count / MSM_LARGE_CHUNK⌉ work items, each summed by one workgroup
        const uint32_t li = atomicAdd(n_large, 1u);
        const uint32_t nch = (mine + MSM_LARGE_CHUNK - 1) / MSM_LARGE_CHUNK;
        const uint32_t firsti = atomicAdd(n_large + 2, nch);
        large_list[li] = b0 + threadIdx.x;
        large_first[li] = firsti;
        for (uint32_t q = 0; q < nch && firsti + q < item_cap; q++) large_items[firsti + q] = make_uint2(b0 + threadIdx.x, q);
      }
    }
  }
  __syncthreads();
  const uint32_t stride = blockDim.x * gridDim.y, first = start + blockIdx.y * blockDim.x + threadIdx.x;
  if (first - threadIdx.x >= end) return; // nothing for this workgroup (uniform across the workgroup)
  for (uint32_t e = first; e < end; e += stride) atomicAdd(&hist[(tmp[e] >> fs) & fmask], 1u);
  __syncthreads();
  if (threadIdx.x < nbk) {
    const uint32_t h = hist[threadIdx.x];
    cur[threadIdx.x] += h ? atomicAdd(&cursor[b0 + threadIdx.x], h) : 0;
  }
  __syncthreads();
  for (uint32_t e = first; e < end; e += stride) {
    const uint32_t v = tmp[e];
    const uint32_t pos = atomicAdd(&cur[(v >> fs) & fmask], 1u);
    sorted[pos] = v & ~(fmask << fs);
  }
}


// ---- LDS-staged two-pass sort (table mode, large sets) ------------------------------------------------------------------
// PMC of the two-level scatter above on MI355X (profiles/r02_pmc_*): the partition pass WRITES 640 MB for 83 MB of entries
// and the place pass 197 MB — a store instruction costs one 32-byte fabric write per sector it touches and the L2 does not
// merge the 4-byte stores of different instructions, so entries must leave a workgroup as runs of consecutive lanes.  Here
// both passes sort their tile in LDS first and copy it out run by run:
//  A  a tile of 1024 scalars counts its digits per PARTITION (top bits of the bucket; sort2_tile_hist), a column scan over the
//     tiles gives every (tile, partition) run its place (no atomics, deterministic), and the tile — ALL its windows, one recode —
//     is staged in LDS sorted by partition and written as ≤ P runs of ≈ 13 entries (sort2_tile_partition).  Entries carry the
//     scalar's index WITHIN THE TILE (10 bits), which leaves room for up to 17 low bucket bits in 32 bits;
//  B  partitions are cut into chunks of 12 K entries; a chunk is counting-sorted by its low bucket bits in LDS and written out as
//     runs of ≈ 23 entries per bucket.  A chunk is walked ROW BY ROW — the runs of the tiles that fed the partition, in tile
//     order, whose first entries are the partition's column of `off` — so the tile of an entry (needed to restore the global
//     scalar index) is known from the row being walked.  Per-bucket counts and offsets fall out of the chunk histograms.
// Round 5: every scalar is read and recoded TWICE per sort (histogram, placement) instead of three times (the tile used to be
// staged in two window groups), and no entry is located by binary search any more — rounds 2–4 searched the partition of every
// entry in pass A's copy-out (10 steps), its tile (11 steps) and its bucket (9 steps) in pass B: 138 / 205 VALU instructions per
// entry in the two big kernels (profiles/r04_pmc_kernels_1600k.txt).  The copy-outs are now run-cooperative: 16 lanes take one
// run (of a partition, a row, a bucket), so the run is known by construction and consecutive lanes still write consecutive
// words; runs longer than S2_LONG entries (the 0/1 wires of a real witness all land in one bucket) are left to the whole
// workgroup.  Launches per sort: 14 → 9 (zeroing folded into the histogram, the column scan's middle kernel into its first,
// the size order's histogram into the bucket scan, its three-kernel scan into one).
// HBM bytes: scalars 2 × 32·L, entries 4·L·W written and read twice, 4·L·W written — each once, in full lines.
constexpr int S2_TILE = 1024;    // scalars per tile (pass A): a row = (tile, all W ≤ 16 windows) stages ≤ 64 KiB of entries in LDS
constexpr int S2_CHUNK = 12288;  // entries per chunk (pass B): 48 KiB of LDS
constexpr int S2_LI_BITS = 10, S2_W_SHIFT = 10, S2_LOW_SHIFT = 14;
// Workgroup size of the three big kernels (partition, chunk histogram, placement): 512 threads when the sort has the GPU to itself
// or shares it with the transforms, 256 (`crowded`) when it runs beside the bucket accumulations — a 512-thread workgroup needs two
// free wave slots with 64 registers each on ALL FOUR SIMDs of a CU at once, and beside three 136-register accumulation waves per
// SIMD (104 registers left) that happens only when waves retire in step: H's sort took 9 ms beside the witness accumulations
// (0.45 ms alone; its partition pass alone 4.9–5.3 ms) and had become what H's accumulation waits for.  One wave per SIMD fits the
// gap: 2.9 ms (profiles/r05_ab_sort_wg256.txt).  The witness sort keeps 512: beside the transforms the smaller groups slow the
// front end by 0.14 ms.
constexpr int S2_THREADS_WIDE = 512, S2_THREADS_CROWDED = 256;
constexpr int S2_RG = 32;        // row groups of the column scan
constexpr int S2_GRP = 16;       // lanes that copy one run
constexpr uint32_t S2_LONG = 192; // a run above this is copied by the whole workgroup
constexpr size_t S2_LDS_MAX = 76 * 1024; // two workgroups per CU; ≤ 66 KiB at the benchmark sizes (room next to one NTT workgroup)

// largest i < n with a[i] <= x (a non-decreasing, a[0] <= x)
__device__ __forceinline__ uint32_t s2_upper(const uint32_t* a, uint32_t n, uint32_t x)
{
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] <= x) lo = mid;
    else hi = mid;
  }
  return lo;
}
// exclusive scan of n ≤ 64·per LDS counters by wave 0 (in → out, out[n] = total); the other waves wait at the caller's barrier
__device__ __forceinline__ void s2_wave_scan(const uint32_t* in, uint32_t* out, uint32_t n)
{
  if (threadIdx.x >= 64) return;
  const uint32_t per = (n + 63) / 64, lo = threadIdx.x * per;
  uint32_t s = 0;
  for (uint32_t k = 0; k < per; k++)
    if (lo + k < n) s += in[lo + k];
  uint32_t incl = s;
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const uint32_t o = __shfl_up(incl, dlt, 64);
    if ((int)threadIdx.x >= dlt) incl += o;
  }
  uint32_t run = incl - s;
  for (uint32_t k = 0; k < per; k++)
    if (lo + k < n) {
      const uint32_t v = in[lo + k];
      out[lo + k] = run;
      run += v;
    }
  if (threadIdx.x == 63) out[n] = incl;
}
// Copy-out of `nruns` runs staged back to back in LDS (`start[k]` … `start[k + 1]` = run k in src; delta[k] = its place in dst minus
// start[k]): S2_GRP lanes per run, so that the run is known without a search and consecutive lanes write consecutive words;
// a run above S2_LONG entries is listed and copied by all threads afterwards.  `nlong` / `longs`: LDS scratch (counter zeroed by the
// caller before the barrier that precedes the call; ≤ S2_CHUNK / S2_LONG entries).
__device__ __forceinline__ void s2_copy_runs(const uint32_t* src, const uint32_t* start, const uint32_t* delta, uint32_t nruns, uint32_t* __restrict__ dst, uint32_t* nlong, uint32_t* longs)
{
  const uint32_t grp = threadIdx.x / S2_GRP, l = threadIdx.x % S2_GRP, ngrp = blockDim.x / S2_GRP;
  for (uint32_t k = grp; k < nruns; k += ngrp) {
    const uint32_t b0 = start[k], b1 = start[k + 1], d = delta[k];
    if (b1 - b0 > S2_LONG) {
      if (l == 0) longs[atomicAdd(nlong, 1u)] = k;
      continue;
    }
    for (uint32_t e = b0 + l; e < b1; e += S2_GRP) dst[e + d] = src[e];
  }
  __syncthreads();
  const uint32_t nl = *nlong;
  for (uint32_t j = 0; j < nl; j++) {
    const uint32_t k = longs[j], b0 = start[k], b1 = start[k + 1], d = delta[k];
    for (uint32_t e = b0 + threadIdx.x; e < b1; e += blockDim.x) dst[e + d] = src[e];
  }
}
constexpr uint32_t S2_MAXLONG = 1024 * 16 / S2_LONG + 2; // runs above S2_LONG that 16 K staged entries can hold

// pass A, step 1: digits of a tile counted per partition: cnt[tile·P + p].  Workgroup 0 also zeroes the plan's counters
// (n_large | tickets | the column scan's ticket): one launch less on a latency chain.
__global__ __launch_bounds__(256) void sort2_tile_hist_kernel(const fe* __restrict__ scalars, uint32_t L, MsmGeom g, int mont, int low_b, uint32_t P, uint32_t* __restrict__ cnt,
                                                              uint32_t* __restrict__ zero_p, uint32_t zero_n)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  extern __shared__ uint32_t sh[]; // [P]
  if (blockIdx.x == 0)
    for (uint32_t k = threadIdx.x; k < zero_n; k += 256) zero_p[k] = 0;
  for (uint32_t p = threadIdx.x; p < P; p += 256) sh[p] = 0;
  __syncthreads();
  const uint32_t first = blockIdx.x * S2_TILE;
  for (int u = 0; u < S2_TILE / 256; u++) {
    const uint32_t i = first + u * 256 + threadIdx.x;
    if (i < L) {
      uint32_t t[9], neg;
      recode(scalars, i, g, mont, t, neg);
      for (int w = 0; w < g.W; w++) {
        const uint32_t d = digit(t, w, g);
        if (d) atomicAdd(&sh[((d & 0x7fffffffu) - 1) >> low_b], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < P; p += 256) cnt[(size_t)blockIdx.x * P + p] = sh[p];
}
// column scan of cnt[R][P] over the rows, two short kernels: (1) sums per row group — and, in the workgroup that finishes last,
// the scan of the group sums, the partition starts and the chunk numbering — (2) running offsets written back.  Every load and
// store coalesced over p.
__global__ __launch_bounds__(256) void sort2_col_sum_kernel(const uint32_t* __restrict__ cnt, uint32_t R, uint32_t P, uint32_t* __restrict__ partial, uint32_t* __restrict__ ticket,
                                                            uint32_t* __restrict__ part_start, uint32_t* __restrict__ chunk_first)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t ps[4097], cf[4097]; // totals / chunk counts, scanned in place (last workgroup only)
  __shared__ uint32_t s_last;
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p < P) {
    const uint32_t rpg = (R + S2_RG - 1) / S2_RG, r0 = blockIdx.y * rpg, r1 = r0 + rpg < R ? r0 + rpg : R;
    uint32_t s = 0;
    for (uint32_t r = r0; r < r1; r++) s += cnt[(size_t)r * P + p];
    partial[(size_t)blockIdx.y * P + p] = s;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  for (uint32_t q = threadIdx.x; q < P; q += 256) {
    uint32_t t[S2_RG]; // all loads first: a load behind each store of the running sum would make this a chain of 32 round trips
#pragma unroll
    for (int gidx = 0; gidx < S2_RG; gidx++) t[gidx] = partial[(size_t)gidx * P + q];
    uint32_t run = 0;
#pragma unroll
    for (int gidx = 0; gidx < S2_RG; gidx++) {
      partial[(size_t)gidx * P + q] = run;
      run += t[gidx];
    }
    ps[q] = run;
    cf[q] = (run + S2_CHUNK - 1) / S2_CHUNK;
  }
  __syncthreads();
  s2_wave_scan(ps, ps, P);
  __syncthreads();
  s2_wave_scan(cf, cf, P);
  __syncthreads();
  for (uint32_t q = threadIdx.x; q <= P; q += 256) {
    part_start[q] = ps[q];
    chunk_first[q] = cf[q];
  }
}
// (out of place: the counts stay — the partition pass loads its row of them instead of recoding its scalars once more to count)
__global__ __launch_bounds__(256) void sort2_col_apply_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off, uint32_t R, uint32_t P, const uint32_t* __restrict__ partial)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const uint32_t rpg = (R + S2_RG - 1) / S2_RG, r0 = blockIdx.y * rpg, r1 = r0 + rpg < R ? r0 + rpg : R;
  uint32_t run = partial[(size_t)blockIdx.y * P + p];
  for (uint32_t r = r0; r < r1; r++) {
    off[(size_t)r * P + p] = run;
    run += cnt[(size_t)r * P + p];
  }
}
// pass A, step 2: the tile's entries — every window, one recode — sorted by partition in LDS, then copied out run by run
template <int S2_THREADS>
__global__ __launch_bounds__(S2_THREADS) void sort2_tile_partition_kernel(const fe* __restrict__ scalars, uint32_t L, MsmGeom g, int mont, int low_b, uint32_t P,
                                                                           const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, const uint32_t* __restrict__ part_start,
                                                                           uint32_t* __restrict__ tmp)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  extern __shared__ uint32_t sh[];
  __shared__ uint32_t nlong, longs[S2_MAXLONG];
  uint32_t* base = sh;          // [P + 1] exclusive prefix of the row's partition counts
  uint32_t* cur = sh + P + 1;   // [P] counts / rank counters, then the global position minus the local one
  uint32_t* buf = cur + P;      // [S2_TILE · W]
  const uint32_t row = blockIdx.x, first = row * S2_TILE;
  const uint32_t low_mask = (1u << low_b) - 1;
  // the row's digit counts per partition: what sort2_tile_hist_kernel counted for it
  for (uint32_t p = threadIdx.x; p < P; p += S2_THREADS) cur[p] = cnt[(size_t)row * P + p];
  if (threadIdx.x == 0) nlong = 0;
  __syncthreads();
  s2_wave_scan(cur, base, P);
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < P; p += S2_THREADS) cur[p] = 0;
  __syncthreads();
  for (int u = 0; u < S2_TILE / S2_THREADS; u++) {
    const uint32_t li = u * S2_THREADS + threadIdx.x, i = first + li;
    if (i < L) {
      uint32_t t[9], neg;
      recode(scalars, i, g, mont, t, neg);
      for (int w = 0; w < g.W; w++) {
        const uint32_t d = digit(t, w, g);
        if (d) {
          const uint32_t bk = (d & 0x7fffffffu) - 1, p = bk >> low_b;
          const uint32_t r = atomicAdd(&cur[p], 1u);
          buf[base[p] + r] = li | ((uint32_t)w << S2_W_SHIFT) | ((bk & low_mask) << S2_LOW_SHIFT) | (((d >> 31) ^ neg) << 31);
        }
      }
    }
  }
  __syncthreads();
  // cur[p] ← where the run of partition p starts in tmp, minus its start in buf
  for (uint32_t p = threadIdx.x; p < P; p += S2_THREADS) cur[p] = part_start[p] + off[(size_t)row * P + p] - base[p];
  __syncthreads();
  s2_copy_runs(buf, base, cur, P, tmp, &nlong, longs);
}
// the chunk a workgroup owns: partition, first entry, number of entries (false: beyond the last chunk)
__device__ __forceinline__ bool s2_chunk_of(uint32_t c, const uint32_t* __restrict__ part_start, const uint32_t* __restrict__ chunk_first, uint32_t P, uint32_t& p, uint32_t& start, uint32_t& n)
{
  __shared__ uint32_t s_p;
  if (c >= chunk_first[P]) return false;
  if (threadIdx.x == 0) s_p = s2_upper(chunk_first, P, c); // partitions without chunks share a value with their successor: the largest index wins, a non-empty one
  __syncthreads();
  p = s_p;
  const uint32_t k = c - chunk_first[p];
  start = part_start[p] + k * S2_CHUNK;
  const uint32_t end = part_start[p + 1];
  n = end - start < (uint32_t)S2_CHUNK ? end - start : (uint32_t)S2_CHUNK;
  return true;
}
// pass B, step 1: histogram of a chunk over the partition's 2^low_b buckets
template <int S2_THREADS>
__global__ __launch_bounds__(S2_THREADS) void sort2_chunk_hist_kernel(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ part_start, const uint32_t* __restrict__ chunk_first, uint32_t P, int low_b,
                                                                       uint32_t* __restrict__ chunk_hist)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t hist[1024];
  uint32_t p, start, n;
  if (!s2_chunk_of(blockIdx.x, part_start, chunk_first, P, p, start, n)) return;
  const uint32_t NL = 1u << low_b;
  for (uint32_t b = threadIdx.x; b < NL; b += S2_THREADS) hist[b] = 0;
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < n; q += S2_THREADS) atomicAdd(&hist[(tmp[start + q] >> S2_LOW_SHIFT) & (NL - 1)], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < NL; b += S2_THREADS) chunk_hist[(size_t)blockIdx.x * NL + b] = hist[b];
}
// pass B, step 2: one workgroup per partition, one thread per bucket: counts, offsets, the chunk's place inside each bucket,
// large buckets registered (as msm_fine_place_kernel does) — and the partition's histogram of bucket SIZES for the order by
// decreasing size (blockhist[key·P + p]: what msm_order_hist_kernel computes per 256 buckets in the other paths)
__global__ __launch_bounds__(1024) void sort2_bucket_scan_kernel(const uint32_t* __restrict__ chunk_hist, uint32_t* __restrict__ chunk_off, const uint32_t* __restrict__ part_start,
                                                                  const uint32_t* __restrict__ chunk_first, int low_b, uint32_t thr, uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets,
                                                                  uint32_t* __restrict__ n_large, uint32_t* __restrict__ large_list, uint32_t* __restrict__ large_first, uint2* __restrict__ large_items,
                                                                  uint32_t item_cap, uint32_t* __restrict__ blockhist)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t sc[1024], oh[ORDER_BINS];
  const uint32_t NL = 1u << low_b, p = blockIdx.x, b = threadIdx.x, P = gridDim.x;
  for (uint32_t k = b; k < (uint32_t)ORDER_BINS; k += NL) oh[k] = 0;
  const uint32_t c0 = chunk_first[p], c1 = chunk_first[p + 1];
  uint32_t run = 0;
  for (uint32_t c = c0; c < c1; c++) {
    const uint32_t h = chunk_hist[(size_t)c * NL + b];
    chunk_off[(size_t)c * NL + b] = run;
    run += h;
  }
  sc[b] = run;
  __syncthreads();
  atomicAdd(&oh[order_key(run)], 1u);
  for (uint32_t d = 1; d < NL; d <<= 1) { // inclusive scan over the partition's buckets
    const uint32_t x = b >= d ? sc[b - d] : 0;
    __syncthreads();
    sc[b] += x;
    __syncthreads();
  }
  __syncthreads();
  for (uint32_t k = b; k < (uint32_t)ORDER_BINS; k += NL) blockhist[(size_t)k * P + p] = oh[k];
  const uint32_t bucket = (p << low_b) + b;
  counts[bucket] = run;
  offsets[bucket] = part_start[p] + sc[b] - run;
  if (run > thr) {
    const uint32_t li = atomicAdd(n_large, 1u);
    const uint32_t nch = (run + MSM_LARGE_CHUNK - 1) / MSM_LARGE_CHUNK;
    const uint32_t firsti = atomicAdd(n_large + 2, nch);
    large_list[li] = bucket;
    large_first[li] = firsti;
    for (uint32_t q = 0; q < nch && firsti + q < item_cap; q++) large_items[firsti + q] = make_uint2(bucket, q);
  }
}
// pass B, step 3: the chunk counting-sorted by bucket in LDS (entries rewritten with the global scalar index), copied out run
// by run.  The chunk is walked row by row: col[r] = first entry of row r's run inside the partition (the partition's column of
// `off`), 16 lanes per row — the tile of an entry is the row it is read from.
template <int S2_THREADS>
__global__ __launch_bounds__(S2_THREADS) void sort2_chunk_place_kernel(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ part_start, const uint32_t* __restrict__ chunk_first, uint32_t P, int low_b,
                                                                        const uint32_t* __restrict__ chunk_hist, const uint32_t* __restrict__ chunk_off, const uint32_t* __restrict__ offsets,
                                                                        const uint32_t* __restrict__ off, uint32_t R, MsmGeom g, uint32_t* __restrict__ sorted)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  extern __shared__ uint32_t sh[];
  __shared__ uint32_t nlong, longs[S2_MAXLONG], s_rows[2];
  const uint32_t NL = 1u << low_b;
  uint32_t* lbase = sh;             // [NL + 1]
  uint32_t* lcur = sh + NL + 1;     // [NL] counts / rank counters, then global position minus local one
  uint32_t* col = lcur + NL;        // [R + 1] first entry of every row's run inside the partition
  uint32_t* buf = col + R + 1;      // [S2_CHUNK]
  uint32_t p, start, n;
  if (!s2_chunk_of(blockIdx.x, part_start, chunk_first, P, p, start, n)) return;
  const uint32_t pstart = part_start[p], in_part = start - pstart; // position of the chunk inside its partition
  for (uint32_t r = threadIdx.x; r <= R; r += S2_THREADS) col[r] = r < R ? off[(size_t)r * P + p] : 0xffffffffu;
  for (uint32_t b = threadIdx.x; b < NL; b += S2_THREADS) lcur[b] = chunk_hist[(size_t)blockIdx.x * NL + b];
  if (threadIdx.x == 0) nlong = 0;
  __syncthreads();
  s2_wave_scan(lcur, lbase, NL);
  // rows the chunk touches (empty rows share their neighbour's value: the largest index wins, which is a row that holds the entry)
  if (threadIdx.x == 64) s_rows[0] = s2_upper(col, R, in_part);
  if (threadIdx.x == 128) s_rows[1] = s2_upper(col, R, in_part + n - 1);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < NL; b += S2_THREADS) lcur[b] = 0;
  __syncthreads();
  {
    const uint32_t r_first = s_rows[0], r_last = s_rows[1], c_end = in_part + n;
    const uint32_t grp = threadIdx.x / S2_GRP, l = threadIdx.x % S2_GRP, ngrp = S2_THREADS / S2_GRP;
    auto place = [&](uint32_t r, uint32_t q) {
      const uint32_t v = tmp[pstart + q];
      const uint32_t b = (v >> S2_LOW_SHIFT) & (NL - 1);
      const uint32_t i = r * S2_TILE + (v & ((1u << S2_LI_BITS) - 1));
      const uint32_t w = (v >> S2_W_SHIFT) & 15u;
      const uint32_t rk = atomicAdd(&lcur[b], 1u);
      buf[lbase[b] + rk] = entry_idx(g, (int)w, i) | (v & 0x80000000u);
    };
    for (uint32_t r = r_first + grp; r <= r_last; r += ngrp) {
      const uint32_t lo = col[r] > in_part ? col[r] : in_part, hi = col[r + 1] < c_end ? col[r + 1] : c_end;
      if (hi > lo && hi - lo > S2_LONG) { // (a tile of 0/1 wires: hundreds of entries of one row in one partition)
        if (l == 0) longs[atomicAdd(&nlong, 1u)] = r;
        continue;
      }
      for (uint32_t q = lo + l; q < hi; q += S2_GRP) place(r, q);
    }
    __syncthreads();
    const uint32_t nl = nlong;
    for (uint32_t j = 0; j < nl; j++) {
      const uint32_t r = longs[j];
      const uint32_t lo = col[r] > in_part ? col[r] : in_part, hi = col[r + 1] < c_end ? col[r + 1] : c_end;
      for (uint32_t q = lo + threadIdx.x; q < hi; q += S2_THREADS) place(r, q);
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < NL; b += S2_THREADS) lcur[b] = offsets[(p << low_b) + b] + chunk_off[(size_t)blockIdx.x * NL + b] - lbase[b];
  if (threadIdx.x == 0) nlong = 0;
  __syncthreads();
  s2_copy_runs(buf, lbase, lcur, NL, sorted, &nlong, longs);
}
// ---- order by decreasing size for the LDS-staged path: blockhist[key][P] (sort2_bucket_scan_kernel) → per key the exclusive
// prefix over the partitions (in place), and in the workgroup that finishes last the exclusive prefix of the key totals
__global__ __launch_bounds__(256) void sort2_order_scan_kernel(uint32_t* __restrict__ blockhist, uint32_t P, uint32_t* __restrict__ keytot, uint32_t* __restrict__ ticket)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t sh[SCAN_T];
  __shared__ uint32_t s_last;
  const uint32_t key = blockIdx.x, per = (P + 255) / 256, lo = threadIdx.x * per;
  uint32_t* row = blockhist + (size_t)key * P;
  uint32_t s = 0;
  for (uint32_t k = 0; k < per; k++)
    if (lo + k < P) s += row[lo + k];
  uint32_t total;
  uint32_t run = block_exclusive_scan(s, sh, &total);
  for (uint32_t k = 0; k < per; k++)
    if (lo + k < P) {
      const uint32_t v = row[lo + k];
      row[lo + k] = run;
      run += v;
    }
  if (threadIdx.x == 0) keytot[key] = total;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const uint32_t v = threadIdx.x < gridDim.x ? keytot[threadIdx.x] : 0; // ORDER_BINS = 256 keys = one per thread
  const uint32_t ex = block_exclusive_scan(v, sh, nullptr);
  if (threadIdx.x < gridDim.x) keytot[threadIdx.x] = ex;
}
__global__ __launch_bounds__(1024) void sort2_order_scatter_kernel(const uint32_t* __restrict__ counts, int low_b, const uint32_t* __restrict__ scanned, const uint32_t* __restrict__ keybase,
                                                                   uint32_t* __restrict__ order)
{
  ISNARK_CRITICAL_CHAIN_KERNEL();
  __shared__ uint32_t h[ORDER_BINS];
  const uint32_t NL = 1u << low_b, p = blockIdx.x, P = gridDim.x;
  for (uint32_t k = threadIdx.x; k < (uint32_t)ORDER_BINS; k += NL) h[k] = 0;
  __syncthreads();
  const uint32_t b = (p << low_b) + threadIdx.x;
  const uint32_t key = order_key(counts[b]);
  const uint32_t r = atomicAdd(&h[key], 1u);
  order[keybase[key] + scanned[(size_t)key * P + p] + r] = b;
}

int ilog2_ceil(uint64_t x)
{
  int l = 0;
  while ((1ull << l) < x) l++;
  return l;
}

} // namespace

// Table mode: low bucket bits carried in the 32-bit entries of the two-level sort (partitions = buckets >> low).  8192
// partitions (64 KiB of LDS cursors in the partition pass) as a rule; 16384 (128 KiB, one workgroup per CU) when only
// that lets the entry — point index | window | low bucket bits | sign — fit: 3.2 M constraints keep c = 20 / 13 digits
// instead of c = 19 / 14.  −1: does not fit.
static int tab_low_bits(int c, int ib, int W)
{
  for (int pb = 13; pb <= 14; pb++) {
    int low = (c - 1) - pb;
    if (low < 0) low = 0;
    if (low > 7) continue;
    if (ib + ilog2_ceil((uint64_t)W) + low <= 31) return low;
  }
  return -1;
}

MsmGeom msm_geometry(uint32_t L, int c_cfg, int tab, int bits, int pf)
{
  if (bits <= 0 || bits > 254) bits = 254;
  if (pf < 1) pf = 1;
  if (bits != 254 || pf > 1) tab = 0; // the table mode belongs to the prover's cached keys: full-width scalars, own tables
  // window size: as the reference, ≈ log2(L) − 4 (cuda_msm.cuh:45-48), capped so that bucket magnitudes fit
  // 15 bits + sign
  MsmGeom g;
  memset(&g, 0, sizeof g);
  int c = c_cfg > 0 ? c_cfg : ilog2_ceil(L ? L : 1) - 4;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  if (tab) {
    // one bucket set: as many buckets as the classic layout has over all its windows (≈ 2^(c+3)) → digits up to 4 bits
    // wider; the widest digit whose entry (point index | window | low bucket bits | sign) still fits 32 bits is taken
    const int ib = ilog2_ceil(L ? L : 1);
    int ct = 0;
    if (c_cfg <= 0) {
      for (int t = c + 4 > 20 ? 20 : c + 4; t > c; t--) {
        if (tab_low_bits(t, ib, 254 / t + 1) >= 0) {
          ct = t;
          break;
        }
      }
      // same number of digits with a narrower digit: fewer buckets, and the top digit keeps enough bits to spread over
      // many buckets (c = 18 leaves it 2 bits — three buckets then hold a quarter of all entries each; c = 17 leaves 16)
      while (ct > c + 1 && 254 / (ct - 1) + 1 == 254 / ct + 1) ct--;
      // tab > 1: table mode with THIS digit width (a base subset that keeps the geometry of the full set, prover.cpp: sparse B)
      if (tab > 1 && tab <= 20 && tab_low_bits(tab, ib, 254 / tab + 1) >= 0) ct = tab;
    }
    if (ct) c = ct;
    else tab = 0;
    g.IB = ib;
  }
  g.tab = tab ? 1 : 0;
  g.c = c;
  g.W = bits / c + 1; // the top window holds the remaining bits + the carry of the signed recoding
  g.NB = 1u << (c - 1);
  g.pf = g.tab ? 1 : pf;
  g.nbms = (g.W + g.pf - 1) / g.pf;
  if (g.tab) {
    g.NBb = g.NB > 32768u ? 32768u : g.NB;
    g.Wb = (int)(g.NB / g.NBb);
  } else {
    g.NBb = g.NB;
    g.Wb = g.nbms;
  }
  g.wide = g.W;
  {
    // Table mode: always (one shared bucket set: nothing is lost).  Classic layout (one bucket set per window): a narrower
    // window leaves half of its buckets empty, so only when the top digit would be 1–3 bits short — then its entries sit in
    // 1/2 … 1/8 of a window's buckets, too few per bucket for the large-bucket path and several times the average
    // (c = 16: a 14-bit top digit); a top digit 4 or more bits short is handled by the large-bucket kernels.
    const int spare = g.W * c - 254; // ≥ 0: bits the W windows cover beyond the 254 of a scalar
    // (only for full-width scalars — the argument rests on the recoded value being ≤ (r − 1)/2 — and equal windows are what
    //  the shift c·nbms of precomputed bases assumes)
    if (bits == 254 && g.pf == 1 && c >= 5 && (g.tab || spare <= 3)) g.wide = g.W - (spare < g.W ? spare : g.W); // (a 3-bit top window would have no room for offset + carry)
  }
  uint32_t H[10] = {0};
  for (int w = 0; w < g.W; w++) {
    const int bit = (w < g.wide ? w * c : g.wide * c + (w - g.wide) * (c - 1)) + (w < g.wide ? c : c - 1) - 1;
    H[bit >> 5] |= 1u << (bit & 31);
  }
  memcpy(g.H, H, sizeof g.H);
  return g;
}

eIcicleError msm_sort_run(const fe* d_scalars, uint32_t L, int c_cfg, int lbf, int mont_sc, hipStream_t s, SortPlan* pl, int tab, int bits, int pf, uint64_t entries_hint, bool crowded)
{
  pl->g = msm_geometry(L, c_cfg, tab, bits, pf);
  const MsmGeom& g = pl->g;
  pl->L = L;
  pl->stream = s;
  const uint32_t nb = g.NBb * (uint32_t)g.Wb;
  pl->nbuckets = nb;
  // Large-bucket threshold.  One thread sums one bucket, so the accumulation kernel lasts as long as its longest chain: buckets
  // with more than `thr` entries go to the workgroup-per-chunk kernels instead.  The reference takes large_bucket_factor (10) ×
  // the average (cuda_msm.cuh:205-220); here the default is 3 × the average with a floor of 64 — uniform scalars have no bucket
  // beyond 2 × the average, while witnesses of real circuits (0/1 wires, bytes, small packed values) put hundreds of entries
  // into the low buckets of the first window: with 10 × / floor 512 the synthetic stand-in circuits spent 2 ms (G1) / 6.5 ms (G2)
  // in 500-addition chains of single threads (prove 11.2 → 6.3 ms at 1.0 M constraints, 14.1 → 8.3 ms at 1.4 M; 1.6 M uniform:
  // unchanged).  The caller's large_bucket_factor (ConfigExtension) is honoured when given.
  constexpr uint32_t large_floor = 64u;
  const uint64_t avg = (entries_hint ? entries_hint : g.tab ? (uint64_t)L * g.W : (uint64_t)L * g.pf) / g.NB + 1;
  uint32_t thr = (uint32_t)(avg * (uint64_t)(lbf > 0 ? lbf : 3));
  if (thr < large_floor) thr = large_floor;
  if (thr < 8) thr = 8;
  pl->large_thr = thr;
  const uint32_t nblocks = (nb + SCAN_B - 1) / SCAN_B;
  const uint64_t nentries = (uint64_t)L * g.W;
  // bucket index = partition | low bits (≤ 128 buckets per partition; ≤ 8192 partitions: two u32 per partition in LDS)
  int low_bits = (g.c - 1) > 8 ? (g.c - 1) - 8 : 0;
  if (g.tab) {
    low_bits = tab_low_bits(g.c, g.IB, g.W);
    if (low_bits < 0) low_bits = 7; // forced c that does not fit: the two_level test below fails and the global-histogram path runs
  }
  const uint32_t NP = g.NBb >> low_bits, nparts = nb >> low_bits;
  const int idx_bits = g.tab ? g.IB + ilog2_ceil((uint64_t)g.W) : ilog2_ceil((L ? (uint64_t)L : 1) * g.pf);
  if (idx_bits > 31) {
    set_last_error("msm: %u scalars x precompute_factor %d exceed the 31-bit point index of a sort entry", L, g.pf);
    return ICICLE_INVALID_ARGUMENT;
  }
  const bool two_level = idx_bits + low_bits <= 31 && low_bits <= 7 && (size_t)nparts * 8 <= 128 * 1024;
  const uint32_t oblk = (nb + ORDER_BINS - 1) / ORDER_BINS;           // workgroups of the size-order pass
  const uint32_t om = oblk * ORDER_BINS, oscan = (om + SCAN_B - 1) / SCAN_B;
  // work items of large buckets: ≤ entries/CHUNK full chunks + one partial chunk per large bucket (≤ entries/thr of those)
  pl->item_cap = (uint32_t)(nentries / MSM_LARGE_CHUNK + nentries / thr + 2);
  // layout: counts | offsets | cursor | large_list | order | large_first | n_large[4] | tickets[TK] | bsum[nblocks] | part_count[nparts] | part_start[nparts+1] |
  //         part_cursor[nparts] | blockhist[om] | obsum[oscan] | large_items[2·item_cap]
  constexpr uint32_t TK = MSM_TICKET_SLOTS * 64;
  constexpr uint32_t S2TK = 4; // tickets of the LDS-staged path's own last-workgroup steps (column scan, order scan)
  const size_t head = (size_t)nb * 6 + 4 + TK + S2TK + nblocks + 3 * (size_t)nparts + 1 + om + oscan;
  HIP_TRY(ws_alloc((void**)&pl->ws, (head + 2 * (size_t)pl->item_cap + 2) * 4, s), ICICLE_ALLOCATION_FAILED);
  pl->counts = pl->ws;
  pl->offsets = pl->counts + nb;
  uint32_t* cursor = pl->offsets + nb;
  pl->large_list = cursor + nb;
  pl->order = pl->large_list + nb;
  pl->large_first = pl->order + nb;
  pl->n_large = pl->large_first + nb;
  pl->tickets = pl->n_large + 4;
  uint32_t* s2tickets = pl->tickets + TK;
  uint32_t* bsum = s2tickets + S2TK;
  uint32_t* part_count = bsum + nblocks;
  uint32_t* part_start = part_count + nparts;
  uint32_t* part_cursor = part_start + nparts + 1;
  uint32_t* blockhist = part_cursor + nparts;
  uint32_t* obsum = blockhist + om;
  pl->large_items = reinterpret_cast<uint2*>(obsum + oscan + (head & 1)); // 8-byte aligned
  HIP_TRY(ws_alloc((void**)&pl->sorted, (size_t)(nentries ? nentries : 1) * 4, s), ICICLE_ALLOCATION_FAILED);
  uint32_t* tmp = nullptr;
  WsScoped<uint32_t> tmp_block;
  if (two_level) {
    HIP_TRY(tmp_block.alloc((size_t)nentries, s), ICICLE_ALLOCATION_FAILED);
    tmp = tmp_block;
  }

  // LDS-staged path (table mode, ≥ 2^17 entries): partitions of ≤ 28 K entries, ≤ 1024 buckets each
  bool lds_sort = false;
  int s2_pb = 0, s2_low = 0;
  uint32_t s2_ntiles = 0, s2_maxchunks = 0;
  size_t s2_lds_a = 0, s2_lds_b = 0;
  if (g.tab && nentries >= (1u << 17) && g.W <= 16) { // (2^20 until round 5: with nine launches the staged path is no slower than the two-level one down to the witness heads and 8-way shards)
    while ((nentries >> s2_pb) > 28000) s2_pb++;
    s2_low = (g.c - 1) - s2_pb;
    if (s2_low > 10) {
      s2_pb += s2_low - 10;
      s2_low = 10;
    }
    s2_ntiles = (L + S2_TILE - 1) / S2_TILE;
    s2_maxchunks = (uint32_t)(nentries / S2_CHUNK) + (1u << s2_pb);
    s2_lds_a = ((size_t)2 * (1u << s2_pb) + 1 + (size_t)S2_TILE * g.W) * 4;
    s2_lds_b = ((size_t)2 * (1u << (s2_low > 0 ? s2_low : 0)) + 1 + s2_ntiles + 1 + S2_CHUNK) * 4;
    lds_sort = s2_low >= 0 && s2_pb <= 12 && (1u << s2_pb) * (uint64_t)(1u << s2_low) == nb && s2_lds_a <= S2_LDS_MAX && s2_lds_b <= S2_LDS_MAX && S2_LOW_SHIFT + s2_low <= 31 &&
               (size_t)(1u << s2_pb) * 4 <= 64 * 1024;
  }
  if (lds_sort) {
    const uint32_t P = 1u << s2_pb, NL = 1u << s2_low, R = s2_ntiles;
    // workspace: cnt[R][P] | off[R][P] | partial[S2_RG][P] | part_start[P + 1] | chunk_first[P + 1] | chunk_hist | chunk_off [maxchunks][NL] | order hist [256][P] | keytot[256]
    WsScoped<uint32_t> s2ws;
    const size_t n_cnt = (size_t)R * P, n_part = (size_t)S2_RG * P, n_ch = (size_t)s2_maxchunks * NL, n_bh = (size_t)ORDER_BINS * P;
    HIP_TRY(s2ws.alloc(2 * n_cnt + n_part + 2 * (size_t)P + 2 + 2 * n_ch + n_bh + ORDER_BINS, s), ICICLE_ALLOCATION_FAILED);
    uint32_t* cnt = s2ws.p;
    uint32_t* off = cnt + n_cnt;      // first entry of every (row, partition) run inside its partition
    uint32_t* partial = off + n_cnt;
    uint32_t* pstart = partial + n_part;
    uint32_t* cfirst = pstart + P + 1;
    uint32_t* chist = cfirst + P + 1;
    uint32_t* coff = chist + n_ch;
    uint32_t* s2bh = coff + n_ch;
    uint32_t* keytot = s2bh + n_bh;
    WsScoped<uint32_t> s2tmp_own;
    uint32_t* s2tmp = tmp; // the scatter buffer of the two-level path when that one was set up as well
    if (!s2tmp) {
      HIP_TRY(s2tmp_own.alloc((size_t)nentries, s), ICICLE_ALLOCATION_FAILED);
      s2tmp = s2tmp_own.p;
    }
    static std::atomic<int> attr_dev_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_dev_mask.load() & (1 << (dev & 31)))) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sort2_tile_partition_kernel<S2_THREADS_WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2_LDS_MAX), ICICLE_UNKNOWN_ERROR);
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sort2_chunk_place_kernel<S2_THREADS_WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2_LDS_MAX), ICICLE_UNKNOWN_ERROR);
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sort2_tile_partition_kernel<S2_THREADS_CROWDED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2_LDS_MAX), ICICLE_UNKNOWN_ERROR);
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sort2_chunk_place_kernel<S2_THREADS_CROWDED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2_LDS_MAX), ICICLE_UNKNOWN_ERROR);
      attr_dev_mask.fetch_or(1 << (dev & 31));
    }
    const dim3 cgrid((P + 255) / 256, S2_RG);
    // nine launches: histogram (+ zeroing of n_large | tickets | this path's tickets), column scan (2), partition, chunk histogram,
    // bucket scan (+ histogram of the bucket sizes), placement, size order (2)
    hipLaunchKernelGGL(sort2_tile_hist_kernel, dim3(s2_ntiles), dim3(256), (size_t)P * 4, s, d_scalars, L, g, mont_sc, s2_low, P, cnt, pl->n_large, 4u + TK + S2TK);
    hipLaunchKernelGGL(sort2_col_sum_kernel, cgrid, dim3(256), 0, s, cnt, R, P, partial, s2tickets, pstart, cfirst);
    hipLaunchKernelGGL(sort2_col_apply_kernel, cgrid, dim3(256), 0, s, cnt, off, R, P, partial);
#define ISNARK_S2_BIG(T)                                                                                                                                                                    \
  hipLaunchKernelGGL(sort2_tile_partition_kernel<T>, dim3(R), dim3(T), s2_lds_a, s, d_scalars, L, g, mont_sc, s2_low, P, cnt, off, pstart, s2tmp);                                          \
  hipLaunchKernelGGL(sort2_chunk_hist_kernel<T>, dim3(s2_maxchunks), dim3(T), 0, s, s2tmp, pstart, cfirst, P, s2_low, chist);                                                               \
  hipLaunchKernelGGL(sort2_bucket_scan_kernel, dim3(P), dim3(NL), 0, s, chist, coff, pstart, cfirst, s2_low, thr, pl->counts, pl->offsets, pl->n_large, pl->large_list, pl->large_first,    \
                     pl->large_items, pl->item_cap, s2bh);                                                                                                                                   \
  hipLaunchKernelGGL(sort2_chunk_place_kernel<T>, dim3(s2_maxchunks), dim3(T), s2_lds_b, s, s2tmp, pstart, cfirst, P, s2_low, chist, coff, pl->offsets, off, R, g, pl->sorted)
    if (crowded) {
      ISNARK_S2_BIG(S2_THREADS_CROWDED);
    } else {
      ISNARK_S2_BIG(S2_THREADS_WIDE);
    }
#undef ISNARK_S2_BIG
    hipLaunchKernelGGL(sort2_order_scan_kernel, dim3(ORDER_BINS), dim3(256), 0, s, s2bh, P, keytot, s2tickets + 1);
    hipLaunchKernelGGL(sort2_order_scatter_kernel, dim3(P), dim3(NL), 0, s, pl->counts, s2_low, s2bh, keytot, pl->order);
    return check_launch("msm_sort (LDS-staged)");
  } else {
  unsigned zb = (3 * nb + 255) / 256;
  if (zb > 1024) zb = 1024;
  hipLaunchKernelGGL(msm_zero_kernel, dim3(zb), dim3(256), 0, s, pl->counts, 3 * nb); // counts | offsets | cursor
  hipLaunchKernelGGL(msm_zero_kernel, dim3(8), dim3(256), 0, s, pl->n_large, 4u + TK + S2TK + nblocks + nparts); // n_large | tickets | bsum | part_count
  const unsigned lgrid = (L + 255) / 256;
  if (two_level) {
    // counts and offsets come out of the partitions: no per-digit global atomics
    const unsigned pgrid = (L + PA_SCALARS - 1) / PA_SCALARS;
    if (L) hipLaunchKernelGGL(msm_coarse_hist_kernel, dim3(pgrid), dim3(PA_THREADS), (size_t)nparts * 4, s, d_scalars, L, g, mont_sc, low_bits, nparts, part_count);
    hipLaunchKernelGGL(msm_part_scan_kernel, dim3(1), dim3(SCAN_T), 0, s, part_count, nparts, part_start, part_cursor);
    if (L) {
      if ((size_t)nparts * 8 > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msm_partition_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)nparts * 8));
      hipLaunchKernelGGL(msm_partition_kernel, dim3(pgrid), dim3(PA_THREADS), (size_t)nparts * 8, s, d_scalars, L, g, mont_sc, low_bits, NP, nparts, part_cursor, tmp);
      hipLaunchKernelGGL(msm_fine_count_kernel, dim3(nparts, PB_SPLIT), dim3(256), 0, s, tmp, part_start, low_bits, pl->counts);
    }
    hipLaunchKernelGGL(msm_fine_place_kernel, dim3(nparts, PB_SPLIT), dim3(256), 0, s, tmp, part_start, pl->counts, cursor, low_bits, thr, pl->offsets, pl->n_large, pl->large_list,
                       pl->large_first, pl->large_items, pl->item_cap, pl->sorted);
  } else {
    if (L) hipLaunchKernelGGL(msm_hist_kernel, dim3(lgrid), dim3(256), 0, s, d_scalars, L, g, mont_sc, pl->counts);
    hipLaunchKernelGGL(msm_scan_sums_kernel, dim3(nblocks), dim3(SCAN_T), 0, s, pl->counts, nb, bsum);
    hipLaunchKernelGGL(msm_scan_top_kernel, dim3(1), dim3(SCAN_T), 0, s, bsum, nblocks);
    hipLaunchKernelGGL(msm_scan_finish_kernel, dim3(nblocks), dim3(SCAN_T), 0, s, pl->counts, nb, bsum, pl->offsets, cursor, thr, pl->n_large, pl->large_list, pl->large_first, pl->large_items, pl->item_cap);
    if (L) hipLaunchKernelGGL(msm_scatter_kernel, dim3(lgrid), dim3(256), 0, s, d_scalars, L, g, mont_sc, cursor, pl->sorted);
  }
  }
  // bucket ids by decreasing size
  hipLaunchKernelGGL(msm_order_hist_kernel, dim3(oblk), dim3(ORDER_BINS), 0, s, pl->counts, nb, oblk, blockhist);
  hipLaunchKernelGGL(msm_scan_sums_kernel, dim3(oscan), dim3(SCAN_T), 0, s, blockhist, om, obsum);
  hipLaunchKernelGGL(msm_scan_top_kernel, dim3(1), dim3(SCAN_T), 0, s, obsum, oscan);
  hipLaunchKernelGGL(msm_scan_apply_kernel, dim3(oscan), dim3(SCAN_T), 0, s, blockhist, om, obsum, blockhist);
  hipLaunchKernelGGL(msm_order_scatter_kernel, dim3(oblk), dim3(ORDER_BINS), 0, s, pl->counts, nb, oblk, blockhist, pl->order);
  return check_launch("msm_sort");
}

void msm_sort_release(SortPlan* pl)
{
  if (pl->ws) (void)ws_free(pl->ws, pl->stream);
  if (pl->sorted) (void)ws_free(pl->sorted, pl->stream);
  pl->ws = pl->sorted = nullptr;
}
SortPlan::~SortPlan() { msm_sort_release(this); }

// exclusive prefix sum of m 32-bit counters on stream s (the three kernels above: sums of 2048-counter blocks, their scan by one
// workgroup, the finish): out[i] = Σ_{j<i} in[j].  `bsum`: scratch of ⌈m / 2048⌉ words.  In place (out == in) is fine.  Used by the
// cache build's CSR (prover/csr.hip) as well — no third-party device code on any path of this library.
hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t m, uint32_t* out, uint32_t* bsum, hipStream_t s)
{
  if (!m) return hipSuccess;
  const uint32_t nblocks = (m + SCAN_B - 1) / SCAN_B;
  hipLaunchKernelGGL(msm_scan_sums_kernel, dim3(nblocks), dim3(SCAN_T), 0, s, in, m, bsum);
  hipLaunchKernelGGL(msm_scan_top_kernel, dim3(1), dim3(SCAN_T), 0, s, bsum, nblocks);
  hipLaunchKernelGGL(msm_scan_apply_kernel, dim3(nblocks), dim3(SCAN_T), 0, s, in, m, bsum, out);
  return hipGetLastError();
}
size_t exclusive_scan_u32_scratch_words(uint32_t m) { return (m + SCAN_B - 1) / SCAN_B + 1; }

} // namespace isnark

// Profile of the `back`-th most recent MSM (0 = latest) issued by this process.  The caller must have
// synchronised the MSM's stream.  out_ms = {recode+sort, bucket accumulation kernel, large buckets +
// reduction (+ device tail), total, digit sort alone (0 when this MSM did not issue one: several base sets share one
// sort — the prover records it with the G2 MSM for the witness and with the H MSM)}; geom = {L, nbuckets, c, W, is_g2}.
ISNARK_API eIcicleError icicle_snark_msm_profile(int back, float out_ms[5], uint32_t geom[5])
{
  using namespace isnark;
  std::lock_guard<std::mutex> lk(g_msm_prof_mu);
  const MsmProfileRing& r = msm_ring_of_active_device();
  if (!out_ms || !geom || back < 0 || back >= MSM_PROFILE_RING || (uint64_t)back >= r.seq) return ICICLE_INVALID_ARGUMENT;
  const MsmProfile& p = r.slots[(r.seq - 1 - back) % MSM_PROFILE_RING];
  if (!p.valid) return ICICLE_INVALID_ARGUMENT;
  geom[0] = p.L; geom[1] = p.nbuckets; geom[2] = (uint32_t)p.c; geom[3] = (uint32_t)p.W; geom[4] = (uint32_t)p.is_g2;
  if (p.resolved) {
    memcpy(out_ms, p.ms, sizeof p.ms);
    return ICICLE_SUCCESS;
  }
  if (hipEventElapsedTime(&out_ms[0], p.ev[0], p.ev[1]) != hipSuccess) return ICICLE_UNKNOWN_ERROR;
  (void)hipEventElapsedTime(&out_ms[1], p.ev[1], p.ev[2]);
  (void)hipEventElapsedTime(&out_ms[2], p.ev[2], p.ev[3]);
  (void)hipEventElapsedTime(&out_ms[3], p.ev[0], p.ev[3]);
  out_ms[4] = 0;
  if (p.has_sort_end) (void)hipEventElapsedTime(&out_ms[4], p.ev[0], p.ev[4]);
  geom[0] = p.L; geom[1] = p.nbuckets; geom[2] = (uint32_t)p.c; geom[3] = (uint32_t)p.W; geom[4] = (uint32_t)p.is_g2;
  return ICICLE_SUCCESS;
}

// first launch of a translation unit's code object loads it onto the device (milliseconds): prewarm_modules (runtime.cpp) does that ahead
// of the first prove of a process
namespace isnark {
__global__ void module_warm_sort_kernel() {}
void module_warm_sort(hipStream_t s) { hipLaunchKernelGGL(module_warm_sort_kernel, dim3(1), dim3(1), 0, s); }
} // namespace isnark
