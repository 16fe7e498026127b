// msm.hip — Pippenger bucket-method MSM over BN254 G1 / G2 for gfx950.
//
// Replaces icicle/backend/cuda/src/msm/cuda_msm.cuh (bucket_method_msm :960-1127 and its kernels,
// SURVEY.md §2.1) behind bn254_msm / bn254_g2_msm.  Results equal the reference's as group elements
// (Σ sᵢ·Pᵢ); the projective representative returned is any valid one, as in the reference
// (different backends already differ there — the prover normalises with to_affine).
//
// Pipeline (everything is enqueued on the caller's stream, no host synchronisation):
//  1. recode + histogram   one thread per scalar: s ↦ signed c-bit digits (d ∈ [−2^(c−1), 2^(c−1)) via
//                          the "+H" trick, scalars above (r − 1)/2 are negated first, so the top
//                          window never carries out), per-bucket counts by global atomics.
//                          Signed digits halve the bucket count versus the reference's unsigned
//                          digits (cuda_msm.cuh:166-203).
//  2. scan                 exclusive prefix sum of W·2^(c−1) counters (one workgroup).
//  3. scatter              second pass over the scalars: each non-zero digit drops its
//                          (point index | sign) into its bucket's slice of `sorted` — a counting sort
//                          keyed by (window, bucket); replaces the three CUB radix sorts + RLE + scan
//                          of the reference (cuda_msm.cuh:401-485, :561-636).
//                          HBM traffic: 2·32 B per scalar + 4 B per digit written.
//  4. bucket accumulation  one thread per bucket walks its slice, gathers the 64-B (G1) / 128-B (G2)
//                          affine bases and adds them into an XYZZ accumulator held in registers
//                          (mixed add 8M+2S, no inversion).  Buckets holding more than
//                          `large_thr` entries (skewed witnesses: 0/1 wires) are left to
//  4b. large buckets       one workgroup per large bucket: 256 partial sums + LDS tree reduction
//                          (replaces cuda_msm.cuh:257-310).  The list is built on the device.
//  5. bucket reduction     per window Σ_b b·B_b: each thread owns K consecutive buckets (running
//                          sum + triangle sum), scales its line sum by its first index with a
//                          double-and-add, then an LDS tree reduction per workgroup.
//  6. tail                 window sums → Horner with c doublings per window → projective result.
//
// Algorithmic bytes of the scatter pass: 16·L·W (index pairs, here 4·L·W written +
// 4·L·W read thanks to the implicit key) + the L·W·P gather of step 4.
#pragma once
#include <atomic>
#include <chrono>
#include <mutex>
#include <string.h>
#include <vector>

#include "ec.h"
#include "ec29.h"
#include "msm_plan.h"

using namespace bn254;
using namespace isnark;

namespace isnark {
extern thread_local float g_last_msm_ms[4];
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

// The bucket accumulation runs on the lazy radix-2^29 field (ec29.h): ~1.4× fewer VALU instructions per mixed
// addition than the 8×32-bit arithmetic of ec.h.  The bucket array between the accumulation and the reduction
// kernels holds XYZZ in the internal encoding (packed canonical Montgomery R' = 2^261, same size as ec.h's XYZZ).
template <class C> struct Lazy;
template <> struct Lazy<G1> { typedef G1L type; };
template <> struct Lazy<G2> { typedef G2L type; };

// `form`: encoding of the affine bases in memory — 0 standard, 1 Montgomery R = 2^256 (zkey files), 2 internal
// (packed canonical Montgomery R' = 2^261, produced once by msm_points_to_internal; no per-load conversion)
// `ib` = 0: classic entry (point index; base = bases[(i − skip)·stride]).  ib > 0: table mode, entry = i | w << ib and
// the bases are W rows of `stride` points, row w holding 2^(c·w)·P (base = bases[w·stride + i − skip]).
__device__ __forceinline__ uint32_t entry_point(uint32_t e, int ib) { return ib ? (e & ((1u << ib) - 1)) : (e & 0x7fffffffu); }
template <class C>
__device__ __forceinline__ typename C::A fetch_base(const typename C::A* bases, uint32_t e, uint32_t skip_below, uint32_t stride, int ib)
{
  const uint32_t i = entry_point(e, ib);
  const uint32_t j = i < skip_below ? 0u : i - skip_below; // entries below skip_below are ignored by the caller
  if (ib) return bases[(size_t)((e & 0x7fffffffu) >> ib) * stride + j];
  return bases[(size_t)j * stride];
}
template <class C>
__device__ __forceinline__ typename Lazy<C>::type::A load_base_lazy(const typename C::A* bases, uint32_t e, uint32_t skip_below, uint32_t stride, int ib, int form, bool& is_zero)
{
  typedef typename Lazy<C>::type CL;
  const typename C::A p = fetch_base<C>(bases, e, skip_below, stride, ib);
  is_zero = entry_point(e, ib) < skip_below || C::aff_is_zero(p); // scalar outside this base set (C MSM), or the identity
  return CL::load_affine(p, form, (e >> 31) != 0);
}

// one bucket: the thread walks the bucket's slice of `sorted`, gathers the bases and adds them into a lazy XYZZ accumulator
template <class C, bool INTO>
__device__ __forceinline__ void accumulate_bucket(const typename C::A* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                  const uint32_t* __restrict__ order, uint32_t t, uint32_t large_thr, uint32_t skip_below, uint32_t stride, int ib, int form, typename C::X* buckets)
{
  typedef typename Lazy<C>::type CL;
  const uint32_t b = order[t]; // neighbouring lanes own buckets of (nearly) equal size
  const uint32_t cnt = counts[b];
  if (cnt > large_thr) return; // step 4b
  // INTO: the bucket array already holds the sums of an earlier SEGMENT of the same scalar vector (the prover sorts and
  // accumulates the head of a witness while its tail is still on the way over PCIe, prover.cpp) — go on from there
  if (INTO && cnt == 0) return;
  const uint32_t* idx = sorted + offsets[b];
  typename CL::X acc = INTO ? CL::x_load_internal(buckets[b]) : CL::x_zero();
  if (sizeof(typename C::A) > 64) {
    // G2: a prefetched 128-byte point would push the kernel past 256 VGPRs (one wave per SIMD); only the index is prefetched
    uint32_t e_nxt = cnt ? idx[0] : 0u;
    for (uint32_t k = 0; k < cnt; k++) {
      const uint32_t e = e_nxt;
      if (k + 1 < cnt) e_nxt = idx[k + 1];
      bool z;
      const typename CL::A p = load_base_lazy<C>(bases, e, skip_below, stride, ib, form, z);
      if (!z) CL::x_madd(acc, p);
    }
  } else {
    // G1, software pipeline: the index two entries ahead and the (gathered, packed) point one entry ahead are in
    // flight while the current mixed addition (~9 k cycles per wave) runs; without it every iteration starts with two
    // dependent memory latencies (H accumulation alone: 4.0 → 3.0 ms)
    uint32_t e_cur = cnt ? idx[0] : 0u, e_nxt = cnt > 1 ? idx[1] : 0u;
    typename C::A pk_cur = fetch_base<C>(bases, e_cur, skip_below, stride, ib);
    for (uint32_t k = 0; k < cnt; k++) {
      const typename C::A pk = pk_cur;
      const uint32_t e = e_cur;
      e_cur = e_nxt;
      if (k + 1 < cnt) pk_cur = fetch_base<C>(bases, e_cur, skip_below, stride, ib);
      if (k + 2 < cnt) e_nxt = idx[k + 2];
      const bool z = entry_point(e, ib) < skip_below || C::aff_is_zero(pk);
      if (!z) CL::x_madd(acc, CL::load_affine(pk, form, (e >> 31) != 0));
    }
  }
  buckets[b] = CL::x_store_internal(acc);
}

// Three instances are used — plain, INTO and STRIDED — because either option costs registers (G1: 136 → 138, past a
// 16-register allocation step, after which one G2 and two G1 accumulation waves no longer fit a SIMD together).
// STRIDED: grid-stride over the size-ordered bucket list.  A launch capped at the number of workgroups the GPU holds at a
// time (`resident` launches, AccumulateLauncher) walks the list in strides — every thread takes one bucket of each size
// stratum, so the threads stay balanced — and, unlike a grid of several times that size, never leaves workgroups waiting
// in the dispatcher: a kernel whose workgroups queue there blocks its hardware pipe for the barrier packets (events!) and
// small kernels of every other queue on that pipe for as long as it runs (measured: the staging events of a witness upload
// stalled for the whole length of such a kernel, prover.cpp).  !STRIDED: one thread per bucket.
template <class C, bool STRIDED, bool INTO>
__global__ __launch_bounds__(256) void msm_accumulate_kernel(const typename C::A* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ counts, const uint32_t* __restrict__ order, uint32_t nbuckets, uint32_t large_thr, uint32_t skip_below, uint32_t stride, int ib, int form,
                                                              typename C::X* buckets)
{
  const uint32_t t0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (!STRIDED) {
    if (t0 >= nbuckets) return;
    accumulate_bucket<C, INTO>(bases, sorted, offsets, counts, order, t0, large_thr, skip_below, stride, ib, form, buckets);
    return;
  }
  for (uint32_t t = t0; t < nbuckets; t += gridDim.x * blockDim.x) accumulate_bucket<C, INTO>(bases, sorted, offsets, counts, order, t, large_thr, skip_below, stride, ib, form, buckets);
}

// in-place conversion of an affine base array to the internal encoding (cold path, once per key)
template <class C>
__global__ __launch_bounds__(256) void msm_points_to_internal_kernel(typename C::A* pts, uint32_t n, int from_form)
{
  typedef typename Lazy<C>::type CL;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const typename C::A p = pts[i];
  if (C::aff_is_zero(p)) return; // the identity stays (0,0)
  const typename CL::A a = CL::load_affine(p, from_form, false);
  pts[i] = CL::store_affine_internal(a);
}

template <class C>
__device__ __forceinline__ typename C::X block_reduce(typename C::X v, typename C::X* sh, int nthreads)
{
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = nthreads >> 1; s > 0; s >>= 1) {
    if (tid < s) {
      v = C::x_add(v, sh[tid + s]);
      sh[tid] = v;
    }
    __syncthreads();
  }
  return v;
}

// tree sum over the workgroup on the lazy field; LDS holds unpacked lazy XYZZ (144 B G1 / 288 B G2 per thread)
template <class C>
__device__ __forceinline__ typename Lazy<C>::type::X block_reduce_lazy(typename Lazy<C>::type::X v, typename Lazy<C>::type::X* sh, int nthreads)
{
  typedef typename Lazy<C>::type CL;
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = nthreads >> 1; s > 0; s >>= 1) {
    if (tid < s) {
      v = CL::x_add(v, sh[tid + s]);
      sh[tid] = v;
    }
    __syncthreads();
  }
  return v;
}

// large buckets, step 1: one workgroup per (bucket, chunk) work item sums ≤ MSM_LARGE_CHUNK entries: strided mixed additions per
// thread, then a tree over the lazy field no wider than the item (a 200-entry bucket folds 256 → 1 in 8 levels of which the
// first is free).  LDS: blockDim lazy XYZZ.
template <class C>
__global__ __launch_bounds__(256) void msm_accumulate_large_kernel(const typename C::A* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
                                                                    const uint32_t* __restrict__ counts, const uint32_t* __restrict__ n_large, const uint2* __restrict__ items, uint32_t item_cap,
                                                                    uint32_t skip_below, uint32_t stride, int ib, int pts_mont, typename C::X* __restrict__ item_partials)
{
  typedef typename Lazy<C>::type CL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typename CL::X* sh = reinterpret_cast<typename CL::X*>(smem);
  const uint32_t ni = min(n_large[2], item_cap);
  for (uint32_t it = blockIdx.x; it < ni; it += gridDim.x) {
    const uint2 w = items[it];
    const uint32_t b = w.x;
    const uint32_t lo = offsets[b] + w.y * MSM_LARGE_CHUNK, hi = min(offsets[b] + counts[b], lo + MSM_LARGE_CHUNK);
    typename CL::X lacc = CL::x_zero();
    for (uint32_t k = lo + threadIdx.x; k < hi; k += blockDim.x) {
      bool z;
      const typename CL::A p = load_base_lazy<C>(bases, sorted[k], skip_below, stride, ib, pts_mont, z);
      if (!z) CL::x_madd(lacc, p);
    }
    int width = 1;
    while (width < (int)blockDim.x && (uint32_t)width < hi - lo) width <<= 1;
    lacc = block_reduce_lazy<C>(lacc, sh, width);
    if (threadIdx.x == 0) item_partials[it] = CL::x_store_internal(lacc); // the bucket array's encoding
    __syncthreads();
  }
}
// large buckets, step 2: one workgroup per large bucket sums its chunk partials into the bucket
template <class C>
__global__ __launch_bounds__(256) void msm_combine_large_kernel(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ n_large, const uint32_t* __restrict__ large_list,
                                                                 const uint32_t* __restrict__ large_first, const typename C::X* __restrict__ item_partials, int into, typename C::X* buckets)
{
  typedef typename Lazy<C>::type CL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typename CL::X* sh = reinterpret_cast<typename CL::X*>(smem);
  const uint32_t nl = n_large[0];
  for (uint32_t li = blockIdx.x; li < nl; li += gridDim.x) {
    const uint32_t b = large_list[li];
    const uint32_t nch = (counts[b] + MSM_LARGE_CHUNK - 1) / MSM_LARGE_CHUNK, first = large_first[li];
    if (nch == 1) { // one chunk (most buckets just above the threshold): nothing to sum
      if (threadIdx.x == 0) {
        if (into) buckets[b] = CL::x_store_internal(CL::x_add(CL::x_load_internal(buckets[b]), CL::x_load_internal(item_partials[first])));
        else buckets[b] = item_partials[first];
      }
      continue;
    }
    typename CL::X acc = CL::x_zero();
    for (uint32_t k = threadIdx.x; k < nch; k += blockDim.x) acc = CL::x_add(acc, CL::x_load_internal(item_partials[first + k]));
    int width = 1;
    while (width < (int)blockDim.x && (uint32_t)width < nch) width <<= 1;
    acc = block_reduce_lazy<C>(acc, sh, width);
    if (threadIdx.x == 0) {
      if (into) acc = CL::x_add(acc, CL::x_load_internal(buckets[b])); // what an earlier segment left in this bucket
      buckets[b] = CL::x_store_internal(acc);
    }
    __syncthreads();
  }
}

// "last workgroup folds": a workgroup publishes its results, takes a ticket of its window, and the one that draws the
// last ticket sums the window's gridDim.x results — no separate fold launch, whose waves (392 registers for G2) could only
// start on a SIMD that the running accumulations had drained completely (G2 fold at 1.6 M constraints: 1.4 ms, nearly
// all of it waiting).  Returns true in the workgroup that has to fold (after an agent-scope fence: the others' stores
// are visible).
__device__ __forceinline__ bool last_workgroup_of_window(uint32_t* tickets)
{
  __shared__ uint32_t s_ticket;
  if (threadIdx.x == 0) {
    __threadfence();
    s_ticket = atomicAdd(tickets + blockIdx.y, 1u);
  }
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return false;
  __threadfence();
  return true;
}

// Σ_b (b+1)·B_b per window of the CLASSIC layout.  grid = (workgroups per window, W); each thread owns K = 2^k_log buckets.
// Buckets arrive in the internal encoding; the window sums leave as ec.h XYZZ (Montgomery R = 2^256) for the tails.
// raw / tickets: scratch for the per-workgroup results (internal encoding) and one zeroed counter per window, used when
// gridDim.x > 1.  (Table mode — the prover's cached keys — reduces its single bucket set with msm_zeta_reduce_kernel.)
template <class C>
__global__ __launch_bounds__(256) void msm_bucket_reduce_kernel(const typename C::X* __restrict__ buckets, uint32_t NB, int k_log, typename C::X* __restrict__ out, typename C::X* raw, uint32_t* tickets)
{
  typedef typename Lazy<C>::type CL;
  typedef typename CL::X X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // thread within the window
  const uint32_t base = t << k_log;
  const typename C::X* B = buckets + (size_t)blockIdx.y * NB + base;
  X tri = CL::x_zero();
  {
    X line = CL::x_zero();
    for (int j = (1 << k_log) - 1; j >= 0; j--) {
      line = CL::x_add(line, CL::x_load_internal(B[j]));
      tri = CL::x_add(tri, line);
    }
    // + base·line  (double-and-add, MSB first)
    if (base != 0 && !CL::x_is_zero(line)) {
      X m = CL::x_zero();
      for (int bit = 31 - __clz(base); bit >= 0; bit--) {
        m = CL::x_dbl(m);
        if ((base >> bit) & 1) m = CL::x_add(m, line);
      }
      tri = CL::x_add(tri, m);
    }
  }
  tri = block_reduce_lazy<C>(tri, sh, (int)blockDim.x);
  const size_t bpw = gridDim.x;
  if (bpw == 1) {
    if (threadIdx.x == 0) out[blockIdx.y] = CL::x_store(tri);
    return;
  }
  if (threadIdx.x == 0) raw[(size_t)blockIdx.y * bpw + blockIdx.x] = CL::x_store_internal(tri);
  if (!last_workgroup_of_window(tickets)) return;
  X v = threadIdx.x < bpw ? CL::x_load_internal(raw[(size_t)blockIdx.y * bpw + threadIdx.x]) : CL::x_zero();
  int width = 1;
  while ((size_t)width < bpw) width <<= 1;
  __syncthreads();
  v = block_reduce_lazy<C>(v, sh, width);
  if (threadIdx.x == 0) out[blockIdx.y] = CL::x_store(v);
}

// ---- bit-plane tree reduction (table mode) ----------------------------------------------------------------------------
// Table mode has ONE bucket set of NB = 2^t buckets per MSM and needs  Σ_b (b + 1)·B_b = T + Σ_{j<t} 2^j·S_j  with the plain sum
// T = Σ_b B_b and the bit-plane sums S_j = Σ_{b : bit j of b set} B_b: t + 1 PLAIN sums — no per-thread running-sum pair, no
// double-and-add by the thread's first index, no suffix scan — and the t doublings of the Horner form are left to the host tail
// (one core: ≈ 30 µs for G1, 90 µs for G2).  All t + 1 sums come out of one pruned superset-sum ("zeta") transform: a workgroup
// holds M = 2^m buckets in LDS; stage j (from m − 1 down to 0) adds the upper half of every live group of 2^(j+1) entries into
// its lower half, and the upper half of group 0 — still holding its values from before the addition — stays behind as the
// new group of bit j: after stage 0, a[0] = T and a[2^j] = S_j of the block.  (m − j)·2^j additions at stage j: 2M per block,
// the same 2·NB additions as a running-sum reduction, but a dependent chain of m additions per level instead of 2K + ≈ 25 + 2·8:
//   level 1  blocks of 256 buckets          → R1[i][block],   i = 0 (total), 1 + j (bit j)                       (chain: 8)
//   level 2  the same transform over the blocks of every R1[i][·] row                                               (chain: 8)
//   final    the workgroup that draws the last ticket of level 2 sums the ≤ 8 level-2 blocks:                        (chain: 3)
//            low bits from row 1 + j's totals, middle bits from row 0's bit sums, high bits from subsets of row 0's totals.
// The reference halves the bucket count per launch and doubles the window count (cuda_msm.cuh:821-956: log c launches of
// dependent additions); the three table-mode reductions this replaces (per-thread triangle sums, suffix scans for small
// sets, row / column sums) had chains of 56, 34 and 26 additions.  Measured on MI355X, G1 / G2 set of 2^19 buckets alone:
// see HISTORY.md §3.2-5.
constexpr int ZR_LOG = 8, ZR_M = 1 << ZR_LOG, ZR_T = ZR_M / 2, ZR_OUT = ZR_LOG + 1;

// in: `gridDim.y` rows of n entries (internal encoding; entries ≥ n count as the identity), row r at in + r·n.
// out[(r·ZR_OUT + i)·gridDim.x + block]: the block's total (i = 0) and bit sums (i = 1 + j), internal encoding.
// final_out != nullptr (level 2 of `nsets` bucket sets, rows = nsets × ZR_OUT): the last workgroup writes, per set,
// [T | S_0 … S_{nbits−1}] as ec.h XYZZ (Montgomery R = 2^256) for the host tail.
template <class C>
__global__ __launch_bounds__(ZR_T) void msm_zeta_reduce_kernel(const typename C::X* __restrict__ in, uint32_t n, typename C::X* __restrict__ out, uint32_t* ticket, int nbits,
                                                               typename C::X* __restrict__ final_out)
{
  typedef typename Lazy<C>::type CL;
  typedef typename CL::X X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* a = reinterpret_cast<X*>(smem); // ZR_M entries
  const uint32_t tid = threadIdx.x, blk = blockIdx.x, nblk = gridDim.x, row = blockIdx.y;
  {
    const uint32_t e0 = blk * ZR_M + tid, e1 = e0 + ZR_T;
    const typename C::X* src = in + (size_t)row * n;
    const X x1 = e1 < n ? CL::x_load_internal(src[e1]) : CL::x_zero();
    const X x0 = e0 < n ? CL::x_load_internal(src[e0]) : CL::x_zero();
    a[tid] = CL::x_add(x0, x1); // stage m − 1
    a[tid + ZR_T] = x1;
  }
  for (int j = ZR_LOG - 2; j >= 0; j--) {
    __syncthreads();
    const uint32_t active = (uint32_t)(ZR_LOG - j) << j;
    if (tid < active) {
      const uint32_t gi = tid >> j, y = tid & ((1u << j) - 1);
      const uint32_t p = (gi ? (1u << (j + gi)) : 0u) + y;
      a[p] = CL::x_add(a[p], a[p + (1u << j)]);
    }
  }
  __syncthreads();
  if (tid < ZR_OUT) out[((size_t)row * ZR_OUT + tid) * nblk + blk] = CL::x_store_internal(a[tid ? (1u << (tid - 1)) : 0u]);
  if (!final_out) return;
  // ---- last workgroup of level 2: the sums over the ≤ 8 level-2 blocks
  __shared__ uint32_t s_last;
  if (tid == 0) {
    __threadfence();
    s_last = atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const uint32_t nsets = gridDim.y / ZR_OUT, nout = 1u + (uint32_t)nbits;
  // R2(r1, i2, b2) with r1 = set·ZR_OUT + i1: what level 2 wrote for level-1 row i1 of the set
  auto R2 = [&](uint32_t set, uint32_t i1, uint32_t i2, uint32_t b2) -> X {
    if (b2 >= nblk) return CL::x_zero();
    return CL::x_load_internal(out[((size_t)(set * ZR_OUT + i1) * ZR_OUT + i2) * nblk + b2]);
  };
  for (uint32_t set = 0; set < nsets; set++) {
    // output q: 0 = T, 1 + j = S_j;  four threads per output, each adding two of the ≤ 8 terms, then a two-level tree
    for (uint32_t q0 = 0; q0 < nout; q0 += ZR_T / 4) {
      const uint32_t q = q0 + tid / 4, k = tid & 3;
      X v = CL::x_zero();
      if (q < nout) {
        uint32_t i1 = 0, i2 = 0, hb = 0xffffffffu; // hb: only blocks with this bit set (high bits)
        if (q >= 1) {
          const uint32_t j = q - 1;
          if (j < (uint32_t)ZR_LOG) i1 = 1 + j;
          else if (j < 2u * ZR_LOG) i2 = 1 + (j - ZR_LOG);
          else hb = j - 2u * ZR_LOG;
        }
        const bool t0 = hb == 0xffffffffu || ((k >> hb) & 1), t1 = hb == 0xffffffffu || (((k + 4) >> hb) & 1);
        const X u0 = t0 ? R2(set, i1, i2, k) : CL::x_zero();
        const X u1 = t1 ? R2(set, i1, i2, k + 4) : CL::x_zero();
        v = CL::x_add(u0, u1);
      }
      __syncthreads();
      a[tid] = v;
      __syncthreads();
      if (k < 2) a[tid] = v = CL::x_add(v, a[tid + 2]);
      __syncthreads();
      if (k == 0 && q < nout) final_out[(size_t)set * nout + q] = CL::x_store(CL::x_add(v, a[tid + 1]));
    }
  }
}

// window sums → Horner → projective standard form.  One workgroup of 64 threads.
template <class C>
__global__ __launch_bounds__(64) void msm_tail_kernel(const typename C::X* __restrict__ partials, int W, int bpw, int c, int wide, typename C::P* __restrict__ result)
{
  typedef typename C::X X;
  __shared__ X wsum[64];
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    X acc = C::x_zero();
    for (int k = 0; k < bpw; k++) acc = C::x_add(acc, partials[(size_t)w * bpw + k]);
    wsum[w] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    X acc = wsum[W - 1];
    for (int w = W - 2; w >= 0; w--) {
      for (int j = 0; j < (w < wide ? c : c - 1); j++) acc = C::x_dbl(acc); // window w + 1 starts where window w ends
      acc = C::x_add(acc, wsum[w]);
    }
    typename C::P p = C::p_from_mont(C::x_to_projective(acc));
    // the identity must come out as (0, 1, 0) in STANDARD form
    *result = p;
  }
}

// out[i] = s[i]·G for a fixed generator, 8-bit windows over a precomputed table (32 × 255 affine points)
template <class C>
__global__ __launch_bounds__(256) void fixed_base_table_kernel(typename C::A gen_mont, typename C::X* table)
{
  // single thread per window row start would serialise 255 adds; rows are independent given 256^w·G
  // computed sequentially by thread 0 first.
  typedef typename C::X X;
  __shared__ X rowbase[32];
  if (threadIdx.x == 0) {
    X cur = C::x_from_affine(gen_mont);
    for (int w = 0; w < 32; w++) {
      rowbase[w] = cur;
      for (int k = 0; k < 8; k++) cur = C::x_dbl(cur);
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int w = threadIdx.x;
    X acc = rowbase[w];
    table[w * 255] = acc;
    for (int d = 1; d < 255; d++) {
      acc = C::x_add(acc, rowbase[w]);
      table[w * 255 + d] = acc;
    }
  }
}
template <class C>
__global__ __launch_bounds__(256) void fixed_base_mul_kernel(const fe* __restrict__ s, uint64_t n, const typename C::X* __restrict__ table, typename C::P* __restrict__ out)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe sc = ld_fe(s + i);
  typename C::X acc = C::x_zero();
  for (int w = 0; w < 32; w++) {
    const uint32_t d = (sc.l[w >> 2] >> ((w & 3) * 8)) & 0xff;
    if (d) acc = C::x_add(acc, table[w * 255 + d - 1]);
  }
  out[i] = C::x_to_projective(acc); // Montgomery projective; normalised by the batch-affine kernels below
}

// the one inversion of a chunk: 600 divsteps on the lazy field (ff29.h: inv_ds, ≈ 14 k instructions) instead of ff.h's Fermat ladder
// (≈ 100 k instructions on ONE lane: 0.25 ms of latency per launch of the sliced table build, a fifth of its length).  Montgomery-256
// in and out, 0 ↦ 0 like Fq::inv.
__device__ __forceinline__ fe fq_inv_fast(const fe& a) { return f29::to_mont256(f29::inv_ds(f29::from_mont256(a))); }
template <class F> struct FastInv;
template <> struct FastInv<FqOps> {
  static __device__ __forceinline__ fe inv(const fe& a) { return fq_inv_fast(a); }
};
template <> struct FastInv<Fq2Ops> {
  static __device__ __forceinline__ fe2 inv(const fe2& a)
  {
    const fe d = fq_inv_fast(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
    return {Fq::mul(a.c0, d), Fq::neg(Fq::mul(a.c1, d))};
  }
};
// batched projective → affine (standard form) with one inversion per thread-chunk (Montgomery trick)
template <class C, class F>
__global__ __launch_bounds__(64) void batch_to_affine_kernel(const typename C::P* __restrict__ in, uint64_t n, int chunk, typename C::A* __restrict__ out, typename F::T* __restrict__ scratch)
{
  typedef typename F::T T;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lo = t * chunk;
  if (lo >= n) return;
  const uint64_t hi = lo + chunk < n ? lo + chunk : n;
  // prefix products of z (skipping zeros) kept in scratch[i]
  T run = F::one();
  for (uint64_t i = lo; i < hi; i++) {
    scratch[i] = run;
    T z = in[i].z;
    if (!F::is_zero(z)) run = F::mul(run, z);
  }
  T inv = FastInv<F>::inv(run);
  for (uint64_t i = hi; i-- > lo;) {
    T z = in[i].z;
    typename C::A a;
    if (F::is_zero(z)) {
      a.x = F::zero();
      a.y = F::zero();
    } else {
      T zi = F::mul(inv, scratch[i]);
      inv = F::mul(inv, z);
      a.x = F::from_mont(F::mul(in[i].x, zi));
      a.y = F::from_mont(F::mul(in[i].y, zi));
    }
    out[i] = a;
  }
}


// table mode, cold path: rows[w·n + i] = 2^(c·w)·P_i (w < W) as Montgomery-256 projective points; the caller turns
// them into affine (batch inversion) and into the internal encoding.  One thread per base, W·c doublings.
template <class C>
__global__ __launch_bounds__(256) void msm_table_rows_kernel(const typename C::A* __restrict__ pts, uint32_t n, int from_form, int c, int W, int wide, typename C::P* __restrict__ rows)
{
  typedef typename Lazy<C>::type CL;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const typename C::A p = pts[i];
  typename CL::X x = CL::x_zero();
  if (!C::aff_is_zero(p)) {
    const typename CL::A a = CL::load_affine(p, from_form, false);
    CL::x_madd(x, a);
  }
  for (int w = 0; w < W; w++) {
    rows[(size_t)w * n + i] = C::x_to_projective(CL::x_store(x)); // identity → (0, 1, 0)
    if (w + 1 < W)
      for (int k = 0; k < (w < wide ? c : c - 1); k++) x = CL::x_dbl(x); // row w + 1 sits at the bit where window w ends
  }
}
// ---- content guard of the automatic tables (msm_plan.h) -----------------------------------------------------------------------
// 64-bit sum over all 16-byte words of a position-dependent mix: order-independent (one atomic per workgroup), sensitive to
// any changed, moved or swapped word.  `out` must be zero before the launch.
__global__ __launch_bounds__(256) void bases_hash_sum_kernel(const uint4* __restrict__ p, uint64_t nwords, unsigned long long* __restrict__ out)
{
  unsigned long long h = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 w = p[i];
    const unsigned long long a = (unsigned long long)w.x | ((unsigned long long)w.y << 32), b = (unsigned long long)w.z | ((unsigned long long)w.w << 32);
    const unsigned long long k = (i + 1) * 0x9E3779B97F4A7C15ull;
    unsigned long long x = (a ^ k) * 0xFF51AFD7ED558CCDull;
    x ^= x >> 33;
    unsigned long long y = (b + ((k << 29) | (k >> 35))) * 0xC4CEB9FE1A85EC53ull;
    y ^= y >> 29;
    h += (x + y) * 0xD6E8FEB86659FD93ull + (x ^ (y >> 17));
  }
  for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o, 64);
  __shared__ unsigned long long sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}
// Guarded refresh: nothing when the bases still hash to the sum the table was built from; otherwise every row of the table is
// recomputed from the bases in place — the same functions as the regular build (rows → affine → internal encoding), but one
// thread per base with one field inversion per row and no temporaries: slow (≈ 0.1–0.3 s at a million bases), rare, correct.
template <class C, class F>
__global__ __launch_bounds__(128) void msm_table_refresh_kernel(const typename C::A* __restrict__ pts, uint32_t n, int from_form, int c, int W, int wide, typename C::A* __restrict__ table,
                                                                const unsigned long long* __restrict__ built_sum, const unsigned long long* __restrict__ cur_sum)
{
  typedef typename Lazy<C>::type CL;
  if (*built_sum == *cur_sum) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const typename C::A p = pts[i];
  typename CL::X x = CL::x_zero();
  if (!C::aff_is_zero(p)) {
    const typename CL::A a = CL::load_affine(p, from_form, false);
    CL::x_madd(x, a);
  }
  for (int w = 0; w < W; w++) {
    const typename C::P pr = C::x_to_projective(CL::x_store(x)); // Montgomery-256 projective, identity → (0, 1, 0)
    typename C::A a;
    if (F::is_zero(pr.z)) {
      a.x = F::zero();
      a.y = F::zero();
    } else {
      const typename F::T zi = F::inv(pr.z);
      a.x = F::from_mont(F::mul(pr.x, zi));
      a.y = F::from_mont(F::mul(pr.y, zi));
      a = CL::store_affine_internal(CL::load_affine(a, 0, false));
    }
    table[(size_t)w * n + i] = a;
    if (w + 1 < W)
      for (int k = 0; k < (w < wide ? c : c - 1); k++) x = CL::x_dbl(x);
  }
}
__global__ void bases_hash_commit_kernel(unsigned long long* built_sum, const unsigned long long* cur_sum) { *built_sum = *cur_sum; }

inline hipError_t bases_hash_sum(const void* bases, size_t bytes, unsigned long long* out, hipStream_t s)
{
  hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  const uint64_t nwords = bytes / 16;
  unsigned grid = (unsigned)((nwords + 256 * 8 - 1) / (256 * 8));
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(bases_hash_sum_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)bases, nwords, out);
  return hipGetLastError();
}

// `async`: temporaries from the stream's workspace arena, nothing synchronised — the table is complete in stream order (the
// automatic tables of bn254_msm); else plain allocations and a synchronised stream on return (cache build of the prover).
template <class C, class F>
eIcicleError build_table_run(const void* d_points, uint32_t n, int from_form, const MsmGeom& g, hipStream_t s, void** d_table, bool async = false)
{
  typedef typename C::A A;
  typedef typename C::P P;
  *d_table = nullptr;
  const uint64_t m = (uint64_t)n * g.W;
  A* table = nullptr;
  HIP_TRY(hipMalloc((void**)&table, (m ? m : 1) * sizeof(A)), ICICLE_ALLOCATION_FAILED);
  if (n) {
    P* rows = nullptr;
    typename F::T* scratch = nullptr;
    WsScoped<P> ws_rows;
    WsScoped<typename F::T> ws_scratch;
    if (async) {
      if (ws_rows.alloc(m, s) != hipSuccess || ws_scratch.alloc(m, s) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(table);
        return ICICLE_ALLOCATION_FAILED;
      }
      rows = ws_rows.p;
      scratch = ws_scratch.p;
    } else {
      if (hipMalloc((void**)&rows, m * sizeof(P)) != hipSuccess) {
        (void)hipFree(table);
        return ICICLE_ALLOCATION_FAILED;
      }
      if (hipMalloc((void**)&scratch, m * sizeof(typename F::T)) != hipSuccess) {
        (void)hipFree(rows);
        (void)hipFree(table);
        return ICICLE_ALLOCATION_FAILED;
      }
    }
    hipLaunchKernelGGL((msm_table_rows_kernel<C>), dim3((n + 255) / 256), dim3(256), 0, s, (const A*)d_points, n, from_form, g.c, g.W, g.wide, rows);
    const int chunk = 32;
    const uint64_t nthreads = (m + chunk - 1) / chunk;
    hipLaunchKernelGGL((batch_to_affine_kernel<C, F>), dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, s, rows, m, chunk, table, scratch);
    hipLaunchKernelGGL((msm_points_to_internal_kernel<C>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, table, (uint32_t)m, 0);
    const eIcicleError e = check_launch("msm_build_table");
    if (async) {
      if (e != ICICLE_SUCCESS) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(table);
        return e;
      }
    } else {
      const hipError_t he = hipStreamSynchronize(s);
      (void)hipFree(rows);
      (void)hipFree(scratch);
      if (e != ICICLE_SUCCESS || he != hipSuccess) {
        (void)hipFree(table);
        return e != ICICLE_SUCCESS ? e : ICICLE_SYNCHRONIZATION_FAILED;
      }
    }
  }
  *d_table = table;
  return ICICLE_SUCCESS;
}

// table[w·n + lo + i] = internal encoding of aff[w·cnt + i] (standard-form affine, identity (0,0) kept): the last step of a
// SLICE of the table build below, out of the slice's compact temporaries into the rows of the full table
template <class C>
__global__ __launch_bounds__(256) void msm_table_place_kernel(const typename C::A* __restrict__ aff, uint32_t cnt, int W, typename C::A* __restrict__ table, uint32_t n, uint32_t lo)
{
  typedef typename Lazy<C>::type CL;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cnt * (uint32_t)W) return;
  const uint32_t w = t / cnt, i = t - w * cnt;
  typename C::A p = aff[t];
  if (!C::aff_is_zero(p)) p = CL::store_affine_internal(CL::load_affine(p, 0, false));
  table[(size_t)w * n + lo + i] = p;
}

// The same table in SLICES of the base array: rows → affine → internal encoding per slice of `slice` bases, with temporaries of
// the slice's size (a whole-array build holds 2 GB (G1) / 5.3 GB (G2) of them at 1.6 M bases) and launches no larger than the
// device holds at a time — what the prover's DEFERRED build needs (cache.cpp: the tables of a key are built behind its first
// proofs, beside the proves that follow: a grid with queued workgroups would keep a hardware pipe busy for its whole length
// and starve their short kernels, prover.cpp "head").  `cancel` is polled between slices (the key is evicted mid-build).
// Everything is enqueued on s; returns after the stream has been synchronised.
template <class C, class F>
eIcicleError build_table_sliced_run(const void* d_points, uint32_t n, int from_form, const MsmGeom& g, hipStream_t s, void** d_table, uint32_t slice, const std::atomic<bool>* cancel)
{
  typedef typename C::A A;
  typedef typename C::P P;
  *d_table = nullptr;
  const uint64_t m = (uint64_t)n * g.W;
  A* table = nullptr;
  HIP_TRY(hipMalloc((void**)&table, (m ? m : 1) * sizeof(A)), ICICLE_ALLOCATION_FAILED);
  if (!n) {
    *d_table = table;
    return ICICLE_SUCCESS;
  }
  if (slice > n) slice = n;
  const uint64_t ms = (uint64_t)slice * g.W;
  static const bool trace_tb = getenv("ICICLE_SNARK_TRACE_TABLES") != nullptr;
  const auto tb0 = std::chrono::steady_clock::now();
  auto tb_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(); };
  P* rows = nullptr;
  A* aff = nullptr;
  typename F::T* scratch = nullptr;
  auto cleanup = [&](bool with_table) {
    (void)hipStreamSynchronize(s);
    if (rows) (void)hipFree(rows);
    if (aff) (void)hipFree(aff);
    if (scratch) (void)hipFree(scratch);
    if (with_table) (void)hipFree(table);
  };
  if (hipMalloc((void**)&rows, ms * sizeof(P)) != hipSuccess || hipMalloc((void**)&aff, ms * sizeof(A)) != hipSuccess || hipMalloc((void**)&scratch, ms * sizeof(typename F::T)) != hipSuccess) {
    (void)hipGetLastError();
    cleanup(true);
    return ICICLE_ALLOCATION_FAILED;
  }
  const double t_alloc = tb_ms();
  const int chunk = 32;
  // up to DEPTH slices enqueued ahead of the one that runs (the temporaries are reused in stream order): the stream does not run
  // dry when this thread is descheduled — with one slice in flight and one queued (rounds 4–5) a host whose CPU quota was taken by
  // other work doubled and quadrupled the build (profiles/r06_cold_path_outliers.txt: 240 → 507 / 900–980 ms beside 32 / 64 busy
  // processes) — and the host is still no more than DEPTH slices (≈ 10 ms) ahead of a cancel
  constexpr int DEPTH = 8;
  hipEvent_t ev[DEPTH] = {};
  for (int q = 0; q < DEPTH; q++) (void)hipEventCreateWithFlags(&ev[q], hipEventDisableTiming);
  struct EvGuard {
    hipEvent_t* e;
    ~EvGuard()
    {
      for (int q = 0; q < DEPTH; q++)
        if (e[q]) (void)hipEventDestroy(e[q]);
    }
  } ev_guard{ev};
  uint32_t k = 0;
  for (uint32_t lo = 0; lo < n; lo += slice, k++) {
    if (k >= (uint32_t)DEPTH && ev[k % DEPTH]) (void)hipEventSynchronize(ev[k % DEPTH]); // slice k − DEPTH is done
    if (cancel && cancel->load(std::memory_order_relaxed)) {
      cleanup(true);
      return ICICLE_UNKNOWN_ERROR;
    }
    const uint32_t cnt = n - lo < slice ? n - lo : slice;
    const uint64_t mc = (uint64_t)cnt * g.W, nthreads = (mc + chunk - 1) / chunk;
    hipLaunchKernelGGL((msm_table_rows_kernel<C>), dim3((cnt + 255) / 256), dim3(256), 0, s, (const A*)d_points + lo, cnt, from_form, g.c, g.W, g.wide, rows);
    hipLaunchKernelGGL((batch_to_affine_kernel<C, F>), dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, s, rows, mc, chunk, aff, scratch);
    hipLaunchKernelGGL((msm_table_place_kernel<C>), dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, s, aff, cnt, g.W, table, n, lo);
    const eIcicleError e = check_launch("msm_build_table (slice)");
    if (e != ICICLE_SUCCESS) {
      cleanup(true);
      return e;
    }
    if (ev[k % DEPTH]) (void)hipEventRecord(ev[k % DEPTH], s);
  }
  const double t_enq = tb_ms();
  const hipError_t he = hipStreamSynchronize(s);
  const double t_sync = tb_ms();
  cleanup(he != hipSuccess);
  if (trace_tb) fprintf(stderr, "[tables] %s n=%u: table+temps hipMalloc %.1f ms, enqueue %.1f, wait %.1f, hipFree %.1f\n", sizeof(A) > 64 ? "G2" : "G1", n, t_alloc, t_enq - t_alloc, t_sync - t_enq, tb_ms() - t_sync);
  if (he != hipSuccess) return ICICLE_SYNCHRONIZATION_FAILED;
  *d_table = table;
  return ICICLE_SUCCESS;
}

template <class C>
eIcicleError points_to_internal_run(void* d_points, uint32_t n, int from_form, hipStream_t s)
{
  if (n == 0) return ICICLE_SUCCESS;
  hipLaunchKernelGGL((msm_points_to_internal_kernel<C>), dim3((n + 255) / 256), dim3(256), 0, s, (typename C::A*)d_points, n, from_form);
  return check_launch("msm_points_to_internal");
}

// launch of the bucket-accumulation kernel; the G2 instance lives in its own translation unit
// (msm_g2_acc.hip, Fq2 arithmetic inlined)
template <class C>
struct AccumulateLauncher {
  // `resident`: no more workgroups than the device holds at a time (the kernel strides over the bucket list)
  static void launch(const SortPlan* pl, const typename C::A* d_points, int mont_pt, uint32_t skip_below, uint32_t stride, hipStream_t s, typename C::X* buckets, int into, bool resident = false)
  {
    unsigned grid = (pl->nbuckets + 255) / 256;
    if (resident) {
      static std::atomic<unsigned> cap{0};
      unsigned c = cap.load();
      if (!c) {
        int per_cu = 0, dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msm_accumulate_kernel<C, true, false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        // (the occupancy API can come out one workgroup per CU high on this stack — MI355X guide, "Correctness boundaries" — and one
        //  queued workgroup is what must not happen here: bound it by the register file, 512 VGPRs per SIMD lane, a 256-thread
        //  workgroup = one wave on each of the four SIMDs)
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(msm_accumulate_kernel<C, true, false>)) == hipSuccess && fa.numRegs > 0) {
          const int by_regs = 512 / ((fa.numRegs + 15) & ~15);
          if (by_regs >= 1 && by_regs < per_cu) per_cu = by_regs;
        }
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        (void)hipGetLastError();
        c = (unsigned)per_cu * (unsigned)cus;
        cap.store(c);
      }
      if (grid > c) grid = c;
    }
#define ISNARK_ACC_LAUNCH(STRIDED_, INTO_)                                                                                                                                                   \
  hipLaunchKernelGGL((msm_accumulate_kernel<C, STRIDED_, INTO_>), dim3(grid), dim3(256), 0, s, d_points, pl->sorted, pl->offsets, pl->counts, pl->order, pl->nbuckets, pl->large_thr, skip_below, \
                     stride, pl->g.tab ? pl->g.IB : 0, mont_pt, buckets)
    // plain | INTO (tail of a witness) | STRIDED (head of a witness; a strided launch that continues buckets does not occur)
    if (resident && !into) ISNARK_ACC_LAUNCH(true, false);
    else if (into) ISNARK_ACC_LAUNCH(false, true);
    else ISNARK_ACC_LAUNCH(false, false);
#undef ISNARK_ACC_LAUNCH
  }
};
#if defined(ISNARK_G2_ACC_EXTERN)
template <>
struct AccumulateLauncher<G2> {
  static void launch(const SortPlan* pl, const G2::A* d_points, int mont_pt, uint32_t skip_below, uint32_t stride, hipStream_t s, G2::X* buckets, int into, bool resident = false)
  {
    isnark::msm_g2_accumulate_launch(pl, d_points, mont_pt, skip_below, stride, s, buckets, into, resident);
  }
};
#endif

// classic layout: geometry of msm_bucket_reduce_kernel
struct ReduceShape {
  int k_log;
  uint32_t tpw, rblock, bpw;
};
template <class X>
ReduceShape reduce_shape(const MsmGeom& g)
{
  ReduceShape r;
  int lnb = 0;
  while ((1u << lnb) < g.NBb) lnb++;
  // buckets per thread: 16 for the large bucket sets; fewer when that would leave the GPU mostly empty (the kernel is a
  // latency-bound chain of 2K + log₂(first index) + log₂(block) point additions per thread) — aim at ≥ 32 K threads
  int ltot = 0;
  while ((1u << ltot) < g.NBb * (uint32_t)g.Wb) ltot++;
  int k = ltot - 15;
  if (k < 0) k = 0;
  if (k > 4) k = 4;
  const uint32_t rb_max = sizeof(X) > 128 ? 128 : 256;   // LDS tree buffer ≤ 36 KiB
  while (k < 4 && (uint64_t)g.Wb * (g.NBb >> k) / rb_max > 128) k++;
  r.k_log = k < lnb ? k : lnb;
  for (;; r.k_log++) {
    r.tpw = g.NBb >> r.k_log;                            // reduce threads per window
    r.rblock = r.tpw < rb_max ? r.tpw : rb_max;
    r.bpw = r.tpw / r.rblock;
    if (r.bpw <= r.rblock) break; // the workgroup that finishes a window last folds its bpw results with one thread each
  }
  return r;
}

// more than 64 KiB of dynamic LDS (two lazy XYZZ per thread) has to be allowed once per kernel
template <class K>
inline void allow_big_lds(K kernel, size_t bytes)
{
  if (bytes > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// stages 4 + 4b for one base set: bucket accumulation (+ large buckets) into `buckets` (pl->nbuckets XYZZ, internal encoding).
// `into`: the array holds the sums of an earlier segment of the same scalar vector (same geometry) and is continued;
// otherwise every bucket is written (empty ones as the identity).
template <class C>
eIcicleError msm_accumulate_stage(const SortPlan* pl, const typename C::A* d_points, int mont_pt, uint32_t skip_below, uint32_t stride, hipStream_t s, typename C::X* buckets, bool into, MsmProfile* prof, bool resident = false, const LargeSide* side = nullptr)
{
  typedef typename C::X X;
  WsScoped<X> item_partials;
  // (side stream: everything enqueued on s so far — the bucket array's earlier users, the head's sums that `into` continues — is
  //  ordered in front of the large-bucket kernels by `fork`; they write the large buckets only, the accumulation the others)
  const bool on_side = side && side->stream && side->fork && side->join;
  hipStream_t sl = on_side ? side->stream : s;
  if (on_side) {
    HIP_TRY(hipEventRecord(side->fork, s), ICICLE_UNKNOWN_ERROR);
    HIP_TRY(hipStreamWaitEvent(sl, side->fork, 0), ICICLE_UNKNOWN_ERROR);
  }
  if (prof) (void)hipEventRecord(prof->ev[1], s);
  AccumulateLauncher<C>::launch(pl, d_points, mont_pt, skip_below, stride, s, buckets, into ? 1 : 0, resident);
  ICICLE_TRY(check_launch("msm_accumulate"));
  if (prof) (void)hipEventRecord(prof->ev[2], s);
  const uint32_t lb = sizeof(X) > 128 ? 128 : 256;
  const size_t lds_l = lb * sizeof(typename Lazy<C>::type::X); // 36 KiB
  HIP_TRY(item_partials.alloc(pl->item_cap, s), ICICLE_ALLOCATION_FAILED);
  hipLaunchKernelGGL((msm_accumulate_large_kernel<C>), dim3(1024), dim3(lb), lds_l, sl, d_points, pl->sorted, pl->offsets, pl->counts, pl->n_large, pl->large_items, pl->item_cap, skip_below, stride, pl->g.tab ? pl->g.IB : 0, mont_pt, item_partials.p);
  hipLaunchKernelGGL((msm_combine_large_kernel<C>), dim3(256), dim3(lb), lds_l, sl, pl->counts, pl->n_large, pl->large_list, pl->large_first, item_partials.p, into ? 1 : 0, buckets);
  ICICLE_TRY(check_launch("msm_accumulate_large"));
  if (on_side) { // whatever follows on s (the reduction; the arena's next user of item_partials) is behind the side stream's kernels
    HIP_TRY(hipEventRecord(side->join, sl), ICICLE_UNKNOWN_ERROR);
    HIP_TRY(hipStreamWaitEvent(s, side->join, 0), ICICLE_UNKNOWN_ERROR);
  }
  return ICICLE_SUCCESS;
}

// stage 5 for one base set: the bucket array → partial sums for the host tail (geometry and tickets of `pl`)
template <class C>
eIcicleError msm_reduce_stage(const SortPlan* pl, hipStream_t s, const typename C::X* buckets, typename C::X* d_partials, int ticket_slot = 0)
{
  typedef typename C::X X;
  const MsmGeom& g = pl->g;
  typedef typename Lazy<C>::type::X LX;
  WsScoped<uint32_t> own_tickets;
  uint32_t* tickets = nullptr;
  const bool plan_tickets = ticket_slot >= 0 && ticket_slot < MSM_TICKET_SLOTS && pl->tickets && g.Wb <= 64;
  if (g.tab) {
    // table mode: bit-plane tree over the single bucket set (msm_zeta_reduce_kernel) → [T | S_0 … S_{t−1}] for the host tail
    const uint32_t n1 = pl->nbuckets, nblk1 = (n1 + ZR_M - 1) / ZR_M, nblk2 = (nblk1 + ZR_M - 1) / ZR_M;
    int nbits = 0;
    while ((1u << nbits) < n1) nbits++;
    if (nblk2 > 8 || (1u << nbits) != n1) {
      set_last_error("msm: bucket set of %u is outside the table-mode reduction's range", n1);
      return ICICLE_INVALID_ARGUMENT;
    }
    WsScoped<X> r1, r2;
    HIP_TRY(r1.alloc((size_t)ZR_OUT * nblk1, s), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(r2.alloc((size_t)ZR_OUT * ZR_OUT * nblk2, s), ICICLE_ALLOCATION_FAILED);
    if (plan_tickets) tickets = pl->tickets + ticket_slot * 64; // zeroed by the sort (a memset here is one more launch on a latency chain)
    else {
      HIP_TRY(own_tickets.alloc(1, s), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemsetAsync(own_tickets.p, 0, sizeof(uint32_t), s), ICICLE_UNKNOWN_ERROR);
      tickets = own_tickets.p;
    }
    const size_t lds_z = (size_t)ZR_M * sizeof(LX);
    allow_big_lds(msm_zeta_reduce_kernel<C>, lds_z);
    hipLaunchKernelGGL((msm_zeta_reduce_kernel<C>), dim3(nblk1, 1), dim3(ZR_T), lds_z, s, buckets, n1, r1.p, (uint32_t*)nullptr, 0, (X*)nullptr);
    hipLaunchKernelGGL((msm_zeta_reduce_kernel<C>), dim3(nblk2, ZR_OUT), dim3(ZR_T), lds_z, s, r1.p, nblk1, r2.p, tickets, nbits, d_partials);
    ICICLE_TRY(check_launch("msm_zeta_reduce"));
    return ICICLE_SUCCESS;
  }
  const ReduceShape rs = reduce_shape<X>(g);
  WsScoped<X> raw;
  if (rs.bpw > 1) {
    // per-workgroup results + one ticket counter per window: the workgroup that finishes a window last folds it
    HIP_TRY(raw.alloc((size_t)g.Wb * rs.bpw, s), ICICLE_ALLOCATION_FAILED);
    if (plan_tickets) tickets = pl->tickets + ticket_slot * 64;
    else {
      HIP_TRY(own_tickets.alloc((size_t)g.Wb, s), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemsetAsync(own_tickets.p, 0, (size_t)g.Wb * sizeof(uint32_t), s), ICICLE_UNKNOWN_ERROR);
      tickets = own_tickets.p;
    }
  }
  const size_t lds_r = (size_t)rs.rblock * sizeof(LX);
  allow_big_lds(msm_bucket_reduce_kernel<C>, lds_r);
  hipLaunchKernelGGL((msm_bucket_reduce_kernel<C>), dim3(rs.bpw, g.Wb), dim3(rs.rblock), lds_r, s, buckets, g.NBb, rs.k_log, d_partials, raw.p, tickets);
  ICICLE_TRY(check_launch("msm_bucket_reduce"));
  return ICICLE_SUCCESS;
}

// stages 4, 4b, 5 for one base set
template <class C>
eIcicleError msm_buckets_run(const SortPlan* pl, const typename C::A* d_points, int mont_pt, uint32_t skip_below, uint32_t stride, hipStream_t s, typename C::X* d_partials, MsmProfile* prof, int ticket_slot = 0, const LargeSide* side = nullptr)
{
  WsScoped<typename C::X> buckets;
  HIP_TRY(buckets.alloc(pl->nbuckets, s), ICICLE_ALLOCATION_FAILED);
  ICICLE_TRY(msm_accumulate_stage<C>(pl, d_points, mont_pt, skip_below, stride, s, buckets.p, false, prof, false, side));
  return msm_reduce_stage<C>(pl, s, buckets.p, d_partials, ticket_slot);
}

// host tail, table mode: part = [T | S_0 … S_{t−1}] of msm_zeta_reduce_kernel;  Σ_b (b+1)·B_b = T + Σ_j 2^j·S_j (Horner from the top bit)
template <class C>
typename C::P msm_host_tail_tab(const typename C::X* part, uint32_t nbits)
{
  typedef typename C::X X;
  X acc = C::x_zero();
  for (int j = (int)nbits - 1; j >= 0; j--) {
    acc = C::x_dbl(acc);
    acc = C::x_add(acc, part[1 + j]);
  }
  return C::p_from_mont(C::x_to_projective(C::x_add(acc, part[0])));
}

// host tail: Σ partials per window, Horner, standard-form projective (identity → (0,1,0))
template <class C>
typename C::P msm_host_tail(const typename C::X* part, uint32_t W, uint32_t bpw, int c, int wide)
{
  typedef typename C::X X;
  X acc = C::x_zero();
  for (int w = (int)W - 1; w >= 0; w--) {
    for (int j = 0; j < (w < wide ? c : c - 1); j++) acc = C::x_dbl(acc); // window w + 1 starts where window w ends
    X ws = C::x_zero();
    for (uint32_t k = 0; k < bpw; k++) ws = C::x_add(ws, part[(size_t)w * bpw + k]);
    acc = C::x_add(acc, ws);
  }
  return C::p_from_mont(C::x_to_projective(acc));
}

// Horner over the window sums on the HOST, in stream order.  The chain of c·(W−1) ≈ 240 dependent doublings takes one
// GPU lane 2.5 ms (G1) / 9 ms (G2) — a host core does it in 0.1–0.3 ms.  The W partial sums (≤ 16 KiB) are copied to a
// pinned slot, a host function enqueued with hipLaunchHostFunc computes the result there, and (for a device-resident
// result) a 96/192-byte copy brings it back; everything stays asynchronous on the caller's stream.
struct alignas(64) TailSlot {
  unsigned char partials[64 * 256]; // W ≤ 64 windows (c ≥ 4) of ≤ 256-byte XYZZ
  unsigned char result[192];
  void* host_dst; // result requested in host memory: written by the host function itself
  int W, c, wide;
  int tab_nbits; // > 0: partials = [T | S_0 … S_{tab_nbits−1}] of the table-mode reduction
};
// Ring of pinned slots.  A slot is in use from tail_slot_acquire() until the event its user records behind the last
// stream operation that touches it (tail_slot_commit) has completed: a caller that keeps more than TAIL_SLOTS
// asynchronous MSMs in flight gets nullptr and falls back to the device tail instead of overwriting a slot whose copy
// or host function is still pending.
constexpr int TAIL_SLOTS = 128;
struct TailRing {
  std::mutex mu;
  TailSlot* slots = nullptr;
  hipEvent_t ev[TAIL_SLOTS] = {};
  unsigned char state[TAIL_SLOTS] = {}; // 0 never used / known free, 1 acquired (no event yet), 2 event recorded, 3 event lost and stream not drained
  unsigned next = 0;
  bool failed = false;
};
// one ring per device: a slot's event is recorded on the stream of the MSM that uses it, and an event only records on a
// stream of the device it was created on
constexpr int TAIL_RING_DEVICES = 16;
inline TailRing& tail_ring(int dev)
{
  static TailRing r[TAIL_RING_DEVICES];
  return r[dev >= 0 && dev < TAIL_RING_DEVICES ? dev : 0];
}
inline TailRing& tail_ring_of(const TailSlot* t)
{
  for (int d = 0; d < TAIL_RING_DEVICES; d++) {
    TailRing& r = tail_ring(d);
    if (r.slots && t >= r.slots && t < r.slots + TAIL_SLOTS) return r;
  }
  return tail_ring(0);
}
inline TailSlot* tail_slot_acquire()
{
  int dev = 0;
  (void)hipGetDevice(&dev);
  TailRing& r = tail_ring(dev);
  std::lock_guard<std::mutex> lk(r.mu);
  if (r.failed) return nullptr;
  if (!r.slots) {
    TailSlot* p = nullptr;
    if (hipHostMalloc((void**)&p, sizeof(TailSlot) * TAIL_SLOTS, hipHostMallocPortable) != hipSuccess) {
      r.failed = true;
      return nullptr;
    }
    for (int i = 0; i < TAIL_SLOTS; i++)
      if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) {
        r.failed = true;
        return nullptr;
      }
    r.slots = p;
  }
  for (int k = 0; k < TAIL_SLOTS; k++) {
    const unsigned i = (r.next + k) % TAIL_SLOTS;
    if (r.state[i] == 1 || r.state[i] == 3) continue;
    if (r.state[i] == 2 && hipEventQuery(r.ev[i]) != hipSuccess) continue; // still in flight
    r.state[i] = 1;
    r.next = i + 1;
    return r.slots + i;
  }
  return nullptr;
}
// all stream work that uses the slot has been enqueued on s.  If the event cannot be recorded the slot's work may still
// be in flight with nothing to poll: the stream is drained before the slot is handed out again (state 3 = lost until then).
inline void tail_slot_commit(TailSlot* t, hipStream_t s)
{
  TailRing& r = tail_ring_of(t);
  std::lock_guard<std::mutex> lk(r.mu);
  const size_t i = (size_t)(t - r.slots);
  if (hipEventRecord(r.ev[i], s) == hipSuccess) {
    r.state[i] = 2;
    return;
  }
  (void)hipGetLastError();
  r.state[i] = 3;
  if (hipStreamSynchronize(s) == hipSuccess) r.state[i] = 0;
}
template <class C>
void host_tail_callback(void* ud)
{
  TailSlot* t = (TailSlot*)ud;
  typename C::P p = t->tab_nbits > 0 ? msm_host_tail_tab<C>((const typename C::X*)t->partials, (uint32_t)t->tab_nbits)
                                     : msm_host_tail<C>((const typename C::X*)t->partials, (uint32_t)t->W, 1, t->c, t->wide);
  memcpy(t->result, &p, sizeof p);
  if (t->host_dst) memcpy(t->host_dst, &p, sizeof p);
}

// the extern "C" entry (bn254_msm / bn254_g2_msm): sort + bucket stages + tail.
// batch_size > 1 (msm.h:21-53): `batch_size` scalar vectors of msm_size elements, one result each; the bases are
// shared (are_points_shared_in_batch) or one set per batch element.  The batch elements run back to back on the
// caller's stream.  precompute_factor f > 1: `bases` came from msm_precompute_bases and holds f points per original
// base, [f·i] being the base itself; this backend reads only those (stride f) — the extra multiples trade memory for a
// cheaper bucket reduction in the reference's backends, which is not where the time goes here.
constexpr uint32_t MSM_AUTO_TABLE_MIN_L = 1u << 15; // smaller MSMs are launch-bound either way
template <class C, class F, class AT, class PT>
eIcicleError msm_impl(const bn254_scalar_t* scalars, const AT* bases, int msm_size, const MSMConfig* cfg, PT* results)
{
  typedef typename C::A A;
  typedef typename C::X X;
  typedef typename C::P P;
  static_assert(sizeof(A) == sizeof(AT) && sizeof(P) == sizeof(PT), "ABI layout");
  if (!cfg || !results || (msm_size > 0 && (!scalars || !bases))) return ICICLE_INVALID_POINTER;
  if (msm_size < 0 || cfg->batch_size < 0 || cfg->precompute_factor < 0 || cfg->bitsize < 0 || cfg->bitsize > 254) return ICICLE_INVALID_ARGUMENT;
  ICICLE_TRY(require_device());
  hipStream_t s = (hipStream_t)cfg->stream;
  const uint32_t L = (uint32_t)msm_size;
  const uint32_t batch = cfg->batch_size > 1 ? (uint32_t)cfg->batch_size : 1;
  const uint32_t stride = cfg->precompute_factor > 1 ? (uint32_t)cfg->precompute_factor : 1;
  const bool shared = cfg->are_points_shared_in_batch || batch == 1;
  const bool profile = getenv("ICICLE_SNARK_PROFILE") != nullptr;

  Staged ss, sb;
  ICICLE_TRY(ss.in(scalars, (size_t)L * batch * sizeof(fe), cfg->are_scalars_on_device, s));
  ICICLE_TRY(sb.in(bases, (size_t)L * stride * (shared ? 1 : batch) * sizeof(A), cfg->are_points_on_device, s));

  int lbf = 0; // 0 = this library's default (msm_sort_run)
  ext_get_int(cfg->ext, "large_bucket_factor", &lbf);
  // automatic fixed-base table (msm_plan.h): one MSM of full-width scalars over a device-resident base array this runtime tracks.
  // The table is used for the bases' CURRENT contents only: hash sum of the array in stream order + guarded in-place refresh.
  BaseTableRef tref;
  bool use_table = false;
  struct Unpin {
    uint64_t id = 0;
    hipStream_t s = nullptr;
    ~Unpin() { base_table_unpin(id, s); } // after the last kernel of this call has been enqueued (hipFree waits for enqueued work)
  } unpin;
  unpin.s = s;
  WsScoped<unsigned long long> cur_sum;
  if (batch == 1 && stride == 1 && cfg->c <= 0 && (cfg->bitsize == 0 || cfg->bitsize == 254) && cfg->are_points_on_device && L >= MSM_AUTO_TABLE_MIN_L &&
      is_tracked_device_ptr(bases)) {
    const MsmGeom gt = msm_geometry(L, 0, 1);
    if (gt.tab) {
      const size_t table_bytes = (size_t)L * gt.W * sizeof(A);
      const int form = cfg->are_points_montgomery_form ? 1 : 0;
      const BaseTableState st = base_table_lookup(bases, (size_t)L * sizeof(A), L, sizeof(A) > 64, form, table_bytes, &tref);
      if (st == BASE_TABLE_BUILD) {
        void* table = nullptr;
        unsigned long long* sums = nullptr;
        if (hipMalloc((void**)&sums, 64) == hipSuccess && bases_hash_sum(bases, (size_t)L * sizeof(A), sums, s) == hipSuccess &&
            build_table_run<C, F>(bases, L, form, gt, s, &table, /*async=*/true) == ICICLE_SUCCESS) {
          base_table_publish(bases, L, sizeof(A) > 64, form, table, table_bytes, gt, sums, s, &tref);
          unpin.id = tref.id;
          use_table = true;
        } else {
          (void)hipGetLastError(); // no memory for the table: the classic layout still works
          if (sums) {
            (void)hipStreamSynchronize(s);
            (void)hipFree(sums);
          }
        }
      } else if (st == BASE_TABLE_HIT) {
        unpin.id = tref.id;
        use_table = true;
        if (tref.built) HIP_TRY(hipStreamWaitEvent(s, tref.built, 0), ICICLE_UNKNOWN_ERROR); // built on another stream, perhaps
        // … and behind the kernels of the previous call that read the table, whatever stream they are on: the refresh below may
        // rewrite rows in place (round-4 advisor: only the calling stream's own order protected them)
        if (tref.used) HIP_TRY(hipStreamWaitEvent(s, tref.used, 0), ICICLE_UNKNOWN_ERROR);
        // the caller may have rewritten its bases by means this library does not see (own kernels, raw hipMemcpy): hash them
        // now, in stream order, and let the guarded refresh bring the table up to date when the sum has moved
        HIP_TRY(cur_sum.alloc(8, s), ICICLE_ALLOCATION_FAILED);
        HIP_TRY(bases_hash_sum(bases, (size_t)L * sizeof(A), cur_sum.p, s), ICICLE_UNKNOWN_ERROR);
        hipLaunchKernelGGL((msm_table_refresh_kernel<C, F>), dim3((L + 127) / 128), dim3(128), 0, s, (const A*)bases, L, form, tref.g.c, tref.g.W, tref.g.wide, (A*)const_cast<void*>(tref.table),
                           tref.sums, cur_sum.p);
        hipLaunchKernelGGL(bases_hash_commit_kernel, dim3(1), dim3(1), 0, s, tref.sums, cur_sum.p);
        ICICLE_TRY(check_launch("msm_table_refresh"));
      }
    }
  }
  MsmProfile* prof = nullptr;
  for (uint32_t bi = 0; bi < batch; bi++) {
    prof = msm_profile_next();
    SortPlan pl;
    (void)hipEventRecord(prof->ev[0], s);
    ICICLE_TRY(msm_sort_run(ss.ptr<fe>() + (size_t)bi * L, L, cfg->c, lbf, cfg->are_scalars_montgomery_form, s, &pl, use_table ? 1 : 0, cfg->bitsize, (int)stride));
    if (use_table && (pl.g.tab != tref.g.tab || pl.g.c != tref.g.c || pl.g.W != tref.g.W)) return ICICLE_UNKNOWN_ERROR; // (same L, same rule: cannot differ)
    (void)hipEventRecord(prof->ev[4], s);
    prof->has_sort_end = true;
    prof->L = L;
    prof->nbuckets = pl.nbuckets;
    prof->c = pl.g.c;
    prof->W = pl.g.W;
    prof->is_g2 = sizeof(A) > 64;
    WsScoped<X> partials;
    uint32_t tail_w = 0;
    const size_t part_bytes = msm_partials_bytes(&pl, sizeof(A) > 64, &tail_w, nullptr);
    const int Wt = (int)(part_bytes / sizeof(X)); // sums left for the tail: one per window (= W, or ⌈W / f⌉ with precomputed bases), or 1 + t in table mode
    HIP_TRY(partials.alloc((size_t)Wt, s), ICICLE_ALLOCATION_FAILED);
    if (use_table) {
      ICICLE_TRY(msm_buckets_run<C>(&pl, (const A*)tref.table, 2, 0, L, s, partials.p, prof));
    } else {
      const A* pts = sb.ptr<A>() + (shared ? 0 : (size_t)bi * L * stride);
      ICICLE_TRY(msm_buckets_run<C>(&pl, pts, cfg->are_points_montgomery_form, 0, 1, s, partials.p, prof)); // the sort entries index the (precomputed) base array directly
    }
    if (cfg->are_results_on_device) note_device_write(results + bi, sizeof(P));
    TailSlot* slot = Wt <= 64 ? tail_slot_acquire() : nullptr;
    if (slot) {
      slot->W = Wt;
      slot->c = pl.g.c;
      slot->wide = pl.g.wide < Wt ? pl.g.wide : Wt;
      slot->tab_nbits = use_table ? (int)tail_w : 0;
      slot->host_dst = cfg->are_results_on_device ? nullptr : (void*)(results + bi);
      hipError_t he = hipMemcpyAsync(slot->partials, partials.p, (size_t)Wt * sizeof(X), hipMemcpyDeviceToHost, s);
      if (he == hipSuccess) he = hipLaunchHostFunc(s, host_tail_callback<C>, slot);
      if (he == hipSuccess && cfg->are_results_on_device) he = hipMemcpyAsync(results + bi, slot->result, sizeof(P), hipMemcpyHostToDevice, s);
      tail_slot_commit(slot, s); // on every path: the slot is free again once what was enqueued has run
      HIP_TRY(he, ICICLE_COPY_FAILED);
    } else if (use_table) {
      // no pinned slot (more than TAIL_SLOTS asynchronous MSMs in flight): wait for the sums and finish here
      std::vector<X> hp((size_t)Wt);
      HIP_TRY(hipMemcpyAsync(hp.data(), partials.p, (size_t)Wt * sizeof(X), hipMemcpyDeviceToHost, s), ICICLE_COPY_FAILED);
      HIP_TRY(hipStreamSynchronize(s), ICICLE_SYNCHRONIZATION_FAILED);
      const P p = msm_host_tail_tab<C>(hp.data(), tail_w);
      HIP_TRY(hipMemcpy(results + bi, &p, sizeof(P), cfg->are_results_on_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost), ICICLE_COPY_FAILED);
    } else {
      // no pinned slot: single-lane Horner on the device (2.5 ms G1 / 9 ms G2)
      WsScoped<P> dres;
      P* dst = reinterpret_cast<P*>(results + bi);
      if (!cfg->are_results_on_device) {
        HIP_TRY(dres.alloc(1, s), ICICLE_ALLOCATION_FAILED);
        dst = dres.p;
      }
      hipLaunchKernelGGL((msm_tail_kernel<C>), dim3(1), dim3(64), 0, s, partials.p, Wt, 1, pl.g.c, pl.g.wide < Wt ? pl.g.wide : Wt, dst);
      ICICLE_TRY(check_launch("msm_tail"));
      if (!cfg->are_results_on_device) {
        HIP_TRY(hipMemcpyAsync(results + bi, dres.p, sizeof(P), hipMemcpyDeviceToHost, s), ICICLE_COPY_FAILED);
        HIP_TRY(hipStreamSynchronize(s), ICICLE_SYNCHRONIZATION_FAILED);
      }
    }
    (void)hipEventRecord(prof->ev[3], s);
    prof->valid = true;
  }
  if (profile && prof) {
    HIP_TRY(hipStreamSynchronize(s), ICICLE_SYNCHRONIZATION_FAILED);
    (void)hipEventElapsedTime(&g_last_msm_ms[0], prof->ev[0], prof->ev[1]);
    (void)hipEventElapsedTime(&g_last_msm_ms[1], prof->ev[1], prof->ev[2]);
    (void)hipEventElapsedTime(&g_last_msm_ms[2], prof->ev[2], prof->ev[3]);
    (void)hipEventElapsedTime(&g_last_msm_ms[3], prof->ev[0], prof->ev[3]);
  }
  return end_call(s, cfg->is_async);
}

// msm_precompute_bases (msm.h, icicle/src/msm.cpp:45-72): output[f·i + j] = 2^(j·shift)·P_i, j < f, with
// shift = c·⌈W / f⌉ for this backend's window geometry of an MSM of nof_bases elements (cfg->c if given).
// Input and output honour are_points_montgomery_form / are_points_on_device (the output flag is are_results_on_device).
template <class C>
__global__ __launch_bounds__(256) void precompute_kernel(const typename C::A* __restrict__ in, uint32_t n, int f, int shift, int mont, typename C::P* __restrict__ out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename C::A a = in[i];
  const bool zero = C::aff_is_zero(a);
  if (!mont) a = C::aff_to_mont(a);
  typename C::X x = zero ? C::x_zero() : C::x_from_affine(a);
  for (int j = 0; j < f; j++) {
    out[(size_t)i * f + j] = C::x_to_projective(x); // Montgomery projective; identity → (0, 1, 0)
    if (j + 1 < f)
      for (int k = 0; k < shift; k++) x = C::x_dbl(x);
  }
}
template <class A>
__global__ __launch_bounds__(256) void affine_to_mont_kernel(A* pts, uint64_t ncoord)
{
  fe* c = reinterpret_cast<fe*>(pts);
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncoord) c[i] = Fq::to_mont(c[i]);
}
template <class C, class F, class AT>
eIcicleError precompute_impl(const AT* bases, int nof_bases, const MSMConfig* cfg, AT* out)
{
  typedef typename C::A A;
  typedef typename C::P P;
  if (!cfg || (nof_bases > 0 && (!bases || !out))) return ICICLE_INVALID_POINTER;
  if (nof_bases < 0) return ICICLE_INVALID_ARGUMENT;
  ICICLE_TRY(require_device());
  const int f = cfg->precompute_factor > 1 ? cfg->precompute_factor : 1;
  const uint32_t n = (uint32_t)nof_bases;
  hipStream_t s = (hipStream_t)cfg->stream;
  Staged sb, so;
  ICICLE_TRY(sb.in(bases, (size_t)n * sizeof(A), cfg->are_points_on_device, s));
  ICICLE_TRY(so.out(out, (size_t)n * f * sizeof(A), cfg->are_results_on_device, s));
  if (n) {
    const MsmGeom g = msm_geometry(n, cfg->c, 0, cfg->bitsize, f);
    const int shift = g.c * g.nbms;
    const uint64_t m = (uint64_t)n * f;
    WsScoped<P> proj;
    WsScoped<typename F::T> scratch;
    HIP_TRY(proj.alloc(m, s), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(scratch.alloc(m, s), ICICLE_ALLOCATION_FAILED);
    hipLaunchKernelGGL((precompute_kernel<C>), dim3((n + 255) / 256), dim3(256), 0, s, sb.ptr<A>(), n, f, shift, cfg->are_points_montgomery_form ? 1 : 0, proj.p);
    ICICLE_TRY(check_launch("msm_precompute"));
    const int chunk = 32;
    const uint64_t nthreads = (m + chunk - 1) / chunk;
    hipLaunchKernelGGL((batch_to_affine_kernel<C, F>), dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, s, proj.p, m, chunk, so.ptr<A>(), scratch.p);
    ICICLE_TRY(check_launch("batch_to_affine"));
    if (cfg->are_points_montgomery_form) {
      const uint64_t nc = m * (sizeof(A) / sizeof(fe));
      hipLaunchKernelGGL((affine_to_mont_kernel<A>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, so.ptr<A>(), nc);
      ICICLE_TRY(check_launch("affine_to_mont"));
    }
  }
  ICICLE_TRY(so.finish());
  return end_call(s, cfg->is_async);
}

template <class C, class F, class AT>
eIcicleError generator_mul_impl(const bn254_scalar_t* sc, uint64_t n, hipStream_t s, AT* out, const typename C::A& gen_std)
{
  typedef typename C::X X;
  typedef typename C::P P;
  typedef typename C::A A;
  if (!sc || !out) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  if (n == 0) return ICICLE_SUCCESS;
  note_device_write(out, (size_t)n * sizeof(A));
  WsScoped<X> table;
  WsScoped<P> proj;
  WsScoped<typename F::T> scratch;
  HIP_TRY(table.alloc(32 * 255, s), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(proj.alloc(n, s), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(scratch.alloc(n, s), ICICLE_ALLOCATION_FAILED);
  A gm = C::aff_to_mont(gen_std);
  hipLaunchKernelGGL((fixed_base_table_kernel<C>), dim3(1), dim3(256), 0, s, gm, table.p);
  ICICLE_TRY(check_launch("fixed_base_table"));
  hipLaunchKernelGGL((fixed_base_mul_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const fe*>(sc), n, table.p, proj.p);
  ICICLE_TRY(check_launch("fixed_base_mul"));
  const int chunk = 32;
  const uint64_t nthreads = (n + chunk - 1) / chunk;
  hipLaunchKernelGGL((batch_to_affine_kernel<C, F>), dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, s, proj.p, n, chunk, reinterpret_cast<A*>(out), scratch.p);
  ICICLE_TRY(check_launch("batch_to_affine"));
  return ICICLE_SUCCESS;
}

} // namespace
