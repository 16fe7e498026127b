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

#ifndef ACC_MIN_WAVES
#define ACC_MIN_WAVES 1
#endif
#ifndef REDUCE_MIN_BLOCKS
#define REDUCE_MIN_BLOCKS 1 // experiment hook (msm_g2.hip: -DG2_REDUCE_MIN_BLOCKS=2 caps the G2 reduction at 256 VGPRs)
#endif
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

template <class C>
__global__ __launch_bounds__(256, ACC_MIN_WAVES) void msm_accumulate_kernel(const typename C::A* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ counts, const uint32_t* __restrict__ order, uint32_t nbuckets, uint32_t large_thr, uint32_t skip_below, uint32_t stride, int ib, int form,
                                                              typename C::X* __restrict__ buckets)
{
  typedef typename Lazy<C>::type CL;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbuckets) return;
  const uint32_t b = order[t]; // neighbouring lanes own buckets of (nearly) equal size
  const uint32_t cnt = counts[b];
  if (cnt > large_thr) return; // step 4b
  const uint32_t* idx = sorted + offsets[b];
  typename CL::X acc = CL::x_zero();
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
                                                                 const uint32_t* __restrict__ large_first, const typename C::X* __restrict__ item_partials, typename C::X* __restrict__ buckets)
{
  typedef typename Lazy<C>::type CL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typename CL::X* sh = reinterpret_cast<typename CL::X*>(smem);
  const uint32_t nl = n_large[0];
  for (uint32_t li = blockIdx.x; li < nl; li += gridDim.x) {
    const uint32_t b = large_list[li];
    const uint32_t nch = (counts[b] + MSM_LARGE_CHUNK - 1) / MSM_LARGE_CHUNK, first = large_first[li];
    if (nch == 1) { // one chunk (most buckets just above the threshold): nothing to sum
      if (threadIdx.x == 0) buckets[b] = item_partials[first];
      continue;
    }
    typename CL::X acc = CL::x_zero();
    for (uint32_t k = threadIdx.x; k < nch; k += blockDim.x) acc = CL::x_add(acc, CL::x_load_internal(item_partials[first + k]));
    int width = 1;
    while (width < (int)blockDim.x && (uint32_t)width < nch) width <<= 1;
    acc = block_reduce_lazy<C>(acc, sh, width);
    if (threadIdx.x == 0) buckets[b] = CL::x_store_internal(acc);
    __syncthreads();
  }
}

// two tree sums side by side over n ≤ blockDim entries (n a power of two): the lower half of the workgroup folds sa, the
// upper half sb (only if `both`) — log₂ n additions on the chain.  Results in sa[0], sb[0]; ends with a barrier.
template <class C>
__device__ __forceinline__ void block_dual_tree(typename Lazy<C>::type::X* sa, typename Lazy<C>::type::X* sb, int n, bool both)
{
  typedef typename Lazy<C>::type CL;
  const int half = n >> 1;
  const bool upper = half && (int)threadIdx.x >= half;
  typename CL::X* arr = upper ? sb : sa;
  const int i = upper ? (int)threadIdx.x - half : (int)threadIdx.x;
  for (int st = half; st > 0; st >>= 1) {
    if (i < st && (both || !upper)) arr[i] = CL::x_add(arr[i], arr[i + st]);
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

// Σ_b (b+1)·B_b per window.  grid = (workgroups per window, W); each thread owns K = 2^k_log buckets.
// Buckets arrive in the internal encoding; the window sums leave as ec.h XYZZ (Montgomery R = 2^256) for the tails:
// out = [Σ_b (b+1)·B_b | Σ_b B_b (emit_line: table mode, the windows are slices of ONE bucket set and the tail needs both)].
// raw / tickets: scratch for the per-workgroup results (internal encoding) and one zeroed counter per window, used when
// gridDim.x > 1.
template <class C>
__global__ __launch_bounds__(256, REDUCE_MIN_BLOCKS) void msm_bucket_reduce_kernel(const typename C::X* __restrict__ buckets, uint32_t NB, int k_log, typename C::X* __restrict__ out, int emit_line,
                                                                typename C::X* raw, uint32_t* tickets)
{
  typedef typename Lazy<C>::type CL;
  typedef typename CL::X X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem);
  X* sb = sh + blockDim.x;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // thread within the window
  const uint32_t base = t << k_log;
  const typename C::X* B = buckets + (size_t)blockIdx.y * NB + base;
  {
    X line = CL::x_zero(), tri = CL::x_zero();
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
    sh[threadIdx.x] = tri;
    if (emit_line) sb[threadIdx.x] = line;
  }
  __syncthreads();
  block_dual_tree<C>(sh, sb, (int)blockDim.x, emit_line != 0);
  const size_t nw = gridDim.y, bpw = gridDim.x;
  if (bpw == 1) {
    if (threadIdx.x == 0) {
      out[blockIdx.y] = CL::x_store(sh[0]);
      if (emit_line) out[nw + blockIdx.y] = CL::x_store(sb[0]);
    }
    return;
  }
  if (threadIdx.x == 0) {
    raw[(size_t)blockIdx.y * bpw + blockIdx.x] = CL::x_store_internal(sh[0]);
    if (emit_line) raw[nw * bpw + (size_t)blockIdx.y * bpw + blockIdx.x] = CL::x_store_internal(sb[0]);
  }
  if (!last_workgroup_of_window(tickets)) return;
  if (threadIdx.x < bpw) {
    sh[threadIdx.x] = CL::x_load_internal(raw[(size_t)blockIdx.y * bpw + threadIdx.x]);
    if (emit_line) sb[threadIdx.x] = CL::x_load_internal(raw[nw * bpw + (size_t)blockIdx.y * bpw + threadIdx.x]);
  }
  __syncthreads();
  block_dual_tree<C>(sh, sb, (int)bpw, emit_line != 0);
  if (threadIdx.x == 0) {
    out[blockIdx.y] = CL::x_store(sh[0]);
    if (emit_line) out[nw + blockIdx.y] = CL::x_store(sb[0]);
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
  T inv = F::inv(run);
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
template <class C, class F>
eIcicleError build_table_run(const void* d_points, uint32_t n, int from_form, const MsmGeom& g, hipStream_t s, void** d_table)
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
    HIP_TRY(hipMalloc((void**)&rows, m * sizeof(P)), ICICLE_ALLOCATION_FAILED);
    if (hipMalloc((void**)&scratch, m * sizeof(typename F::T)) != hipSuccess) {
      (void)hipFree(rows);
      (void)hipFree(table);
      return ICICLE_ALLOCATION_FAILED;
    }
    hipLaunchKernelGGL((msm_table_rows_kernel<C>), dim3((n + 255) / 256), dim3(256), 0, s, (const A*)d_points, n, from_form, g.c, g.W, g.wide, rows);
    const int chunk = 32;
    const uint64_t nthreads = (m + chunk - 1) / chunk;
    hipLaunchKernelGGL((batch_to_affine_kernel<C, F>), dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, s, rows, m, chunk, table, scratch);
    hipLaunchKernelGGL((msm_points_to_internal_kernel<C>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, table, (uint32_t)m, 0);
    const eIcicleError e = check_launch("msm_build_table");
    const hipError_t he = hipStreamSynchronize(s);
    (void)hipFree(rows);
    (void)hipFree(scratch);
    if (e != ICICLE_SUCCESS || he != hipSuccess) {
      (void)hipFree(table);
      return e != ICICLE_SUCCESS ? e : ICICLE_SYNCHRONIZATION_FAILED;
    }
  }
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
  static void launch(const SortPlan* pl, const typename C::A* d_points, int mont_pt, uint32_t skip_below, uint32_t stride, hipStream_t s, typename C::X* buckets)
  {
    hipLaunchKernelGGL((msm_accumulate_kernel<C>), dim3((pl->nbuckets + 255) / 256), dim3(256), 0, s, d_points, pl->sorted, pl->offsets, pl->counts, pl->order, pl->nbuckets, pl->large_thr, skip_below, stride, pl->g.tab ? pl->g.IB : 0, mont_pt, buckets);
  }
};
#if defined(ISNARK_G2_ACC_EXTERN)
template <>
struct AccumulateLauncher<G2> {
  static void launch(const SortPlan* pl, const G2::A* d_points, int mont_pt, uint32_t skip_below, uint32_t stride, hipStream_t s, G2::X* buckets)
  {
    isnark::msm_g2_accumulate_launch(pl, d_points, mont_pt, skip_below, stride, s, buckets);
  }
};
#endif

// scalar vectors up to this length leave the GPU far from full (cf. the H stream rule of the prover): the bucket
// reduction is then a pure latency chain and the suffix-scan kernels shorten it (100 k constraints: 2.70 → 2.43 ms per
// prove); beyond, their blockDim·log₂ extra additions cost more than the chain saves (1.6 M: 16.85 → 17.2 ms)
constexpr uint32_t MSM_SCAN_REDUCE_MAX_L = 1u << 19;
struct ReduceShape {
  int k_log;
  uint32_t tpw, rblock, bpw;
  bool scan;  // small table-mode bucket set: suffix-scan kernels, factor M = rblock << k_log left to the host tail
  uint32_t M; // 0 unless scan && bpw > 1
  // large table-mode bucket set: row / column sums, then the scan kernel on the LO + HI sums (msm_reduce_rows_kernel …)
  bool two;
  uint32_t LO, HI, gy; // bucket b = h·LO + l;  gy = row groups of the column-sum kernel
};
constexpr uint32_t MSM_TWO_LEVEL_MARK = 0x80000000u; // msm_partials_bytes: M = MARK | LO tells the host tail which layout it gets
template <class X>
ReduceShape reduce_shape(const MsmGeom& g, uint32_t L = 0xffffffffu, int pref = 0)
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
  while (k < 4 && (uint64_t)g.Wb * (g.NBb >> k) / rb_max > 128) k++; // ≤ 128 partial sums per kind for the host tail
  // (8 buckets per thread — twice the threads — was measured for both curves: stand-in circuits 6.1 → 5.9 ms, benchmark/1600k
  //  16.3 → 16.7 ms: the reductions are latency chains that run beside the accumulations, whose SIMDs they would take)
  // the prover's LAST MSM (pref == 1: H) reduces on an otherwise idle GPU and its reduction is the tail of the prove: 8 buckets per thread
  // (a chain of 16 + 16 + 10 instead of 32 + 16 + 8 additions on twice the threads): −0.1 ms at 1.6 M constraints; 4 per thread: none
  if (pref == 1 && k > 3) k = 3;
  r.k_log = k < lnb ? k : lnb;
  for (;; r.k_log++) {
    r.tpw = g.NBb >> r.k_log;                            // reduce threads per (pseudo-)window
    r.rblock = r.tpw < rb_max ? r.tpw : rb_max;
    r.bpw = r.tpw / r.rblock;
    if (r.bpw <= r.rblock) break; // the workgroup that finishes a window last folds its bpw results with one thread each
  }
  static const uint32_t scan_max_l = getenv("ICICLE_SNARK_SCAN_REDUCE_MAX_L") ? (uint32_t)atoll(getenv("ICICLE_SNARK_SCAN_REDUCE_MAX_L")) : MSM_SCAN_REDUCE_MAX_L;
  r.scan = g.tab && L <= scan_max_l;
  r.M = r.scan && r.bpw > 1 ? r.rblock << r.k_log : 0;
  // Opt-in (ICICLE_SNARK_REDUCE_TWO_LEVEL=1): interleaved A/B runs of benchmark/1600k on MI355X put it within ±0.15 ms of
  // the single kernel (16.2–16.5 ms either way) — it does half the additions on 4–8× the threads, but its four dependent
  // launches are as long as the single kernel's chain when the GPU is otherwise idle (0.8 vs 0.5 ms at the end of a prove).
  static const bool two_cfg = getenv("ICICLE_SNARK_REDUCE_TWO_LEVEL") && atoi(getenv("ICICLE_SNARK_REDUCE_TWO_LEVEL")) != 0;
  const uint32_t nb_all = g.NBb * (uint32_t)g.Wb;
  int nbits = 0;
  while ((1u << nbits) < nb_all) nbits++;
  r.two = two_cfg && pref != 1 && g.tab && !r.scan && nbits >= 14 && nbits <= 20 && (1u << nbits) == nb_all;
  r.LO = r.HI = r.gy = 0;
  if (r.two) {
    r.LO = 1u << ((nbits + 1) / 2);
    r.HI = nb_all / r.LO;
    r.gy = r.HI / 32 ? r.HI / 32 : 1; // 4 row groups of ≤ 8 rows per workgroup
    if (r.gy > 16) r.gy = 16;
  }
  return r;
}

// Suffix scan + two concurrent tree sums over one workgroup, shared by the bucket reduction and the fold of its
// per-workgroup results.  In: sa[i] = t_i, sb[i] = l_i (i < n, n a power of two).  Out (valid in thread 0 after the
// call): sa[0] = Σ_i t_i,  sb[0] = Σ_{i ≥ 1} i·l_i  (as Σ_{i ≥ 1} suffix_i, suffix_i = Σ_{u ≥ i} l_u — no scalar
// multiplication), return value = Σ_i l_i.  Chain length: 2·log₂ n point additions (the lower half of the workgroup
// sums sa while the upper half sums sb).
template <class C>
__device__ __forceinline__ typename Lazy<C>::type::X block_weighted_sums(typename Lazy<C>::type::X* sa, typename Lazy<C>::type::X* sb, int n)
{
  typedef typename Lazy<C>::type CL;
  typedef typename CL::X X;
  const int tid = threadIdx.x;
  X suf = sb[tid];
  for (int d = 1; d < n; d <<= 1) {
    const bool has = tid + d < n;
    X o = suf;
    if (has) o = sb[tid + d];
    __syncthreads();
    if (has) {
      suf = CL::x_add(suf, o);
      sb[tid] = suf;
    }
    __syncthreads();
  }
  if (tid == 0) sb[0] = CL::x_zero(); // weight 0
  __syncthreads();
  const int half = n >> 1;
  const bool upper = half && tid >= half;
  X* arr = upper ? sb : sa;
  const int i = upper ? tid - half : tid;
  for (int st = half; st > 0; st >>= 1) {
    if (i < st) arr[i] = CL::x_add(arr[i], arr[i + st]);
    __syncthreads();
  }
  return suf; // thread 0: Σ l_i
}

// Bucket reduction for SMALL bucket sets in table mode (the GPU is far from full and only the length of the dependent
// chain counts), grid = (workgroups per window, windows), each thread owns K = 2^k_log consecutive buckets.
// Workgroup (x, y) covers the M = blockDim·K buckets i of window y behind x·M and emits
//   TRI = Σ_i (i+1)·B_i  (i local to the workgroup)   and   LINE = Σ_i B_i,
// so that the window's Σ_b (b+1)·B_b = Σ_x TRI_x + M·Σ_x x·LINE_x — the weights of whole threads / workgroups come
// from suffix sums (block_weighted_sums), not from a double-and-add per thread: a chain of 2K + 2·log₂(blockDim) + k_log
// additions instead of 2K + 1.5·log₂(first index) + 2·log₂(blockDim), the factor M is applied by the host tail.  The
// suffix scan costs blockDim·log₂(blockDim) additions per workgroup, which is why the large (work-bound) bucket sets
// stay with msm_bucket_reduce_kernel.
// One workgroup per window: out = [TRI | LINE][window] as ec.h XYZZ (Montgomery R = 2^256) for the tails.  Several: the
// workgroup that draws the window's last ticket folds the others' results (raw, internal encoding) into [TT | L | LL][window].
template <class C>
__global__ __launch_bounds__(256) void msm_bucket_reduce_scan_kernel(const typename C::X* __restrict__ buckets, uint32_t NB, int k_log, typename C::X* __restrict__ out, typename C::X* raw,
                                                                     uint32_t* tickets)
{
  typedef typename Lazy<C>::type CL;
  typedef typename CL::X X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sa = reinterpret_cast<X*>(smem);
  X* sb = sa + blockDim.x;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // thread within the window
  const typename C::X* B = buckets + (size_t)blockIdx.y * NB + ((size_t)t << k_log);
  {
    X line = CL::x_zero(), tri = CL::x_zero();
    for (int j = (1 << k_log) - 1; j >= 0; j--) {
      line = CL::x_add(line, CL::x_load_internal(B[j]));
      tri = CL::x_add(tri, line);
    }
    sa[threadIdx.x] = tri;
    sb[threadIdx.x] = line;
  }
  __syncthreads();
  X total = block_weighted_sums<C>(sa, sb, (int)blockDim.x);
  const size_t nw = gridDim.y, bpw = gridDim.x;
  if (threadIdx.x == 0) {
    X w = sb[0]; // Σ_t t·line_t; thread t's first bucket is t·K
    for (int j = 0; j < k_log; j++) w = CL::x_dbl(w);
    const X tri = CL::x_add(sa[0], w);
    if (bpw == 1) {
      out[blockIdx.y] = CL::x_store(tri);
      out[nw + blockIdx.y] = CL::x_store(total);
    } else {
      raw[(size_t)blockIdx.y * bpw + blockIdx.x] = CL::x_store_internal(tri);
      raw[nw * bpw + (size_t)blockIdx.y * bpw + blockIdx.x] = CL::x_store_internal(total);
    }
  }
  if (bpw == 1 || !last_workgroup_of_window(tickets)) return;
  // the last workgroup of the window: [TT | L | LL] = Σ_x TRI_x, Σ_x LINE_x, Σ_x x·LINE_x (the host applies M to LL)
  if (threadIdx.x < bpw) {
    sa[threadIdx.x] = CL::x_load_internal(raw[(size_t)blockIdx.y * bpw + threadIdx.x]);
    sb[threadIdx.x] = CL::x_load_internal(raw[nw * bpw + (size_t)blockIdx.y * bpw + threadIdx.x]);
  }
  __syncthreads();
  total = block_weighted_sums<C>(sa, sb, (int)bpw);
  if (threadIdx.x == 0) {
    out[blockIdx.y] = CL::x_store(sa[0]);
    out[nw + blockIdx.y] = CL::x_store(total);
    out[2 * nw + blockIdx.y] = CL::x_store(sb[0]);
  }
}

// ---- two-level bucket reduction (large table-mode sets) ------------------------------------------------------------------
// Σ_b (b+1)·B_b over ONE bucket set of NB = LO·HI buckets, b = h·LO + l:
//     Σ_b (b+1)·B_b = Σ_l (l+1)·C_l + LO·Σ_h h·R_h,    C_l = Σ_h B[h][l] (column sums),  R_h = Σ_l B[h][l] (row sums).
// The 2·NB additions of the row and column sums are plain sums — no per-thread scalar multiplication, chains of ≈ 12
// additions on 128 K – 256 K threads — and the two weighted sums that remain have LO and HI ≤ 1024 terms: the suffix-scan
// kernel (msm_bucket_reduce_scan_kernel, one workgroup each) finishes them.  msm_bucket_reduce_kernel gives every thread 16
// buckets, a running-sum pair AND a double-and-add by its first index (≈ 30 further point operations on the chain): 32 K
// threads, 0.49 ms for a G1 set of 2^19 buckets and 1.8 ms for G2 — latency-bound kernels whose 392-register G2 waves also
// kept the accumulation of the next MSM off the SIMDs they sat on.
// rows: grid = HI workgroups; a workgroup sums the LO consecutive buckets of row h (threads own LO / blockDim consecutive ones)
template <class C>
__global__ __launch_bounds__(256) void msm_reduce_rows_kernel(const typename C::X* __restrict__ buckets, uint32_t LO, typename C::X* __restrict__ sums /* [LO + h] */)
{
  typedef typename Lazy<C>::type CL;
  typedef typename CL::X X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem);
  const uint32_t per = LO / blockDim.x;
  const typename C::X* B = buckets + (size_t)blockIdx.x * LO + (size_t)threadIdx.x * per;
  X acc = CL::x_load_internal(B[0]);
  for (uint32_t j = 1; j < per; j++) acc = CL::x_add(acc, CL::x_load_internal(B[j]));
  acc = block_reduce_lazy<C>(acc, sh, (int)blockDim.x);
  if (threadIdx.x == 0) sums[LO + blockIdx.x] = CL::x_store_internal(acc);
}
// columns: grid = (LO / 64, gy); a workgroup of 4 waves covers 64 columns × (HI / gy) rows — wave q the rows of its quarter,
// lane l one column (a row's 64 buckets are one contiguous 8 / 16 KiB read) — and emits 64 partial column sums
template <class C>
__global__ __launch_bounds__(256) void msm_reduce_cols_kernel(const typename C::X* __restrict__ buckets, uint32_t LO, uint32_t HI, typename C::X* __restrict__ partial /* [gy][LO] */)
{
  typedef typename Lazy<C>::type CL;
  typedef typename CL::X X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem); // [4][64]
  const uint32_t lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const uint32_t col = blockIdx.x * 64 + lane;
  const uint32_t rows_wg = HI / gridDim.y, rows_q = rows_wg / 4 ? rows_wg / 4 : 1;
  const uint32_t h0 = blockIdx.y * rows_wg + q * rows_q;
  X acc = CL::x_zero();
  if (q * rows_q < rows_wg) {
    acc = CL::x_load_internal(buckets[(size_t)h0 * LO + col]);
    for (uint32_t j = 1; j < rows_q; j++) acc = CL::x_add(acc, CL::x_load_internal(buckets[(size_t)(h0 + j) * LO + col]));
  }
  sh[q * 64 + lane] = acc;
  __syncthreads();
  if (q < 2) sh[q * 64 + lane] = acc = CL::x_add(acc, sh[(q + 2) * 64 + lane]);
  __syncthreads();
  if (q == 0) partial[(size_t)blockIdx.y * LO + col] = CL::x_store_internal(CL::x_add(acc, sh[64 + lane]));
}
// columns, last step: C_l = Σ_y partial[y][l]; workgroup = 16 columns × 16 slots (gy ≤ 16), tree over the slots
template <class C>
__global__ __launch_bounds__(256) void msm_reduce_cols_final_kernel(const typename C::X* __restrict__ partial, uint32_t LO, uint32_t gy, typename C::X* __restrict__ sums /* [l] */)
{
  typedef typename Lazy<C>::type CL;
  typedef typename CL::X X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem); // [16 slots][16 columns]
  const uint32_t c = threadIdx.x & 15, y = threadIdx.x >> 4;
  const uint32_t col = blockIdx.x * 16 + c;
  X acc = y < gy ? CL::x_load_internal(partial[(size_t)y * LO + col]) : CL::x_zero();
  sh[y * 16 + c] = acc;
  __syncthreads();
  for (uint32_t st = 8; st > 0; st >>= 1) {
    if (y < st) sh[y * 16 + c] = acc = CL::x_add(acc, sh[(y + st) * 16 + c]);
    __syncthreads();
  }
  if (y == 0) sums[col] = CL::x_store_internal(acc);
}
// zero the padding of the row-sum half when HI < LO (the scan kernel treats both halves as LO-bucket windows)
template <class C>
__global__ __launch_bounds__(256) void msm_reduce_pad_kernel(typename C::X* __restrict__ sums, uint32_t first, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sums[first + i] = C::x_zero();
}

// more than 64 KiB of dynamic LDS (two lazy XYZZ per thread) has to be allowed once per kernel
template <class K>
inline void allow_big_lds(K kernel, size_t bytes)
{
  if (bytes > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// stages 4, 4b, 5 for one base set
template <class C>
eIcicleError msm_buckets_run(const SortPlan* pl, const typename C::A* d_points, int mont_pt, uint32_t skip_below, uint32_t stride, hipStream_t s, typename C::X* d_partials, MsmProfile* prof, int ticket_slot = 0)
{
  typedef typename C::X X;
  const MsmGeom& g = pl->g;
  const ReduceShape rs = reduce_shape<X>(g, pl->L, pl->reduce_pref);
  WsScoped<X> buckets, item_partials;
  HIP_TRY(buckets.alloc(pl->nbuckets, s), ICICLE_ALLOCATION_FAILED);
  if (prof) (void)hipEventRecord(prof->ev[1], s);
  AccumulateLauncher<C>::launch(pl, d_points, mont_pt, skip_below, stride, s, buckets.p);
  ICICLE_TRY(check_launch("msm_accumulate"));
  if (prof) (void)hipEventRecord(prof->ev[2], s);
  const uint32_t lb = sizeof(X) > 128 ? 128 : 256;
  const size_t lds_l = lb * sizeof(typename Lazy<C>::type::X); // 36 KiB
  HIP_TRY(item_partials.alloc(pl->item_cap, s), ICICLE_ALLOCATION_FAILED);
  hipLaunchKernelGGL((msm_accumulate_large_kernel<C>), dim3(1024), dim3(lb), lds_l, s, d_points, pl->sorted, pl->offsets, pl->counts, pl->n_large, pl->large_items, pl->item_cap, skip_below, stride, pl->g.tab ? pl->g.IB : 0, mont_pt, item_partials.p);
  hipLaunchKernelGGL((msm_combine_large_kernel<C>), dim3(256), dim3(lb), lds_l, s, pl->counts, pl->n_large, pl->large_list, pl->large_first, item_partials.p, buckets.p);
  ICICLE_TRY(check_launch("msm_accumulate_large"));
  item_partials.release();
  WsScoped<X> raw;
  WsScoped<uint32_t> own_tickets;
  uint32_t* tickets = nullptr;
  if (rs.bpw > 1 && !rs.two) {
    // per-workgroup results + one ticket counter per window: the workgroup that finishes a window last folds it.  The
    // counters were zeroed with the sort's own (a memset here is one more launch on a latency chain)
    HIP_TRY(raw.alloc((size_t)g.Wb * rs.bpw * 2, s), ICICLE_ALLOCATION_FAILED);
    if (g.Wb <= 64 && ticket_slot >= 0 && ticket_slot < MSM_TICKET_SLOTS && pl->tickets) tickets = pl->tickets + ticket_slot * 64;
    else {
      HIP_TRY(own_tickets.alloc((size_t)g.Wb, s), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemsetAsync(own_tickets.p, 0, (size_t)g.Wb * sizeof(uint32_t), s), ICICLE_UNKNOWN_ERROR);
      tickets = own_tickets.p;
    }
  }
  typedef typename Lazy<C>::type::X LX;
  const size_t lds_r = 2 * (size_t)rs.rblock * sizeof(LX);
  if (rs.two) {
    // row sums | column sums → [C_0 … C_{LO−1} | R_0 … R_{HI−1}, 0 …] → the scan kernel on two LO-bucket windows:
    // out = [Σ(l+1)·C_l, Σ(h+1)·R_h | Σ C_l, Σ R_h]; the host tail forms TRI_C + LO·(TRI_R − LINE_R)
    WsScoped<X> sums, partial;
    HIP_TRY(sums.alloc(2 * (size_t)rs.LO, s), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(partial.alloc((size_t)rs.gy * rs.LO, s), ICICLE_ALLOCATION_FAILED);
    uint32_t rb = sizeof(X) > 128 ? 128 : 256; // LDS tree buffer ≤ 36 KiB
    if (rb > rs.LO) rb = rs.LO;
    hipLaunchKernelGGL((msm_reduce_rows_kernel<C>), dim3(rs.HI), dim3(rb), rb * sizeof(LX), s, buckets.p, rs.LO, sums.p);
    allow_big_lds(msm_reduce_cols_kernel<C>, 256 * sizeof(LX));
    allow_big_lds(msm_reduce_cols_final_kernel<C>, 256 * sizeof(LX));
    hipLaunchKernelGGL((msm_reduce_cols_kernel<C>), dim3(rs.LO / 64, rs.gy), dim3(256), 256 * sizeof(LX), s, buckets.p, rs.LO, rs.HI, partial.p);
    hipLaunchKernelGGL((msm_reduce_cols_final_kernel<C>), dim3(rs.LO / 16), dim3(256), 256 * sizeof(LX), s, partial.p, rs.LO, rs.gy, sums.p);
    if (rs.HI < rs.LO) hipLaunchKernelGGL((msm_reduce_pad_kernel<C>), dim3((rs.LO - rs.HI + 255) / 256), dim3(256), 0, s, sums.p, rs.LO + rs.HI, rs.LO - rs.HI);
    const uint32_t sblock = sizeof(X) > 128 ? 128 : 256;
    int sk = 0;
    while ((sblock << sk) < rs.LO) sk++;
    const size_t lds_s = 2 * (size_t)sblock * sizeof(LX);
    allow_big_lds(msm_bucket_reduce_scan_kernel<C>, lds_s);
    hipLaunchKernelGGL((msm_bucket_reduce_scan_kernel<C>), dim3(1, 2), dim3(sblock), lds_s, s, sums.p, rs.LO, sk, d_partials, (X*)nullptr, (uint32_t*)nullptr);
    ICICLE_TRY(check_launch("msm_bucket_reduce (two-level)"));
    return ICICLE_SUCCESS;
  }
  if (rs.scan) {
    // small table-mode set: [TT | L | LL] (LL only with more than one workgroup per slice; the host applies M)
    allow_big_lds(msm_bucket_reduce_scan_kernel<C>, lds_r);
    hipLaunchKernelGGL((msm_bucket_reduce_scan_kernel<C>), dim3(rs.bpw, g.Wb), dim3(rs.rblock), lds_r, s, buckets.p, g.NBb, rs.k_log, d_partials, raw.p, tickets);
  } else {
    allow_big_lds(msm_bucket_reduce_kernel<C>, lds_r);
    hipLaunchKernelGGL((msm_bucket_reduce_kernel<C>), dim3(rs.bpw, g.Wb), dim3(rs.rblock), lds_r, s, buckets.p, g.NBb, rs.k_log, d_partials, g.tab, raw.p, tickets);
  }
  ICICLE_TRY(check_launch("msm_bucket_reduce"));
  return ICICLE_SUCCESS;
}

// host tail, table mode: partials = [S | L | LL][slice] for the Wb slices (NBb buckets each) of the single bucket set:
// S_v the slice's own weighted sum (short by M·LL_v when the scan kernels ran: M = buckets per reduction workgroup, a
// power of two, else 0 and no LL), L_v = T_v its plain sum;  Σ_b (b+1)·B_b = Σ_v S_v + NBb · Σ_v v·T_v
template <class C>
typename C::P msm_host_tail_tab(const typename C::X* part, uint32_t Wb, uint32_t M, uint32_t NBb)
{
  typedef typename C::X X;
  if (M & MSM_TWO_LEVEL_MARK) {
    // two-level reduction: part = [Σ(l+1)·C_l, Σ(h+1)·R_h, Σ C_l, Σ R_h];  Σ_b (b+1)·B_b = part[0] + LO·(part[1] − part[3])
    X U = C::x_add(part[1], C::x_neg(part[3]));
    for (uint32_t m = M & ~MSM_TWO_LEVEL_MARK; m > 1; m >>= 1) U = C::x_dbl(U);
    return C::p_from_mont(C::x_to_projective(C::x_add(part[0], U)));
  }
  X S = C::x_zero(), run = C::x_zero(), U = C::x_zero(), LL = C::x_zero();
  for (int v = (int)Wb - 1; v >= 0; v--) {
    S = C::x_add(S, part[v]);
    if (M) LL = C::x_add(LL, part[2 * (size_t)Wb + v]);
    if (v >= 1) {
      run = C::x_add(run, part[(size_t)Wb + v]); // Σ_{u ≥ v} T_u
      U = C::x_add(U, run);                       // after the loop: Σ_v v·T_v
    }
  }
  for (uint32_t m = NBb; m > 1; m >>= 1) U = C::x_dbl(U);
  S = C::x_add(S, U);
  if (M) {
    for (uint32_t m = M; m > 1; m >>= 1) LL = C::x_dbl(LL); // one chain for all slices: Σ_v M·LL_v = M·Σ_v LL_v
    S = C::x_add(S, LL);
  }
  return C::p_from_mont(C::x_to_projective(S));
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
  typename C::P p = msm_host_tail<C>((const typename C::X*)t->partials, (uint32_t)t->W, 1, t->c, t->wide);
  memcpy(t->result, &p, sizeof p);
  if (t->host_dst) memcpy(t->host_dst, &p, sizeof p);
}

// the extern "C" entry (bn254_msm / bn254_g2_msm): sort + bucket stages + tail.
// batch_size > 1 (msm.h:21-53): `batch_size` scalar vectors of msm_size elements, one result each; the bases are
// shared (are_points_shared_in_batch) or one set per batch element.  The batch elements run back to back on the
// caller's stream.  precompute_factor f > 1: `bases` came from msm_precompute_bases and holds f points per original
// base, [f·i] being the base itself; this backend reads only those (stride f) — the extra multiples trade memory for a
// cheaper bucket reduction in the reference's backends, which is not where the time goes here.
template <class C, class AT, class PT>
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
  MsmProfile* prof = nullptr;
  for (uint32_t bi = 0; bi < batch; bi++) {
    prof = msm_profile_next();
    SortPlan pl;
    (void)hipEventRecord(prof->ev[0], s);
    ICICLE_TRY(msm_sort_run(ss.ptr<fe>() + (size_t)bi * L, L, cfg->c, lbf, cfg->are_scalars_montgomery_form, s, &pl, 0, cfg->bitsize, (int)stride));
    (void)hipEventRecord(prof->ev[4], s);
    prof->has_sort_end = true;
    prof->L = L;
    prof->nbuckets = pl.nbuckets;
    prof->c = pl.g.c;
    prof->W = pl.g.W;
    prof->is_g2 = sizeof(A) > 64;
    const ReduceShape rs = reduce_shape<X>(pl.g);
    WsScoped<X> partials;
    const int Wt = pl.g.Wb; // windows left for the tail (= W, or ⌈W / f⌉ with precomputed bases)
    HIP_TRY(partials.alloc((size_t)Wt, s), ICICLE_ALLOCATION_FAILED);
    (void)rs;
    const A* pts = sb.ptr<A>() + (shared ? 0 : (size_t)bi * L * stride);
    ICICLE_TRY(msm_buckets_run<C>(&pl, pts, cfg->are_points_montgomery_form, 0, 1, s, partials.p, prof)); // the sort entries index the (precomputed) base array directly
    TailSlot* slot = Wt <= 64 ? tail_slot_acquire() : nullptr;
    if (slot) {
      slot->W = Wt;
      slot->c = pl.g.c;
      slot->wide = pl.g.wide < Wt ? pl.g.wide : Wt;
      slot->host_dst = cfg->are_results_on_device ? nullptr : (void*)(results + bi);
      hipError_t he = hipMemcpyAsync(slot->partials, partials.p, (size_t)Wt * sizeof(X), hipMemcpyDeviceToHost, s);
      if (he == hipSuccess) he = hipLaunchHostFunc(s, host_tail_callback<C>, slot);
      if (he == hipSuccess && cfg->are_results_on_device) he = hipMemcpyAsync(results + bi, slot->result, sizeof(P), hipMemcpyHostToDevice, s);
      tail_slot_commit(slot, s); // on every path: the slot is free again once what was enqueued has run
      HIP_TRY(he, ICICLE_COPY_FAILED);
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
