// msm.hip — Pippenger bucket-method MSM over BN254 G1 / G2 for gfx950.
//
// Replaces icicle/backend/cuda/src/msm/cuda_msm.cuh (bucket_method_msm :960-1127 and its kernels,
// SURVEY.md §2.1) behind bn254_msm / bn254_g2_msm.  Results equal the reference's as group elements
// (Σ sᵢ·Pᵢ); the projective representative returned is any valid one, as in the reference
// (different backends already differ there — the prover normalises with to_affine).
//
// Pipeline (everything is enqueued on the caller's stream, no host synchronisation):
//  1. recode + histogram   one thread per scalar: s ↦ signed c-bit digits (d ∈ [−2^(c−1), 2^(c−1)) via
//                          the "+H" trick, scalars with bit 253 set are negated first, so the top
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
#include <string.h>
#include <vector>

#include "common.h"
#include "ec.h"

using namespace bn254;
using namespace isnark;

namespace isnark {
// ring of the most recent MSM launches of this process: HIP events (recorded on the MSM's own stream,
// never synchronised here) + geometry, read back by icicle_snark_msm_profile() after the caller synced.
struct MsmProfile {
  hipEvent_t ev[4]; // start, before accumulate, after accumulate, end
  uint32_t L, nbuckets;
  int c, W, is_g2;
  bool valid;
};
constexpr int MSM_PROFILE_RING = 32;
extern MsmProfile g_msm_ring[MSM_PROFILE_RING];
extern uint64_t g_msm_seq;
MsmProfile* msm_profile_next(uint64_t* seq);
extern thread_local float g_last_msm_ms[4];
bool ext_get_int(const ConfigExtension* ext, const char* key, int* out);
bool ext_get_bool(const ConfigExtension* ext, const char* key, bool* out);
}

namespace {

struct Geom {
  int c, W;
  uint32_t NB;     // buckets per window = 2^(c-1)
  uint32_t H[9];   // Σ_w 2^(c·w + c − 1)
};

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
__device__ __forceinline__ void recode(const fe* scalars, uint32_t i, const Geom& g, int mont, uint32_t t[9], uint32_t& neg)
{
  fe s = ld_fe(scalars + i);
  if (mont) s = Fr::from_mont(s);
  neg = (s.l[7] >> 29) & 1; // bit 253
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
// signed digit of window w: returns 0 for a zero digit, else (mag) with sign in bit 31
__device__ __forceinline__ uint32_t digit(const uint32_t t[9], int w, const Geom& g)
{
  const int bit = w * g.c;
  const int limb = bit >> 5, off = bit & 31;
  uint64_t v = t[limb];
  if (limb < 8) v |= (uint64_t)t[limb + 1] << 32;
  const uint32_t raw = (uint32_t)(v >> off) & ((1u << g.c) - 1);
  const int32_t d = (int32_t)raw - (int32_t)g.NB;
  if (d == 0) return 0;
  return d < 0 ? ((uint32_t)(-d) | 0x80000000u) : (uint32_t)d;
}

// zero `n` u32 words (kernel instead of hipMemsetAsync: keeps every dependency on the compute queue)
__global__ __launch_bounds__(256) void msm_zero_kernel(uint32_t* __restrict__ p, uint32_t n)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0;
}

__global__ __launch_bounds__(256) void msm_hist_kernel(const fe* __restrict__ scalars, uint32_t L, Geom g, int mont, uint32_t* __restrict__ counts)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  uint32_t t[9], neg;
  recode(scalars, i, g, mont, t, neg);
  for (int w = 0; w < g.W; w++) {
    const uint32_t d = digit(t, w, g);
    if (d) atomicAdd(&counts[(uint32_t)w * g.NB + ((d & 0x7fffffffu) - 1)], 1u);
  }
}

__global__ __launch_bounds__(256) void msm_scatter_kernel(const fe* __restrict__ scalars, uint32_t L, Geom g, int mont, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  uint32_t t[9], neg;
  recode(scalars, i, g, mont, t, neg);
  for (int w = 0; w < g.W; w++) {
    const uint32_t d = digit(t, w, g);
    if (d) {
      const uint32_t pos = atomicAdd(&cursor[(uint32_t)w * g.NB + ((d & 0x7fffffffu) - 1)], 1u);
      const uint32_t sign = (d >> 31) ^ neg;
      sorted[pos] = i | (sign << 31);
    }
  }
}

// exclusive scan of m counters by one workgroup of 1024 threads; also finds buckets above `thr`
__global__ __launch_bounds__(1024) void msm_scan_kernel(const uint32_t* __restrict__ counts, uint32_t m, uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                                                          uint32_t thr, uint32_t* __restrict__ n_large, uint32_t* __restrict__ large_list, uint32_t large_cap)
{
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (m + 1023) / 1024;
  const uint32_t lo = tid * chunk, hi = min(lo + chunk, m);
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; k++) s += counts[k];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - s;
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t cnt = counts[k];
    offsets[k] = run;
    cursor[k] = run;
    if (cnt > thr) {
      const uint32_t p = atomicAdd(n_large, 1u);
      if (p < large_cap) large_list[p] = k;
    }
    run += cnt;
  }
}

template <class C>
__device__ __forceinline__ typename C::A load_base(const typename C::A* bases, uint32_t e, int pts_mont, bool& is_zero)
{
  typedef typename C::A A;
  A p = bases[e & 0x7fffffffu];
  is_zero = C::aff_is_zero(p);
  if (!pts_mont) p = C::aff_to_mont(p);
  if (e >> 31) p = C::aff_neg(p);
  return p;
}

template <class C>
__global__ __launch_bounds__(256) void msm_accumulate_kernel(const typename C::A* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ counts, uint32_t nbuckets, uint32_t large_thr, int pts_mont, typename C::X* __restrict__ buckets)
{
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  const uint32_t cnt = counts[b];
  if (cnt > large_thr) return; // step 4b
  const uint32_t off = offsets[b];
  typename C::X acc = C::x_zero();
  for (uint32_t k = 0; k < cnt; k++) {
    bool z;
    typename C::A p = load_base<C>(bases, sorted[off + k], pts_mont, z);
    if (!z) C::x_madd(acc, p);
  }
  buckets[b] = acc;
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

template <class C>
__global__ __launch_bounds__(256) void msm_accumulate_large_kernel(const typename C::A* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
                                                                    const uint32_t* __restrict__ counts, const uint32_t* __restrict__ n_large, const uint32_t* __restrict__ large_list,
                                                                    uint32_t large_cap, int pts_mont, typename C::X* __restrict__ buckets)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typename C::X* sh = reinterpret_cast<typename C::X*>(smem);
  const uint32_t nl = min(*n_large, large_cap);
  for (uint32_t li = blockIdx.x; li < nl; li += gridDim.x) {
    const uint32_t b = large_list[li];
    const uint32_t cnt = counts[b], off = offsets[b];
    typename C::X acc = C::x_zero();
    for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) {
      bool z;
      typename C::A p = load_base<C>(bases, sorted[off + k], pts_mont, z);
      if (!z) C::x_madd(acc, p);
    }
    acc = block_reduce<C>(acc, sh, blockDim.x);
    if (threadIdx.x == 0) buckets[b] = acc;
    __syncthreads();
  }
}

// Σ_b (b+1)·B_b per window.  grid = (blocks per window, W); each thread owns K = 2^k_log buckets.
template <class C>
__global__ __launch_bounds__(256) void msm_bucket_reduce_kernel(const typename C::X* __restrict__ buckets, uint32_t NB, int k_log, typename C::X* __restrict__ partials)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typename C::X* sh = reinterpret_cast<typename C::X*>(smem);
  typedef typename C::X X;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // thread within the window
  const uint32_t base = t << k_log;
  const X* B = buckets + (size_t)blockIdx.y * NB + base;
  X line = C::x_zero(), tri = C::x_zero();
  for (int j = (1 << k_log) - 1; j >= 0; j--) {
    line = C::x_add(line, B[j]);
    tri = C::x_add(tri, line);
  }
  // + base·line  (double-and-add, MSB first)
  if (base != 0 && !C::x_is_zero(line)) {
    X m = C::x_zero();
    for (int bit = 31 - __clz(base); bit >= 0; bit--) {
      m = C::x_dbl(m);
      if ((base >> bit) & 1) m = C::x_add(m, line);
    }
    tri = C::x_add(tri, m);
  }
  tri = block_reduce<C>(tri, sh, blockDim.x);
  if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tri;
}

// window sums → Horner → projective standard form.  One workgroup of 64 threads.
template <class C>
__global__ __launch_bounds__(64) void msm_tail_kernel(const typename C::X* __restrict__ partials, int W, int bpw, int c, typename C::P* __restrict__ result)
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
      for (int j = 0; j < c; j++) acc = C::x_dbl(acc);
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

int ilog2_ceil(uint64_t x)
{
  int l = 0;
  while ((1ull << l) < x) l++;
  return l;
}



template <class C, class AT, class PT>
eIcicleError msm_impl(const bn254_scalar_t* scalars, const AT* bases, int msm_size, const MSMConfig* cfg, PT* results)
{
  typedef typename C::A A;
  typedef typename C::X X;
  typedef typename C::P P;
  static_assert(sizeof(A) == sizeof(AT) && sizeof(P) == sizeof(PT), "ABI layout");
  if (!cfg || !results || (msm_size > 0 && (!scalars || !bases))) return ICICLE_INVALID_POINTER;
  if (msm_size < 0) return ICICLE_INVALID_ARGUMENT;
  if (cfg->batch_size > 1 || cfg->precompute_factor > 1) {
    set_last_error("msm: batch_size > 1 / precompute_factor > 1 are not implemented");
    return ICICLE_API_NOT_IMPLEMENTED;
  }
  ICICLE_TRY(require_device());
  hipStream_t s = (hipStream_t)cfg->stream;
  const uint32_t L = (uint32_t)msm_size;
  const bool profile = getenv("ICICLE_SNARK_PROFILE") != nullptr;

  Staged ss, sb, sr;
  ICICLE_TRY(ss.in(scalars, (size_t)L * sizeof(fe), cfg->are_scalars_on_device, s));
  ICICLE_TRY(sb.in(bases, (size_t)L * sizeof(A), cfg->are_points_on_device, s));
  ICICLE_TRY(sr.out(results, sizeof(P), cfg->are_results_on_device, s));

  // window size: as the reference, ≈ log2(L) − 4 (cuda_msm.cuh:45-48), capped so that bucket
  // magnitudes fit 15 bits + sign.
  Geom g;
  int c = cfg->c > 0 ? cfg->c : ilog2_ceil(L ? L : 1) - 4;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  g.c = c;
  g.W = 254 / c + 1;
  g.NB = 1u << (c - 1);
  {
    uint32_t H[10] = {0};
    for (int w = 0; w < g.W; w++) {
      const int bit = w * c + c - 1;
      H[bit >> 5] |= 1u << (bit & 31);
    }
    memcpy(g.H, H, sizeof g.H);
  }
  const uint32_t nbuckets = g.NB * (uint32_t)g.W;
  const uint64_t nentries = (uint64_t)L * g.W;

  // large-bucket threshold (the reference: large_bucket_factor(10) × average, cuda_msm.cuh:205-220)
  int lbf = 10;
  ext_get_int(cfg->ext, "large_bucket_factor", &lbf);
  uint64_t avg = L / g.NB + 1;
  uint32_t large_thr = (uint32_t)(avg * (uint64_t)lbf);
  if (large_thr < 512) large_thr = 512;
  const uint32_t large_cap = nbuckets; // the list can hold every bucket: no overflow case

  // workspace (stream-ordered pool)
  uint32_t *counts = nullptr, *offsets = nullptr, *cursor = nullptr, *sorted = nullptr, *n_large = nullptr, *large_list = nullptr;
  X *buckets = nullptr, *partials = nullptr;
  const int k_log = (c - 1) > 11 ? (c - 1) - 11 : 0;
  const uint32_t tpw = g.NB >> k_log; // reduce threads per window
  const uint32_t rb_max = sizeof(X) > 128 ? 128 : 256; // LDS tree buffer ≤ 32 KiB
  const uint32_t rblock = tpw < rb_max ? tpw : rb_max;
  const uint32_t bpw = tpw / rblock;
  HIP_TRY(ws_alloc((void**)&counts, (size_t)nbuckets * 4 * 4 + 16, s), ICICLE_ALLOCATION_FAILED);
  offsets = counts + nbuckets;
  cursor = offsets + nbuckets;
  n_large = cursor + nbuckets;
  large_list = n_large + 4;
  HIP_TRY(ws_alloc((void**)&sorted, (size_t)(nentries ? nentries : 1) * 4, s), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(ws_alloc((void**)&buckets, (size_t)nbuckets * sizeof(X), s), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(ws_alloc((void**)&partials, (size_t)g.W * bpw * sizeof(X), s), ICICLE_ALLOCATION_FAILED);
  {
    // counts[nbuckets] and (three arrays further) n_large[4]
    unsigned zb = (nbuckets + 255) / 256;
    if (zb > 1024) zb = 1024;
    hipLaunchKernelGGL(msm_zero_kernel, dim3(zb), dim3(256), 0, s, counts, nbuckets);
    hipLaunchKernelGGL(msm_zero_kernel, dim3(1), dim3(64), 0, s, n_large, 4u);
    ICICLE_TRY(check_launch("msm_zero"));
  }
  // buckets never touched by step 4 (large ones are written by 4b; all others by 4): no memset needed

  uint64_t pseq = 0;
  MsmProfile* prof = msm_profile_next(&pseq);
  hipEvent_t* ev = prof->ev;
  prof->L = L;
  prof->nbuckets = nbuckets;
  prof->c = c;
  prof->W = g.W;
  prof->is_g2 = sizeof(A) > 64;
  (void)hipEventRecord(ev[0], s);

  const int mont_sc = cfg->are_scalars_montgomery_form, mont_pt = cfg->are_points_montgomery_form;
  const unsigned lgrid = (L + 255) / 256;
  if (L) {
    hipLaunchKernelGGL(msm_hist_kernel, dim3(lgrid), dim3(256), 0, s, ss.ptr<fe>(), L, g, mont_sc, counts);
    ICICLE_TRY(check_launch("msm_hist"));
  }
  hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(1024), 0, s, counts, nbuckets, offsets, cursor, large_thr, n_large, large_list, large_cap);
  ICICLE_TRY(check_launch("msm_scan"));
  if (L) {
    hipLaunchKernelGGL(msm_scatter_kernel, dim3(lgrid), dim3(256), 0, s, ss.ptr<fe>(), L, g, mont_sc, cursor, sorted);
    ICICLE_TRY(check_launch("msm_scatter"));
  }
  (void)hipEventRecord(ev[1], s);

  hipLaunchKernelGGL((msm_accumulate_kernel<C>), dim3((nbuckets + 255) / 256), dim3(256), 0, s, sb.ptr<A>(), sorted, offsets, counts, nbuckets, large_thr, mont_pt, buckets);
  ICICLE_TRY(check_launch("msm_accumulate"));
  (void)hipEventRecord(ev[2], s);
  hipLaunchKernelGGL((msm_accumulate_large_kernel<C>), dim3(512), dim3(rb_max), rb_max * sizeof(X), s, sb.ptr<A>(), sorted, offsets, counts, n_large, large_list, large_cap, mont_pt, buckets);
  ICICLE_TRY(check_launch("msm_accumulate_large"));

  hipLaunchKernelGGL((msm_bucket_reduce_kernel<C>), dim3(bpw, g.W), dim3(rblock), rblock * sizeof(X), s, buckets, g.NB, k_log, partials);
  ICICLE_TRY(check_launch("msm_bucket_reduce"));
  hipLaunchKernelGGL((msm_tail_kernel<C>), dim3(1), dim3(64), 0, s, partials, g.W, (int)bpw, c, sr.ptr<P>());
  ICICLE_TRY(check_launch("msm_tail"));
  (void)hipEventRecord(ev[3], s);
  prof->valid = true;

  HIP_TRY(ws_free(counts, s), ICICLE_DEALLOCATION_FAILED);
  HIP_TRY(ws_free(sorted, s), ICICLE_DEALLOCATION_FAILED);
  HIP_TRY(ws_free(buckets, s), ICICLE_DEALLOCATION_FAILED);
  HIP_TRY(ws_free(partials, s), ICICLE_DEALLOCATION_FAILED);
  ICICLE_TRY(sr.finish());
  if (profile) {
    HIP_TRY(hipStreamSynchronize(s), ICICLE_SYNCHRONIZATION_FAILED);
    (void)hipEventElapsedTime(&g_last_msm_ms[0], ev[0], ev[1]);
    (void)hipEventElapsedTime(&g_last_msm_ms[1], ev[1], ev[2]);
    (void)hipEventElapsedTime(&g_last_msm_ms[2], ev[2], ev[3]);
    (void)hipEventElapsedTime(&g_last_msm_ms[3], ev[0], ev[3]);
  }
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
  X* table = nullptr;
  P* proj = nullptr;
  typename F::T* scratch = nullptr;
  HIP_TRY(ws_alloc((void**)&table, 32 * 255 * sizeof(X), s), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(ws_alloc((void**)&proj, n * sizeof(P), s), ICICLE_ALLOCATION_FAILED);
  HIP_TRY(ws_alloc((void**)&scratch, n * sizeof(typename F::T), s), ICICLE_ALLOCATION_FAILED);
  A gm = C::aff_to_mont(gen_std);
  hipLaunchKernelGGL((fixed_base_table_kernel<C>), dim3(1), dim3(256), 0, s, gm, table);
  ICICLE_TRY(check_launch("fixed_base_table"));
  hipLaunchKernelGGL((fixed_base_mul_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const fe*>(sc), n, table, proj);
  ICICLE_TRY(check_launch("fixed_base_mul"));
  const int chunk = 32;
  const uint64_t nthreads = (n + chunk - 1) / chunk;
  hipLaunchKernelGGL((batch_to_affine_kernel<C, F>), dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, s, proj, n, chunk, reinterpret_cast<A*>(out), scratch);
  ICICLE_TRY(check_launch("batch_to_affine"));
  HIP_TRY(ws_free(table, s), ICICLE_DEALLOCATION_FAILED);
  HIP_TRY(ws_free(proj, s), ICICLE_DEALLOCATION_FAILED);
  HIP_TRY(ws_free(scratch, s), ICICLE_DEALLOCATION_FAILED);
  return ICICLE_SUCCESS;
}

} // namespace

