// prover.cpp — the Groth16 prover host (C++ restatement of the reference's Rust host, which cannot be
// built here: no Rust toolchain).  It drives the same C ABI the Rust host would (bn254_msm,
// bn254_g2_msm, bn254_ntt, host curve FFI) plus this library's fused QAP kernels.
//
//   groth16_prove            ← src/lib.rs:33-61
//   CacheManager / ZKeyCache ← src/cache.rs:58-72,110-262
//   construct_r1cs           ← src/proof_helper.rs:31-170      (on the device here: prover/qap.hip)
//   groth16_commitments      ← src/proof_helper.rs:172-241
//   groth16_prove_helper     ← src/proof_helper.rs:243-317
//   snarkjs containers       ← src/file_wrapper.rs:45-208, src/zkey.rs:47-85
//   proof / public JSON      ← src/conversions.rs:30-56, src/file_wrapper.rs:105-113
//
// Deliberate differences from the reference host (SURVEY.md §3.1 "reference inefficiencies"):
//  * A/B evaluation (sparse mat-vec) runs on the GPU from a CSR built once per zkey; there is no host
//    gather, no serial host scatter-add and no D2H/H2D round trip of n_coef elements per proof;
//  * zkey points stay in the Montgomery form the file already has (are_points_montgomery_form = true),
//    so cache construction needs no conversion pass over ~1 GB of points;
//  * the NTT domain is sized 2·domain_size so that the coset keys g^i (g = ω_2n, src/cache.rs:183-184,
//    264-289) are read from the twiddle table instead of a separate array + CWD file cache.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <fcntl.h>
#include <map>
#include <memory>
#include <mutex>
#include <stdarg.h>
#include <string.h>
#include <string>
#include <thread>
#include <errno.h>
#include <sys/mman.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "../../../include/groth16_prover.h"
#include "../common.h"
#include "../ec.h"
#include "../msm_plan.h"
#include "../ntt_fuse.h"
#include "qap.h"

using namespace bn254;
using namespace isnark;

namespace isnark {
const fe* ntt_domain_table(int* log_n); // ntt.hip
}

namespace {

thread_local char g_perr[512] = "";
int fail(int code, const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_perr, sizeof g_perr, fmt, ap);
  va_end(ap);
  return code;
}
#define P_HIP(call)                                                                                                    \
  do {                                                                                                                 \
    hipError_t e__ = (call);                                                                                           \
    if (e__ != hipSuccess) return fail((int)ICICLE_UNKNOWN_ERROR, "%s: %s", #call, hipGetErrorString(e__));            \
  } while (0)
#define P_ICICLE(call)                                                                                                 \
  do {                                                                                                                 \
    eIcicleError e__ = (call);                                                                                         \
    if (e__ != ICICLE_SUCCESS) return fail((int)e__, "%s failed (%d): %s", #call, (int)e__, icicle_snark_last_error()); \
  } while (0)

enum { ERR_IO = -1, ERR_FORMAT = -2, ERR_ARG = -3, ERR_NOCACHE = -4 };

// ------------------------------------------------------------------------------------------------ containers
struct Section {
  const uint8_t* p = nullptr;
  uint64_t size = 0;
  int count = 0;
};
// FileWrapper::read_bin_file — src/file_wrapper.rs:45-103
int read_sections(const uint8_t* data, size_t len, const char* type, uint32_t max_version, std::vector<Section>& out)
{
  if (len < 12 || memcmp(data, type, 4) != 0) return fail(ERR_FORMAT, "Invalid File format (expected '%s')", type);
  uint32_t version, nsec;
  memcpy(&version, data + 4, 4);
  memcpy(&nsec, data + 8, 4);
  if (version > max_version) return fail(ERR_FORMAT, "Version not supported");
  out.assign(nsec + 1 > 16 ? nsec + 1 : 16, Section());
  size_t pos = 12;
  for (uint32_t i = 0; i < nsec; i++) {
    if (len - pos < 12) return fail(ERR_FORMAT, "truncated section table");
    uint32_t ht;
    uint64_t hl;
    memcpy(&ht, data + pos, 4);
    memcpy(&hl, data + pos + 4, 8);
    pos += 12;
    if (hl > len - pos) return fail(ERR_FORMAT, "section %u exceeds the file", ht); // pos <= len here; `pos + hl` could wrap for a hostile 64-bit length
    if (ht < out.size()) {
      out[ht].p = data + pos;
      out[ht].size = hl;
      out[ht].count++;
    }
    pos += hl;
  }
  return 0;
}
int unique_section(const std::vector<Section>& s, size_t id, const Section** sec)
{
  if (id >= s.size() || s[id].count == 0) return fail(ERR_FORMAT, "Missing section %zu", id);
  if (s[id].count > 1) return fail(ERR_FORMAT, "Section Duplicated %zu", id);
  *sec = &s[id];
  return 0;
}

struct MappedFile {
  const uint8_t* data = nullptr;
  size_t len = 0;
  int fd = -1;
  ~MappedFile()
  {
    if (data) munmap((void*)data, len);
    if (fd >= 0) close(fd);
  }
  int open_ro(const char* path)
  {
    fd = ::open(path, O_RDONLY); // the reference opens read-write although it only reads (file_wrapper.rs:50-54)
    if (fd < 0) return fail(ERR_IO, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0) return fail(ERR_IO, "cannot stat %s", path);
    len = (size_t)st.st_size;
    void* p = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) return fail(ERR_IO, "cannot mmap %s", path);
    data = (const uint8_t*)p;
    (void)madvise(p, len, MADV_WILLNEED); // start the read-ahead; the upload workers touch the pages in parallel
    return 0;
  }
};

// ------------------------------------------------------------------------------------------------ cache
struct Wtns {
  uint32_t n8 = 0, n_witness = 0;
  fe q;
  const uint8_t* values = nullptr; // n_witness × 32 B standard form
};
// read_wtns_header + section 2 — src/file_wrapper.rs:169-177, src/proof_helper.rs:247-268
int parse_wtns(const uint8_t* data, size_t len, Wtns& w)
{
  std::vector<Section> s;
  if (int rc = read_sections(data, len, "wtns", 2, s)) return rc;
  const Section *h, *v;
  if (int rc = unique_section(s, 1, &h)) return rc;
  if (int rc = unique_section(s, 2, &v)) return rc;
  if (h->size < 8) return fail(ERR_FORMAT, "wtns header too short");
  memcpy(&w.n8, h->p, 4);
  if (w.n8 != 32 || h->size != 4 + 32 + 4) return fail(ERR_FORMAT, "wtns: unsupported field size %u", w.n8);
  memcpy(w.q.l, h->p + 4, 32);
  memcpy(&w.n_witness, h->p + 36, 4);
  if (v->size != (uint64_t)w.n_witness * 32) return fail(ERR_FORMAT, "wtns: section 2 size mismatch");
  w.values = v->p;
  return 0;
}

// circuits with a domain (and witness) of up to this many elements start their witness MSMs right after the witness sort instead
// of behind the QAP front end (ICICLE_SNARK_EARLY overrides it per process)
// (2^18 is 2-3 % better for the benchmark chain at 300-500 k constraints, 2^19 is 5 % better for the witness-light stand-in at 400 k)
constexpr uint32_t EARLY_MAX_DEFAULT = 1u << 19;
constexpr size_t PARTIALS_STRIDE = 64 * 16 * 256; // ≥ W·bpw·sizeof(XYZZ) for any geometry (W ≤ 64, bpw ≤ 16, G2 256 B)

struct Shard {
  uint32_t lo = 0, hi = 0; // [lo, hi) of the full base array
  void* d_points = nullptr; // internal encoding; in table mode W rows of len() points (row w = 2^(c·w)·P, msm_plan.h)
  uint32_t stride = 1, first = 0; // H of a power-of-two shard count: elements first + k·stride, k < len() (lo = 0, hi = len)
  uint32_t len() const { return hi - lo; }
};

struct ZKeyCache {
  // header — src/zkey.rs:6-21
  uint32_t n8q = 0, n8r = 0, n_vars = 0, n_public = 0, domain_size = 0, n_coef = 0;
  fe q, r;
  G1::P vk_alpha_1, vk_beta_1, vk_delta_1; // standard form projective (host)
  G2::P vk_beta_2, vk_gamma_2, vk_delta_2;
  // device
  int device_id = 0, shard_rank = 0, shard_count = 1;
  MsmGeom geom_w, geom_h; // window geometry of the witness MSMs (A, B1, B2, C) and of the H MSM, fixed at cache build
  // Sparse B: a wire that never occurs on the B side of a constraint has the identity as its B1 and B2 base (snarkjs writes
  // all-zero bytes).  One thread accumulates one bucket, so an identity base skipped inside the shared witness sort saves
  // nothing (the other lanes of the wave still add).  Opt-in: with ICICLE_SNARK_SPARSE_B=<d> set and at most the fraction d of
  // this rank's wires having a B base, B1/B2 hold only those nb bases (d_bidx = their wire numbers relative to A.lo), and the two
  // B MSMs run on their own digit sort of the gathered scalars d_wb (geometry geom_b: the digit width of the full set).
  bool sparse_b = false;
  uint32_t nb = 0;
  uint32_t* d_bidx = nullptr;
  fe* d_wb = nullptr;
  MsmGeom geom_b;
  uint32_t* d_rowptr = nullptr; // 2n+1
  uint32_t* d_cols = nullptr;   // n_coef
  fe* d_vals = nullptr;         // n_coef, Montgomery form
  Shard A, B1, B2, C, H;
  fe* d_witness = nullptr; // n_vars
  fe* d_vec = nullptr;     // 3n
  fe* d_fold = nullptr;    // 3·n/G: folded rows of a strided H shard (qap_coset_fold3)
  // distributed front end (groth16_dist_stage1/2; strided H shards only): Y rows of stage 1, what exchange 1 delivers,
  // what stage 2 sends, the scale table n⁻¹·ω_n^{−r·k2} — 3·m elements each, m = n / shard_count; exchange 2 delivers into d_fold
  fe *d_dist_y = nullptr, *d_dist_recv1 = nullptr, *d_dist_send2 = nullptr, *d_tw1 = nullptr;
  bool dist_ready = false; // d_fold holds the Z rows of this rank: the next commitments call skips its own inverse transform + fold
  fe* d_skeys = nullptr;   // n: n⁻¹·g^i — 1/n and the coset keys folded into the inverse transform's last pass (ntt_fuse.h); built on first use
  uint8_t* d_partials = nullptr; // 5 × PARTIALS_STRIDE: per-window partial sums of the five MSMs
  uint8_t* h_partials = nullptr; // pinned mirror
  hipStream_t s_g1 = nullptr, s_g2 = nullptr, s_g3 = nullptr, s_g4 = nullptr, s_g5 = nullptr, s_qap = nullptr;
  hipEvent_t ev_witness = nullptr, ev_sort = nullptr, ev_sort_b = nullptr, ev_sort_h = nullptr, ev_g2done = nullptr, ev_g4done = nullptr, ev_g5done = nullptr, ev[4] = {nullptr, nullptr, nullptr, nullptr},
             ev_done[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  uint64_t device_bytes = 0;
  bool witness_resident = false; // d_witness holds the witness of the last call (wtns == NULL reuses it)
  Groth16Timings last_tm = {0, 0, 0, 0}; // phase timings of the most recent prove (groth16_last_timings)

  ~ZKeyCache()
  {
    (void)hipSetDevice(device_id);
    if (s_g1) (void)hipStreamSynchronize(s_g1);
    if (s_g2) (void)hipStreamSynchronize(s_g2);
    if (s_g3) (void)hipStreamSynchronize(s_g3);
    if (s_g4) (void)hipStreamSynchronize(s_g4);
    if (s_g5) (void)hipStreamSynchronize(s_g5);
    for (void* p : {(void*)d_rowptr, (void*)d_cols, (void*)d_vals, A.d_points, B1.d_points, B2.d_points, C.d_points, H.d_points, (void*)d_witness, (void*)d_vec, (void*)d_fold, (void*)d_skeys, (void*)d_dist_y, (void*)d_dist_recv1, (void*)d_dist_send2, (void*)d_tw1, (void*)d_partials, (void*)d_bidx, (void*)d_wb})
      if (p) (void)hipFree(p);
    if (h_partials) (void)hipHostFree(h_partials);
    if (s_qap) (void)icicle_destroy_stream(s_qap);
    if (s_g1) (void)icicle_destroy_stream(s_g1);
    if (s_g2) (void)icicle_destroy_stream(s_g2);
    if (s_g3) (void)icicle_destroy_stream(s_g3);
    if (s_g4) (void)icicle_destroy_stream(s_g4);
    if (s_g5) (void)icicle_destroy_stream(s_g5);
    if (ev_witness) (void)hipEventDestroy(ev_witness);
    if (ev_sort) (void)hipEventDestroy(ev_sort);
    if (ev_sort_b) (void)hipEventDestroy(ev_sort_b);
    if (ev_sort_h) (void)hipEventDestroy(ev_sort_h);
    if (ev_g2done) (void)hipEventDestroy(ev_g2done);
    if (ev_g4done) (void)hipEventDestroy(ev_g4done);
    if (ev_g5done) (void)hipEventDestroy(ev_g5done);
    for (auto e : ev)
      if (e) (void)hipEventDestroy(e);
    for (auto e : ev_done)
      if (e) (void)hipEventDestroy(e);
  }
};

G1::P g1_from_mont_affine(const uint8_t* p)
{
  G1::A a;
  memcpy(&a, p, 64);
  if (G1::aff_is_zero(a)) return {Fq::zero(), Fq::one_std(), Fq::zero()};
  return {Fq::from_mont(a.x), Fq::from_mont(a.y), Fq::one_std()};
}
G2::P g2_from_mont_affine(const uint8_t* p)
{
  G2::A a;
  memcpy(&a, p, 128);
  fe2 one = {Fq::one_std(), Fq::zero()};
  if (G2::aff_is_zero(a)) return {Fq2Ops::zero(), one, Fq2Ops::zero()};
  return {Fq2Ops::from_mont(a.x), Fq2Ops::from_mont(a.y), one};
}

// ---- cold path: host → device ingest ------------------------------------------------------------
// The zkey arrives as pageable memory (an mmap of the file, or the caller's buffer).  A pageable hipMemcpy is a
// single-threaded staging copy; isnark::staged_copy (runtime.cpp) runs up to eight workers that copy 2 MB chunks into
// their own pair of pinned buffers and enqueue the DMAs on their own streams, so page faults / memcpy of one chunk
// overlap the DMA of the others.  `lanes`: streams to enqueue the DMAs on (the per-prove witness upload passes the
// prover's own streams, idle at that point); nullptr: short-lived streams of the call (cold path).
typedef CopyJob UploadJob;
int staged_upload(int device_id, const std::vector<UploadJob>& jobs, const hipStream_t* lanes_in = nullptr, int n_lanes = 0)
{
  const hipError_t e = staged_copy(device_id, jobs.data(), jobs.size(), true, lanes_in, n_lanes, /*own_temp_streams=*/lanes_in == nullptr);
  if (e != hipSuccess) return fail((int)ICICLE_COPY_FAILED, "host to device upload: %s", hipGetErrorString(e));
  return 0;
}

int alloc_shard(Shard& sh, const Section* sec, size_t elem, uint32_t total, uint32_t lo, uint32_t hi, uint64_t& bytes, std::vector<UploadJob>& jobs)
{
  if (sec->size != (uint64_t)total * elem) return fail(ERR_FORMAT, "zkey: point section size mismatch");
  sh.lo = lo;
  sh.hi = hi;
  const size_t n = (size_t)sh.len() * elem;
  P_HIP(hipMalloc(&sh.d_points, n ? n : 256));
  if (n) jobs.push_back({sh.d_points, sec->p + (size_t)sh.lo * elem, n});
  bytes += n;
  return 0;
}

// elements per rank when the witness is uploaded in shard_count slices
inline uint64_t witness_slice_elems(uint32_t n_vars, int count) { return ((uint64_t)n_vars + count - 1) / count; }

// CacheManager::compute — src/cache.rs:117-241
int build_cache(const uint8_t* data, size_t len, int device_id, int rank, int count, std::unique_ptr<ZKeyCache>& out)
{
  if (count < 1 || rank < 0 || rank >= count) return fail(ERR_ARG, "bad shard %d/%d", rank, count);
  const bool trace = getenv("ICICLE_SNARK_TRACE_COLD") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[cold] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  std::vector<Section> s;
  if (int rc = read_sections(data, len, "zkey", 2, s)) return rc;
  const Section *s1, *s2, *s4, *s5, *s6, *s7, *s8, *s9;
  if (int rc = unique_section(s, 1, &s1)) return rc;
  uint32_t protocol = 0;
  if (s1->size >= 4) memcpy(&protocol, s1->p, 4);
  if (protocol != 1) return fail(ERR_FORMAT, "Protocol not supported"); // GROTH16_PROTOCOL_ID, file_wrapper.rs:12,196-207
  if (int rc = unique_section(s, 2, &s2)) return rc;
  if (int rc = unique_section(s, 4, &s4)) return rc;
  if (int rc = unique_section(s, 5, &s5)) return rc;
  if (int rc = unique_section(s, 6, &s6)) return rc;
  if (int rc = unique_section(s, 7, &s7)) return rc;
  if (int rc = unique_section(s, 8, &s8)) return rc;
  if (int rc = unique_section(s, 9, &s9)) return rc;

  std::unique_ptr<ZKeyCache> z(new ZKeyCache());
  z->device_id = device_id;
  z->shard_rank = rank;
  z->shard_count = count;
  // read_header_groth16 — src/zkey.rs:47-85
  const uint8_t* h = s2->p;
  if (s2->size < 4 + 32 + 4 + 32 + 12 + 3 * 64 + 3 * 128) return fail(ERR_FORMAT, "zkey header too short");
  memcpy(&z->n8q, h, 4);
  if (z->n8q != 32) return fail(ERR_FORMAT, "zkey: unsupported base field size");
  memcpy(z->q.l, h + 4, 32);
  memcpy(&z->n8r, h + 36, 4);
  if (z->n8r != 32) return fail(ERR_FORMAT, "zkey: unsupported scalar field size");
  memcpy(z->r.l, h + 40, 32);
  memcpy(&z->n_vars, h + 72, 4);
  memcpy(&z->n_public, h + 76, 4);
  memcpy(&z->domain_size, h + 80, 4);
  if (!Fq::eq(z->q, Fq::modulus()) || !Fr::eq(z->r, Fr::modulus())) return fail(ERR_FORMAT, "zkey: not a BN254 key");
  const uint32_t n = z->domain_size;
  if (n == 0 || (n & (n - 1))) return fail(ERR_FORMAT, "zkey: domain size %u is not a power of two", n);
  if (z->n_public + 1 > z->n_vars) return fail(ERR_FORMAT, "zkey: n_public exceeds n_vars");
  const uint8_t* pp = h + 84;
  z->vk_alpha_1 = g1_from_mont_affine(pp);
  z->vk_beta_1 = g1_from_mont_affine(pp + 64);
  z->vk_beta_2 = g2_from_mont_affine(pp + 128);
  z->vk_gamma_2 = g2_from_mont_affine(pp + 256);
  z->vk_delta_1 = g1_from_mont_affine(pp + 384);
  z->vk_delta_2 = g2_from_mont_affine(pp + 448);

  // coefficients (section 4): {m:u32 c:u32 s:u32 value[32]} — src/cache.rs:126-166 (only byte 0 of m is read, :159)
  const size_t rec = 12 + 32;
  if (s4->size < 4 || (s4->size - 4) % rec) return fail(ERR_FORMAT, "zkey: coefficient section size");
  if ((s4->size - 4) / rec > 0xffffffffull) return fail(ERR_FORMAT, "zkey: too many coefficients");
  const uint32_t n_coef = (uint32_t)((s4->size - 4) / rec);
  {
    const uint64_t nv64 = z->n_vars, np1 = (uint64_t)z->n_public + 1;
    if (s5->size != nv64 * 64 || s6->size != nv64 * 64 || s7->size != nv64 * 128 || s8->size != (nv64 - np1) * 64 || s9->size != (uint64_t)n * 64)
      return fail(ERR_FORMAT, "zkey: point section size mismatch");
  }
  z->n_coef = n_coef; // from the section length, like src/cache.rs:129 (the declared count in the first 4 bytes is not read)
  // the container and the header are validated before the device is touched (a malformed key is a format error on any host)
  IcicleDevice dev;
  memset(&dev, 0, sizeof dev);
  strcpy(dev.type, "HIP");
  dev.id = device_id;
  P_ICICLE(icicle_set_device(&dev));
  // device CSR built by kernels from the raw records (prover/csr.hip); the records travel with the points below
  uint32_t* d_records = nullptr;
  const size_t rec_bytes = (size_t)n_coef * rec;
  P_HIP(hipMalloc((void**)&d_records, rec_bytes ? rec_bytes : 4));
  struct FreeTmp {
    void* p;
    ~FreeTmp() { (void)hipFree(p); }
  } free_records{d_records};
  P_HIP(hipMalloc((void**)&z->d_rowptr, (2 * (size_t)n + 1) * 4));
  P_HIP(hipMalloc((void**)&z->d_cols, (size_t)(n_coef ? n_coef : 1) * 4));
  P_HIP(hipMalloc((void**)&z->d_vals, (size_t)(n_coef ? n_coef : 1) * 32));
  z->device_bytes += (2 * (size_t)n + 1) * 4 + (size_t)n_coef * 36;
  std::vector<UploadJob> jobs;
  if (rec_bytes) jobs.push_back({d_records, s4->p + 4, rec_bytes});
  lap("header + coefficient buffers");

  // bases (sections 5-9), this process's point range only
  // A, B1, B2 share the witness range [wlo, whi); C (= witness[n_public+1..]) takes the part of that SAME
  // witness range it covers, so that one digit sort of witness[wlo:whi] serves all four MSMs.
  const uint32_t wlo = (uint32_t)((uint64_t)z->n_vars * rank / count), whi = (uint32_t)((uint64_t)z->n_vars * (rank + 1) / count);
  const uint32_t skip = z->n_public + 1;
  const uint32_t clo = (wlo > skip ? wlo : skip) - skip, chi = (whi > skip ? whi : skip) - skip;
  const uint32_t hlo = (uint32_t)((uint64_t)n * rank / count), hhi = (uint32_t)((uint64_t)n * (rank + 1) / count);
  if (int rc = alloc_shard(z->A, s5, 64, z->n_vars, wlo, whi, z->device_bytes, jobs)) return rc;
  if (int rc = alloc_shard(z->B1, s6, 64, z->n_vars, wlo, whi, z->device_bytes, jobs)) return rc;
  if (int rc = alloc_shard(z->B2, s7, 128, z->n_vars, wlo, whi, z->device_bytes, jobs)) return rc;
  if (int rc = alloc_shard(z->C, s8, 64, z->n_vars - skip, clo, chi, z->device_bytes, jobs)) return rc;
  // H: a power-of-two shard count takes the residue class k ≡ rank (mod count) instead of a contiguous range — the rank
  // then needs the coset evaluations only at those k, which the folded forward transform delivers at 1/count of the cost
  // (qap.h: qap_coset_fold3); the whole section is uploaded once and the class is gathered on the device
  const bool h_strided = count > 1 && (count & (count - 1)) == 0 && n / (uint32_t)count >= 1024;
  void* h_full = nullptr;
  struct FreeFull {
    void** p;
    ~FreeFull() { if (*p) (void)hipFree(*p); }
  } free_full{&h_full};
  if (h_strided) {
    if (s9->size != (uint64_t)n * 64) return fail(ERR_FORMAT, "zkey: point section size mismatch");
    const uint32_t m = n / (uint32_t)count;
    P_HIP(hipMalloc(&h_full, (size_t)n * 64));
    P_HIP(hipMalloc(&z->H.d_points, (size_t)m * 64));
    z->H.lo = 0;
    z->H.hi = m;
    z->H.stride = (uint32_t)count;
    z->H.first = (uint32_t)rank;
    jobs.push_back({h_full, s9->p, (size_t)n * 64});
    z->device_bytes += (size_t)m * 64;
  } else if (int rc = alloc_shard(z->H, s9, 64, n, hlo, hhi, z->device_bytes, jobs)) return rc;
  lap("point buffers (hipMalloc)");
  if (int rc = staged_upload(device_id, jobs)) return rc;
  if (h_strided) {
    P_HIP(qap_gather_strided((const fe*)h_full, (fe*)z->H.d_points, 2, z->H.len(), z->H.stride, z->H.first, nullptr));
    P_HIP(hipStreamSynchronize(nullptr));
    P_HIP(hipFree(h_full));
    h_full = nullptr;
  }
  lap("staged upload");
  {
    uint32_t first_bad = 0;
    P_HIP(qap_build_csr(d_records, n_coef, n, z->n_vars, z->d_rowptr, z->d_cols, z->d_vals, &first_bad, nullptr));
    if (first_bad != 0xffffffffu) return fail(ERR_FORMAT, "zkey: coefficient %u out of range", first_bad);
  }
  lap("device CSR build");
  {
    // sparse B (see ZKeyCache): keep only the wires of this rank's range whose B1 or B2 base is not the identity.  Opt-in
    // (ICICLE_SNARK_SPARSE_B=<largest density>): measured on the stand-in circuits it pays at 1.4 M constraints (8.7 → 8.0 ms) and
    // saves table memory, but costs 0.2–0.8 ms between 0.1 M and 1.0 M — the second digit sort heads the G2 chain, the longest of
    // a witness-light prove
    const double max_density = getenv("ICICLE_SNARK_SPARSE_B") ? atof(getenv("ICICLE_SNARK_SPARSE_B")) : 0.0;
    const uint32_t L = z->B1.len();
    if (max_density > 0 && L >= 2) {
      uint8_t* d_flags = nullptr;
      P_HIP(hipMalloc((void**)&d_flags, L));
      FreeTmp free_flags{d_flags};
      P_HIP(qap_points_nonzero(z->B1.d_points, z->B2.d_points, L, d_flags, nullptr));
      std::vector<uint8_t> flags(L);
      P_HIP(hipMemcpy(flags.data(), d_flags, L, hipMemcpyDeviceToHost));
      std::vector<uint32_t> idx;
      idx.reserve(L);
      for (uint32_t i = 0; i < L; i++)
        if (flags[i]) idx.push_back(i);
      const uint32_t nb = (uint32_t)idx.size();
      if (nb >= 1 && (double)nb <= max_density * (double)L) {
        P_HIP(hipMalloc((void**)&z->d_bidx, (size_t)nb * 4));
        P_HIP(hipMemcpy(z->d_bidx, idx.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
        void *c1 = nullptr, *c2 = nullptr;
        P_HIP(hipMalloc(&c1, (size_t)nb * 64));
        FreeTmp free_c1{c1};
        P_HIP(hipMalloc(&c2, (size_t)nb * 128));
        FreeTmp free_c2{c2};
        P_HIP(qap_gather_idx(z->B1.d_points, z->d_bidx, c1, nb, 64, nullptr));
        P_HIP(qap_gather_idx(z->B2.d_points, z->d_bidx, c2, nb, 128, nullptr));
        P_HIP(hipStreamSynchronize(nullptr));
        std::swap(free_c1.p, z->B1.d_points); // the dense arrays are freed at the end of this block
        std::swap(free_c2.p, z->B2.d_points);
        z->B1.lo = z->B2.lo = 0;
        z->B1.hi = z->B2.hi = nb;
        P_HIP(hipMalloc((void**)&z->d_wb, (size_t)nb * 32));
        z->device_bytes -= (uint64_t)(L - nb) * (64 + 128);
        z->device_bytes += (uint64_t)nb * (4 + 32);
        z->sparse_b = true;
        z->nb = nb;
      }
    }
  }
  lap("sparse B detection");
  // bases: the file's Montgomery form (R = 2^256) → the bucket kernels' internal encoding (R' = 2^261), once.  Table mode
  // (msm_plan.h; ICICLE_SNARK_TABLES=0 disables it): every base array becomes W rows 2^(c·w)·P so that all digits of a
  // scalar share one bucket set — 13 instead of 16 mixed additions per scalar at 1.6 M constraints for 13× the base memory.
  {
    bool tables = !(getenv("ICICLE_SNARK_TABLES") && atoi(getenv("ICICLE_SNARK_TABLES")) == 0);
    z->geom_w = msm_geometry(z->A.len(), 0, tables ? 1 : 0);
    z->geom_h = msm_geometry(z->H.len(), 0, tables ? 1 : 0);
    // the B subset keeps the digit width of the full witness set: fewer buckets would mean longer single-thread chains for the
    // 0/1-heavy witnesses this path exists for (404 k wires, 219 k with a B base: c = 17 instead of 19 cost 1.2 ms of a 4 ms prove)
    z->geom_b = z->sparse_b ? msm_geometry(z->nb, 0, tables ? z->geom_w.c : 0) : z->geom_w;
    if (z->sparse_b && tables && z->geom_b.c != z->geom_w.c) z->geom_b = msm_geometry(z->nb, 0, 1);
    if (tables) {
      // the tables need W× the base memory plus the temporaries of the largest build (projective rows + inversion
      // scratch of the G2 set); keep the classic layout when the device cannot hold them next to what is already there
      size_t free_b = 0, total_b = 0;
      release_cached_device_memory(); // blocks parked by icicle_free count as free
      P_HIP(hipMemGetInfo(&free_b, &total_b));
      const uint64_t ww = (uint64_t)z->geom_w.W, wh = (uint64_t)z->geom_h.W, wb = (uint64_t)z->geom_b.W;
      const uint64_t need = ww * ((uint64_t)z->A.len() * 64 + (uint64_t)z->C.len() * 64) + wb * (uint64_t)z->B1.len() * (64 + 128) + wh * (uint64_t)z->H.len() * 64 +
                            wb * (uint64_t)z->B2.len() * (192 + 64) + ((uint64_t)n * 128 + (uint64_t)z->n_vars * 32 + (64u << 20));
      if (need > free_b) {
        tables = false;
        z->geom_w = msm_geometry(z->A.len(), 0, 0);
        z->geom_h = msm_geometry(z->H.len(), 0, 0);
        z->geom_b = z->sparse_b ? msm_geometry(z->nb, z->geom_w.c, 0) : z->geom_w;
      }
    }
    struct Job { Shard* sh; bool g2; const MsmGeom* g; };
    const Job jobs5[5] = {{&z->A, false, &z->geom_w}, {&z->B1, false, &z->geom_b}, {&z->B2, true, &z->geom_b}, {&z->C, false, &z->geom_w}, {&z->H, false, &z->geom_h}};
    for (const Job& j : jobs5) {
      if (j.g->tab) {
        void* table = nullptr;
        P_ICICLE(j.g2 ? msm_g2_build_table(j.sh->d_points, j.sh->len(), 1, *j.g, nullptr, &table) : msm_g1_build_table(j.sh->d_points, j.sh->len(), 1, *j.g, nullptr, &table));
        P_HIP(hipFree(j.sh->d_points));
        j.sh->d_points = table;
        z->device_bytes += (uint64_t)j.sh->len() * (j.g->W - 1) * (j.g2 ? 128 : 64);
      } else {
        P_ICICLE(j.g2 ? msm_g2_points_to_internal(j.sh->d_points, j.sh->len(), 1, nullptr) : msm_g1_points_to_internal(j.sh->d_points, j.sh->len(), 1, nullptr));
      }
    }
  }
  P_HIP(hipStreamSynchronize(nullptr));
  lap("points to internal form / tables");

  // room for shard_count equal slices (groth16_upload_witness_slice: the in-place all-gather wants equal counts)
  P_HIP(hipMalloc((void**)&z->d_witness, (size_t)witness_slice_elems(z->n_vars, count) * count * 32));
  P_HIP(hipMalloc((void**)&z->d_vec, (size_t)n * 3 * 32));
  if (z->H.stride > 1) P_HIP(hipMalloc((void**)&z->d_fold, (size_t)z->H.len() * 3 * 32));
  P_HIP(hipMalloc((void**)&z->d_partials, 5 * PARTIALS_STRIDE));
  P_HIP(hipHostMalloc((void**)&z->h_partials, 5 * PARTIALS_STRIDE));
  z->device_bytes += (size_t)z->n_vars * 32 + (size_t)n * 96;
  // six streams; the library asks the runtime for eight hardware queues so that they do not share one (runtime.cpp).
  // Stream priorities were tried (QAP chain high, G2 low, …): every variant was 1-2 ms slower than equal priorities.
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_qap)); // QAP front end (its own hardware queue; a higher stream priority made no difference)
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g1));
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g2));
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g3));
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g4));
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g5));
  {
    // the first host→device copy on a stream sets up its DMA queue (milliseconds, measured 20 ms over six streams): do
    // it here, not inside the first prove that brings a new witness
    const hipStream_t all[6] = {z->s_qap, z->s_g1, z->s_g2, z->s_g3, z->s_g4, z->s_g5};
    for (int rep = 0; rep < 2; rep++)
      for (hipStream_t st : all) P_HIP(hipMemcpyAsync(z->d_partials, z->h_partials, 4096, hipMemcpyHostToDevice, st));
    for (hipStream_t st : all) P_HIP(hipStreamSynchronize(st));
  }
  P_HIP(hipEventCreateWithFlags(&z->ev_witness, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_sort, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_sort_b, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_sort_h, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_g2done, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_g4done, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_g5done, hipEventDisableTiming));
  for (auto& e : z->ev) P_HIP(hipEventCreate(&e));
  for (auto& e : z->ev_done) P_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  lap("work buffers, streams, events");
  out = std::move(z);
  return 0;
}

} // namespace

struct Groth16CacheManager {
  std::mutex mu;     // serialises cache builds and proves (one device pipeline per manager)
  std::mutex map_mu; // guards `cache`; entries are shared_ptr so that an evict cannot free a key a prove still uses
  std::map<std::string, std::shared_ptr<ZKeyCache>> cache;
  uint32_t domain_n = 0; // domain_size the NTT domain was last initialised for (get_cache, src/cache.rs:242-256)
};

namespace {

// get_cache — src/cache.rs:242-256: (re)initialise the NTT domain for this key.  Sized 2·domain_size here
// (see file header); the reference sizes it from points_a.len() (a quirk).
int ensure_domain(Groth16CacheManager* cm, const ZKeyCache* z)
{
  if (cm->domain_n == z->domain_size) {
    int lg = 0;
    if (ntt_domain_table(&lg) && (1u << lg) >= 2 * z->domain_size) return 0;
  }
  P_ICICLE(bn254_ntt_release_domain());
  bn254_scalar_t root;
  P_ICICLE(bn254_get_root_of_unity(2ull * z->domain_size, &root));
  NTTInitDomainConfig ic;
  memset(&ic, 0, sizeof ic);
  P_ICICLE(bn254_ntt_init_domain(&root, &ic));
  cm->domain_n = z->domain_size;
  return 0;
}

double ms_since(std::chrono::steady_clock::time_point t0)
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

std::shared_ptr<ZKeyCache> find(Groth16CacheManager* cm, const char* key)
{
  std::lock_guard<std::mutex> lk(cm->map_mu);
  auto it = cm->cache.find(key ? key : "");
  return it == cm->cache.end() ? nullptr : it->second;
}

// ------------------------------------------------------------------------------------------------ JSON
std::string to_decimal(const fe& v) // BigUint::to_str_radix(10) — src/conversions.rs:30-40
{
  uint32_t w[8];
  memcpy(w, v.l, 32);
  std::string out;
  bool nz = true;
  while (nz) {
    uint64_t rem = 0;
    nz = false;
    for (int i = 7; i >= 0; i--) {
      uint64_t cur = (rem << 32) | w[i];
      w[i] = (uint32_t)(cur / 1000000000u);
      rem = cur % 1000000000u;
      if (w[i]) nz = true;
    }
    char buf[16];
    snprintf(buf, sizeof buf, nz ? "%09u" : "%u", (unsigned)rem);
    out.insert(0, buf);
  }
  return out;
}

} // namespace

// ------------------------------------------------------------------------------------------------ C API
extern "C" {

__attribute__((visibility("default"))) const char* groth16_last_error(void) { return g_perr; }

__attribute__((visibility("default"))) Groth16CacheManager* groth16_cache_manager_new(void) { return new Groth16CacheManager(); }
__attribute__((visibility("default"))) void groth16_cache_manager_free(Groth16CacheManager* cm) { delete cm; }

__attribute__((visibility("default"))) int groth16_cache_contains(const Groth16CacheManager* cm, const char* key)
{
  return cm && find(const_cast<Groth16CacheManager*>(cm), key) ? 1 : 0;
}
__attribute__((visibility("default"))) void groth16_cache_evict(Groth16CacheManager* cm, const char* key)
{
  if (!cm) return;
  std::shared_ptr<ZKeyCache> victim; // destroyed outside the map lock; a prove in flight keeps its own reference
  {
    std::lock_guard<std::mutex> lk(cm->map_mu);
    auto it = cm->cache.find(key ? key : "");
    if (it == cm->cache.end()) return;
    victim = std::move(it->second);
    cm->cache.erase(it);
  }
}

__attribute__((visibility("default"))) int groth16_cache_load(Groth16CacheManager* cm, const char* key, const void* zkey, size_t zkey_len, int device_id, int shard_rank, int shard_count)
{
  if (!cm || !key || !zkey) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  if (find(cm, key)) return 0;
  std::unique_ptr<ZKeyCache> z;
  if (int rc = build_cache((const uint8_t*)zkey, zkey_len, device_id, shard_rank, shard_count, z)) return rc;
  std::lock_guard<std::mutex> lm(cm->map_mu);
  cm->cache[key] = std::shared_ptr<ZKeyCache>(z.release());
  return 0;
}

__attribute__((visibility("default"))) int groth16_cache_load_file(Groth16CacheManager* cm, const char* key, const char* zkey_path, int device_id, int shard_rank, int shard_count)
{
  if (!cm || !key || !zkey_path) return fail(ERR_ARG, "null argument");
  if (groth16_cache_contains(cm, key)) return 0;
  MappedFile f;
  if (int rc = f.open_ro(zkey_path)) return rc;
  static const bool file_pread = !(getenv("ICICLE_SNARK_FILE_PREAD") && atoi(getenv("ICICLE_SNARK_FILE_PREAD")) == 0);
  if (file_pread) staged_copy_file_hint(f.data, f.len, f.fd); // sections 4-9 are pread() into the pinned staging buffers
  const int rc = groth16_cache_load(cm, key, f.data, f.len, device_id, shard_rank, shard_count);
  staged_copy_file_hint(nullptr, 0, -1);
  return rc;
}

__attribute__((visibility("default"))) int groth16_last_timings(Groth16CacheManager* cm, const char* key, Groth16Timings* tm)
{
  if (!cm || !tm) return fail(ERR_ARG, "null argument");
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  if (!zp) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  std::lock_guard<std::mutex> lk(cm->mu);
  *tm = zp->last_tm;
  return 0;
}

__attribute__((visibility("default"))) int groth16_cache_info(const Groth16CacheManager* cm, const char* key, Groth16CircuitInfo* info)
{
  if (!cm || !info) return fail(ERR_ARG, "null argument");
  const std::shared_ptr<ZKeyCache> zp = find(const_cast<Groth16CacheManager*>(cm), key);
  if (!zp) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  const ZKeyCache* z = zp.get();
  info->n_vars = z->n_vars;
  info->n_public = z->n_public;
  info->domain_size = z->domain_size;
  info->n_coef = z->n_coef;
  info->device_bytes = z->device_bytes;
  info->b_bases = z->B1.len();
  info->reserved = 0;
  return 0;
}

} // extern "C"

namespace {
// blinding terms that do not depend on the commitments: δ1·r, δ1·s, δ2·s, δ1·r·s (src/proof_helper.rs:280-283);
// groth16_prove_mem computes them on a host thread while the GPU works
struct Blinding {
  bn254_scalar_t r, s;
  bn254_projective_t d1r, d1s, d1rs;
  bn254_g2_projective_t d2s;
};
// The two scalar multiplications of the proof's C term that need a commitment — (A + α1 + δ1·r)·s and
// (B1 + β1 + δ1·s)·r, src/proof_helper.rs:284-287 — only need A and B1, which are complete milliseconds before H:
// the host threads that finish those two MSMs go on to compute them while the GPU still works (single-GPU prove only;
// a sharded prove has to sum the commitments of all ranks first).
struct EarlyTerms {
  const Blinding* bl = nullptr;
  std::atomic<bool> bl_ready{false};
  bn254_projective_t ta, tb;
  std::atomic<int> done{0};
};
int commitments_impl(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, uint8_t* out_points, Groth16Timings* tm, EarlyTerms* et);
} // namespace

extern "C" {

__attribute__((visibility("default"))) int groth16_commitments(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, uint8_t out_points[GROTH16_COMMITMENTS_BYTES], Groth16Timings* tm)
{
  return commitments_impl(cm, key, wtns, wtns_len, out_points, tm, nullptr);
}

} // extern "C"

namespace {
int commitments_impl(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, uint8_t* out_points, Groth16Timings* tm, EarlyTerms* et)
{
  if (!cm || !out_points) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  ZKeyCache* z = zp.get();
  if (!z) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  if (!wtns && !z->witness_resident) return fail(ERR_ARG, "no witness given and none resident on the device");
  const auto t0 = std::chrono::steady_clock::now();
  IcicleDevice dev;
  memset(&dev, 0, sizeof dev);
  strcpy(dev.type, "HIP");
  dev.id = z->device_id;
  P_ICICLE(icicle_set_device(&dev));
  if (int rc = ensure_domain(cm, z)) return rc;
  static const bool trace_host = getenv("ICICLE_SNARK_TRACE_HOST") != nullptr;
  auto mark = [&](const char* what) {
    if (trace_host) fprintf(stderr, "[host] %-12s %8.1f us\n", what, ms_since(t0) * 1e3);
  };

  const uint32_t n = z->domain_size, nv = z->n_vars, npub = z->n_public;
  hipStream_t g1 = z->s_g1, g2 = z->s_g2, g3 = z->s_g3;
  double h2d_host_ms = 0;
  bool pinned_src = false;
  if (wtns) {
    Wtns w;
    if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
    // src/proof_helper.rs:253-262
    if (!Fr::eq(z->r, w.q)) return fail(ERR_FORMAT, "Curve of the witness does not match the curve of the proving key");
    if (w.n_witness != z->n_vars) return fail(ERR_FORMAT, "Invalid witness length. Circuit: %u, witness: %u", z->n_vars, w.n_witness);
    // witness → device through the parallel pinned-staging uploader of the cold path (three workers on prover streams, 2 MB
    // chunks): a single memcpy into one pinned buffer + one DMA took 4 ms for the 51 MB of benchmark/1600k
    const auto tu = std::chrono::steady_clock::now();
    // three lanes: 51 MB in 1.4 ms, stable; with six, one upload in four stalled for ~15 ms on the GPU box (host threads
    // of this call, the MSM tails and the runtime's own compete for the container's CPU quota)
    const hipStream_t lanes[6] = {z->s_qap, z->s_g2, z->s_g3, z->s_g1, z->s_g4, z->s_g5};
    static const int n_lanes = getenv("ICICLE_SNARK_UPLOAD_LANES") ? std::max(1, std::min(6, atoi(getenv("ICICLE_SNARK_UPLOAD_LANES")))) : 3;
    if (is_pinned_host(w.values)) {
      // the caller's buffer is pinned (hipHostMalloc / hipHostRegister): one DMA straight from it on g1, in stream order with
      // everything that waits for ev_witness — no staging copy, no host wait (51 MB: 0.9 instead of 1.35 ms)
      P_HIP(hipEventRecord(z->ev[0], g1));
      P_HIP(hipMemcpyAsync(z->d_witness, w.values, (size_t)nv * 32, hipMemcpyHostToDevice, g1));
      pinned_src = true;
    } else if (int rc = staged_upload(z->device_id, {{z->d_witness, (const uint8_t*)w.values, (size_t)nv * 32}}, lanes, n_lanes)) return rc;
    h2d_host_ms = ms_since(tu);
  }
  if (!pinned_src) P_HIP(hipEventRecord(z->ev[0], g1));
  z->witness_resident = true;
  P_HIP(hipEventRecord(z->ev_witness, g1));
  P_HIP(hipEventRecord(z->ev[1], g1));

  // ---- stream g2: ONE digit sort of witness[wlo:whi] (shared by A, B1, B2, C), then the G2 bucket stages
  P_HIP(hipStreamWaitEvent(g2, z->ev_witness, 0));
  const uint32_t wlo = z->A.lo, wlen = z->A.len(), skip = npub + 1;
  SortPlan plan_w, plan_h, plan_b; // plan_b: the B pair's own sort (sparse B only)
  // declared after the plans, so it runs before their destructors: on an error return the kernels already enqueued may
  // still read the plans' workspace, which ~SortPlan hands back to the arena — drain the six streams first
  struct DrainOnError {
    ZKeyCache* z;
    bool armed = true;
    ~DrainOnError()
    {
      if (!armed) return;
      for (hipStream_t st : {z->s_qap, z->s_g1, z->s_g2, z->s_g3, z->s_g4, z->s_g5})
        if (st) (void)hipStreamSynchronize(st);
    }
  } drain{z};
  MsmProfile* prof[5]; // A, B1, B2, C, H
  for (auto& p : prof) p = msm_profile_next();
  // the witness sort is timed with the profile of the G2 MSM that follows it on g2 — of A when B2 runs on the sparse-B sort
  MsmProfile* psort = z->sparse_b ? prof[0] : prof[2];
  (void)hipEventRecord(psort->ev[0], g2);
  P_ICICLE(msm_sort_run(z->d_witness + wlo, wlen, 0, 0, 0, g2, &plan_w, z->geom_w.tab));
  if (plan_w.g.tab != z->geom_w.tab || plan_w.g.c != z->geom_w.c) return fail((int)ICICLE_UNKNOWN_ERROR, "window geometry of the cached tables does not match the witness sort");
  (void)hipEventRecord(psort->ev[4], g2); // end of the witness digit sort (roofline.scatter)
  psort->has_sort_end = true;
  P_HIP(hipEventRecord(z->ev_sort, g2));
  mark("wsort");
  if (z->sparse_b) {
    // sparse B: the scalars of the wires that have a B base, gathered and sorted on B1's stream next to the witness sort
    hipStream_t gb = z->s_g4;
    P_HIP(hipStreamWaitEvent(gb, z->ev_witness, 0));
    P_HIP(qap_gather_idx(z->d_witness + wlo, z->d_bidx, z->d_wb, z->nb, 32, gb));
    P_ICICLE(msm_sort_run(z->d_wb, z->nb, z->geom_b.tab ? 0 : z->geom_b.c, 0, 0, gb, &plan_b, z->geom_b.tab ? z->geom_b.c : 0));
    if (plan_b.g.tab != z->geom_b.tab || plan_b.g.c != z->geom_b.c) return fail((int)ICICLE_UNKNOWN_ERROR, "window geometry of the cached tables does not match the B sort");
    P_HIP(hipEventRecord(z->ev_sort_b, gb));
    P_HIP(hipStreamWaitEvent(g2, z->ev_sort_b, 0));
    mark("bsort");
  }
  const SortPlan& plan_b12 = z->sparse_b ? plan_b : plan_w; // the plan B1 and B2 run on
  auto fill = [](MsmProfile* p, const SortPlan& pl, int g2flag) {
    p->L = pl.L; p->nbuckets = pl.nbuckets; p->c = pl.g.c; p->W = pl.g.W; p->is_g2 = g2flag;
  };
  uint8_t* DP = z->d_partials;

  // ---- stream gq: construct_r1cs (src/proof_helper.rs:31-170) on the device
  hipStream_t gq = z->s_qap;
  P_HIP(hipStreamWaitEvent(gq, z->ev_witness, 0));
  // measurement knob: the QAP chain starts behind the witness sort, so that the sort's event time (roofline.scatter) is its
  // solo time instead of its time next to the spmv and the first transform pass
  static const bool sort_solo = getenv("ICICLE_SNARK_SORT_SOLO") && atoi(getenv("ICICLE_SNARK_SORT_SOLO")) != 0;
  if (sort_solo) P_HIP(hipStreamWaitEvent(gq, z->ev_sort, 0));
  const bool dist_ready = z->dist_ready && z->H.stride > 1; // the distributed stages already left this rank's Z rows in d_fold
  z->dist_ready = false;
  if (!dist_ready) P_HIP(qap_spmv(z->d_witness, z->d_rowptr, z->d_cols, z->d_vals, n, z->d_vec, gq));
  NTTConfig nc;
  memset(&nc, 0, sizeof nc);
  nc.stream = gq;
  nc.coset_gen.limbs[0] = 1;
  nc.batch_size = 3;
  nc.ordering = kNN;
  nc.are_inputs_on_device = nc.are_outputs_on_device = true;
  nc.is_async = true;
  int dom_log = 0;
  const fe* tw = ntt_domain_table(&dom_log);
  const fe* d_hscalars = z->d_vec + n + z->H.lo; // slot 1 of the result, this rank's range
  static const bool fuse_cfg = !(getenv("ICICLE_SNARK_NTT_FUSE") && atoi(getenv("ICICLE_SNARK_NTT_FUSE")) == 0);
  if (z->H.stride > 1) {
    // strided H shard: coset keys, the fold over the shard count and the twist in one pass, then a size-n/G transform
    const uint32_t m = z->H.len();
    if (!dist_ready) {
      P_ICICLE(bn254_ntt((const bn254_scalar_t*)z->d_vec, (int)n, kInverse, &nc, (bn254_scalar_t*)z->d_vec)); // :116
      P_HIP(qap_coset_fold3(z->d_vec, tw, (1u << dom_log) / (2 * n), n, z->H.stride, z->H.first, z->d_fold, gq));
    }
    if (fuse_cfg && ntt_fusable(m)) {
      NttFuse f;
      f.fused_out = z->d_fold + m;
      P_ICICLE(ntt_fused(z->d_fold, m, 3, false, gq, f));
    } else {
      P_ICICLE(bn254_ntt((const bn254_scalar_t*)z->d_fold, (int)m, kForward, &nc, (bn254_scalar_t*)z->d_fold));
      P_HIP(qap_final(z->d_fold, m, gq));
    }
    d_hscalars = z->d_fold + m;
  } else if (fuse_cfg && ntt_fusable(n)) {
    // inverse transform with 1/n and the coset keys folded into its last pass (:116-141), forward transform with the
    // A·B − C epilogue folded into its last pass (:145-167): no coset sweep, no final sweep, n instead of 3n stores
    if (!z->d_skeys) {
      P_HIP(hipMalloc((void**)&z->d_skeys, (size_t)n * 32));
      z->device_bytes += (size_t)n * 32;
      P_ICICLE(ntt_build_scaled_keys(n, z->d_skeys, gq));
    }
    NttFuse fi, ff;
    fi.scale_tab = z->d_skeys;
    P_ICICLE(ntt_fused(z->d_vec, n, 3, true, gq, fi));
    ff.fused_out = z->d_vec + n;
    P_ICICLE(ntt_fused(z->d_vec, n, 3, false, gq, ff));
  } else {
    P_ICICLE(bn254_ntt((const bn254_scalar_t*)z->d_vec, (int)n, kInverse, &nc, (bn254_scalar_t*)z->d_vec)); // :116
    P_HIP(qap_coset_mul3(z->d_vec, tw, (1u << dom_log) / (2 * n), n, gq));                                   // :121-141
    P_ICICLE(bn254_ntt((const bn254_scalar_t*)z->d_vec, (int)n, kForward, &nc, (bn254_scalar_t*)z->d_vec)); // :145
    P_HIP(qap_final(z->d_vec, n, gq));                                                                        // :154-167
  }
  P_HIP(hipEventRecord(z->ev[2], gq));
  mark("qap");
  static const int early_cfg = getenv("ICICLE_SNARK_EARLY") ? atoi(getenv("ICICLE_SNARK_EARLY")) : -1;
  // Witness MSMs of small circuits (domain up to 2^19) leave the GPU far from full: they start right
  // after the witness sort instead of waiting for the QAP (200 k constraints: 3.71 → 3.51 ms, 400 k: 6.69 → 6.24 ms; the QAP
  // itself slows down — 1.1 → 3.9 ms at 400 k — which is why the large ones are held back: 800 k: 9.87 → 10.27 ms).
  // ICICLE_SNARK_EARLY=<max scalars> moves the threshold (0 = never).
  const uint32_t early_max = early_cfg >= 0 ? (uint32_t)early_cfg : EARLY_MAX_DEFAULT;
  const bool early = wlen <= early_max && n <= early_max; // shards of a large circuit keep the full-size inverse transform: neutral there
  if (!early) P_HIP(hipStreamWaitEvent(g1, z->ev[2], 0));

  // ---- stream g2: G2 bucket stages.  Held back until the QAP front end is done: the G2 accumulation
  // fills every CU with ~4 ms workgroups, and the NTT passes of the (longer) g1 chain measured 8× slower
  // when they had to wait for those to retire (rocprof: 2.9 ms vs 0.35 ms per pass).
  if (!early) P_HIP(hipStreamWaitEvent(g2, z->ev[2], 0));
  fill(prof[2], plan_b12, 1);
  if (z->sparse_b) (void)hipEventRecord(prof[2]->ev[0], g2);
  P_ICICLE(msm_g2_partials(&plan_b12, z->B2.d_points, 2, 0, g2, DP + 2 * PARTIALS_STRIDE, prof[2], z->B2.len(), 3)); // commitment_b — src/proof_helper.rs:206
  (void)hipEventRecord(prof[2]->ev[3], g2);
  prof[2]->valid = true;
  P_HIP(hipEventRecord(z->ev_g2done, g2));
  mark("g2");

  // ---- stream g3: digit sort of the H scalars (atomics / memory bound) overlaps the ALU-bound A, B1, C stages
  P_HIP(hipStreamWaitEvent(g3, z->ev[2], 0));
  (void)hipEventRecord(prof[4]->ev[0], g3);
  P_ICICLE(msm_sort_run(d_hscalars, z->H.len(), 0, 0, 0, g3, &plan_h, z->geom_h.tab));
  if (plan_h.g.tab != z->geom_h.tab || plan_h.g.c != z->geom_h.c) return fail((int)ICICLE_UNKNOWN_ERROR, "window geometry of the cached tables does not match the H sort");
  (void)hipEventRecord(prof[4]->ev[4], g3);
  prof[4]->has_sort_end = true;
  P_HIP(hipEventRecord(z->ev_sort_h, g3));
  mark("hsort");

  // ---- groth16_commitments — src/proof_helper.rs:198-205.  A, B1, C share the witness sort and run on three
  // streams (g1, g4, g5) so that the latency-bound bucket reduction of one overlaps the accumulation of the
  // others; H follows A on g1.
  const uint32_t skip_below = skip > wlo ? skip - wlo : 0; // C ignores witness[0..=n_public]
  const int order[3] = {0, 1, 3};
  const Shard* sh3[3] = {&z->A, &z->B1, &z->C};
  hipStream_t st3[3] = {g1, z->s_g4, z->s_g5};
  for (int k = 0; k < 3; k++) {
    MsmProfile* p = prof[order[k]];
    const SortPlan& pl = k == 1 ? plan_b12 : plan_w;
    fill(p, pl, 0);
    P_HIP(hipStreamWaitEvent(st3[k], k == 1 && z->sparse_b ? z->ev_sort_b : z->ev_sort, 0));
    if (k && !early) P_HIP(hipStreamWaitEvent(st3[k], z->ev[2], 0)); // not before the QAP front end is done (see g2)
    if (p != psort) (void)hipEventRecord(p->ev[0], st3[k]);
    P_ICICLE(msm_g1_partials(&pl, sh3[k]->d_points, 2, k == 2 ? skip_below : 0, st3[k], DP + order[k] * PARTIALS_STRIDE, p, sh3[k]->len(), k)); // ticket slots 0-2 of the plan (B2: 3)
    (void)hipEventRecord(p->ev[3], st3[k]);
    p->valid = true;
  }
  P_HIP(hipEventRecord(z->ev_g4done, z->s_g4));
  P_HIP(hipEventRecord(z->ev_g5done, z->s_g5));
  mark("abc");
  // H: behind one of the witness MSMs for the large circuits (measured at 1.6 M constraints: five concurrent accumulations
  // are slower than four followed by one, 17.7 vs 17.4 ms) — behind B1, whose accumulation is the first of the three G1
  // ones to start and to finish (behind A: +0.1 ms, behind C: +0.4 ms; ICICLE_SNARK_H_BEHIND=0/1/2); on g3 right behind its own sort for the small ones and for multi-GPU
  // shards, where the GPU is far from full and only the length of the chains counts (200 k: 4.3 → 3.9 ms)
  static const int h_behind_cfg = getenv("ICICLE_SNARK_H_BEHIND") ? atoi(getenv("ICICLE_SNARK_H_BEHIND")) : 1; // 0 = A, 1 = B1, 2 = C
  const int h_behind = h_behind_cfg < 0 || h_behind_cfg > 2 ? 1 : h_behind_cfg;
  const bool h_own = z->H.len() <= (1u << 19);
  // Large circuits: H is queued behind the whole chain of the MSM in front of it.  ICICLE_SNARK_H_AFTER=acc lets it wait
  // only for that MSM's ACCUMULATION kernel (prof->ev[2]) on its own stream: the timeline shows ≈1 ms between the end of the
  // four witness accumulations and the start of H's (the other MSM's empty large-bucket kernels and reduction sit in
  // between), yet filling that gap makes the prove SLOWER — interleaved A/B on MI355X, 1.6 M constraints: 16.35–16.48 ms
  // against 16.0–16.2 ms: H then overlaps the tails of three other accumulations, and five concurrent accumulations are
  // less efficient than four followed by one.
  static const bool h_after_chain = !(getenv("ICICLE_SNARK_H_AFTER") && !strcmp(getenv("ICICLE_SNARK_H_AFTER"), "acc"));
  const bool h_chain = !h_own && h_after_chain;
  hipStream_t gh = h_chain ? st3[h_behind] : g3;
  if (h_chain) P_HIP(hipStreamWaitEvent(gh, z->ev_sort_h, 0));
  else if (!h_own) P_HIP(hipStreamWaitEvent(gh, prof[order[h_behind]]->ev[2], 0));
  // H's reduction is the last kernel of the prove: the single-kernel form (0.49 ms alone) even when the two-level
  // reduction is switched on for the others
  static const int h_reduce_pref = getenv("ICICLE_SNARK_H_REDUCE") ? atoi(getenv("ICICLE_SNARK_H_REDUCE")) : 1;
  plan_h.reduce_pref = h_reduce_pref;
  fill(prof[4], plan_h, 0);
  P_ICICLE(msm_g1_partials(&plan_h, z->H.d_points, 2, 0, gh, DP + 4 * PARTIALS_STRIDE, prof[4], z->H.len()));
  (void)hipEventRecord(prof[4]->ev[3], gh);
  prof[4]->valid = true;
  mark("h");
  // Each MSM's partial sums go to pinned memory on ITS OWN stream as soon as its reduction is done, and a host
  // thread per MSM waits for that copy and runs the Horner tail — the tails of the early finishers (B2, A, B1, C)
  // overlap the GPU work still in flight; only the last one (H) is exposed.
  uint32_t Ww = 0, bw1 = 0, Wb = 0, bb1 = 0, bw2 = 0, Wh = 0, bh = 0;
  const size_t by1 = msm_partials_bytes(&plan_w, false, &Ww, &bw1), byb = msm_partials_bytes(&plan_b12, false, &Wb, &bb1), by2 = msm_partials_bytes(&plan_b12, true, &Wb, &bw2),
               byh = msm_partials_bytes(&plan_h, false, &Wh, &bh);
  const size_t sizes[5] = {by1, byb, by2, by1, byh};
  hipStream_t st5[5] = {st3[0], st3[1], g2, st3[2], gh};
  if (h_chain) {
    // the copy of the MSM in front of H must not wait for H (same stream): its partials were complete at its ev[3], copy them on g3 instead
    P_HIP(hipStreamWaitEvent(g3, prof[order[h_behind]]->ev[3], 0));
    st5[order[h_behind]] = g3;
  }
  for (int k = 0; k < 5; k++) {
    P_HIP(hipMemcpyAsync(z->h_partials + k * PARTIALS_STRIDE, DP + k * PARTIALS_STRIDE, sizes[k], hipMemcpyDeviceToHost, st5[k]));
    P_HIP(hipEventRecord(z->ev_done[k], st5[k]));
  }
  for (int k = 0; k < 5; k++) P_HIP(hipStreamWaitEvent(g1, z->ev_done[k], 0));
  P_HIP(hipEventRecord(z->ev[3], g1)); // end of the MSM phase: every chain has delivered its partial sums (timing only)
  mark("copies");
  {
    const uint8_t* HP = z->h_partials;
    const int cw = plan_w.g.c, cb = plan_b12.g.c, ch = plan_h.g.c;
    hipEvent_t* evd = z->ev_done;
    const int dev = z->device_id;
    const MsmGeom gw = plan_w.g, gb = plan_b12.g, gh = plan_h.g;
    auto g1tail = [&](int k, uint32_t W, uint32_t bpw, int c, size_t off) {
      (void)hipSetDevice(dev);
      (void)hipEventSynchronize(evd[k]);
      const MsmGeom& gg = k == 4 ? gh : k == 1 ? gb : gw;
      if (gg.tab) msm_g1_host_tail_tab(HP + k * PARTIALS_STRIDE, W, bpw, gg.NBb, (bn254_projective_t*)(out_points + off));
      else msm_g1_host_tail(HP + k * PARTIALS_STRIDE, W, 1, c, gg.wide, (bn254_projective_t*)(out_points + off));
      if (et && k < 2) {
        while (!et->bl_ready.load(std::memory_order_acquire)) std::this_thread::yield();
        bn254_projective_t p;
        memcpy(&p, out_points + off, sizeof p);
        bn254_ecadd(&p, (const bn254_projective_t*)(k == 0 ? &z->vk_alpha_1 : &z->vk_beta_1), &p);
        bn254_ecadd(&p, k == 0 ? &et->bl->d1r : &et->bl->d1s, &p);
        bn254_mul_scalar(&p, k == 0 ? &et->bl->s : &et->bl->r, k == 0 ? &et->ta : &et->tb);
        et->done.fetch_add(1, std::memory_order_release);
      }
    };
    std::thread t0(g1tail, 0, Ww, bw1, cw, (size_t)0);
    std::thread t1(g1tail, 1, Wb, bb1, cb, (size_t)96);
    std::thread t3(g1tail, 3, Ww, bw1, cw, (size_t)384);
    std::thread t2([&] {
      (void)hipSetDevice(dev);
      (void)hipEventSynchronize(evd[2]);
      if (gb.tab) msm_g2_host_tail_tab(HP + 2 * PARTIALS_STRIDE, Wb, bw2, gb.NBb, (bn254_g2_projective_t*)(out_points + 192));
      else msm_g2_host_tail(HP + 2 * PARTIALS_STRIDE, Wb, 1, cb, gb.wide, (bn254_g2_projective_t*)(out_points + 192));
    });
    g1tail(4, Wh, bh, ch, 480);
    t0.join(); t1.join(); t2.join(); t3.join();
  }
  mark("tails");
  P_HIP(hipStreamSynchronize(g1));
  P_HIP(hipStreamSynchronize(g2));
  P_HIP(hipStreamSynchronize(g3));
  P_HIP(hipStreamSynchronize(z->s_g4));
  P_HIP(hipStreamSynchronize(z->s_g5));
  P_HIP(hipStreamSynchronize(z->s_qap));
  drain.armed = false;
  if (getenv("ICICLE_SNARK_TRACE_LARGE")) {
    // debug: large buckets / work items / threshold of the three digit sorts of this prove
    for (const SortPlan* pl : {(const SortPlan*)&plan_w, (const SortPlan*)&plan_b12, (const SortPlan*)&plan_h}) {
      uint32_t nl[4] = {0, 0, 0, 0};
      if (pl->n_large) (void)hipMemcpy(nl, pl->n_large, sizeof nl, hipMemcpyDeviceToHost);
      fprintf(stderr, "[large] L=%u buckets=%u thr=%u: %u large buckets, %u entries in them, %u work items (cap %u)\n", pl->L, pl->nbuckets, pl->large_thr, nl[0], nl[1], nl[2], pl->item_cap);
    }
  }
  msm_sort_release(&plan_w);
  msm_sort_release(&plan_h);
  if (z->sparse_b) msm_sort_release(&plan_b);
  {
    float a = 0, b = 0, c = 0;
    (void)hipEventElapsedTime(&a, z->ev[0], z->ev[1]);
    (void)hipEventElapsedTime(&b, z->ev[1], z->ev[2]);
    (void)hipEventElapsedTime(&c, z->ev[2], z->ev[3]);
    z->last_tm.h2d_ms = h2d_host_ms + a;
    z->last_tm.qap_ms = b;
    z->last_tm.msm_ms = c;
    z->last_tm.total_ms = ms_since(t0);
    if (tm) *tm = z->last_tm;
  }
  return 0;
}
} // namespace


extern "C" {

// ---- distributed front end: stage 1 and stage 2 (include/groth16_prover.h) ----------------------------------------------
__attribute__((visibility("default"))) int groth16_dist_supported(Groth16CacheManager* cm, const char* key)
{
  if (!cm) return 0;
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  if (!zp) return 0;
  const ZKeyCache* z = zp.get();
  const uint32_t G = (uint32_t)z->shard_count;
  if (z->H.stride <= 1 || (G != 2 && G != 4 && G != 8)) return 0;
  const uint32_t m = z->domain_size / G;
  return m % G == 0 && ntt_fusable(m) ? 1 : 0;
}

// Multi-GPU witness distribution (see include/groth16_prover.h)
__attribute__((visibility("default"))) int groth16_upload_witness_slice(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, void** d_witness, uint64_t* slice_bytes)
{
  if (!cm || !wtns || !d_witness || !slice_bytes) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  ZKeyCache* z = zp.get();
  if (!z) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  IcicleDevice dev;
  memset(&dev, 0, sizeof dev);
  strcpy(dev.type, "HIP");
  dev.id = z->device_id;
  P_ICICLE(icicle_set_device(&dev));
  Wtns w;
  if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
  if (!Fr::eq(z->r, w.q)) return fail(ERR_FORMAT, "Curve of the witness does not match the curve of the proving key");
  if (w.n_witness != z->n_vars) return fail(ERR_FORMAT, "Invalid witness length. Circuit: %u, witness: %u", z->n_vars, w.n_witness);
  z->witness_resident = false; // until groth16_witness_ready
  const uint64_t slice = witness_slice_elems(z->n_vars, z->shard_count);
  const uint64_t lo = std::min<uint64_t>(z->n_vars, slice * (uint64_t)z->shard_rank), hi = std::min<uint64_t>(z->n_vars, lo + slice);
  if (hi > lo) {
    const hipStream_t lanes[3] = {z->s_qap, z->s_g2, z->s_g3};
    if (int rc = staged_upload(z->device_id, {{z->d_witness + lo, (const uint8_t*)w.values + lo * 32, (size_t)(hi - lo) * 32}}, lanes, 3)) return rc;
  }
  *d_witness = z->d_witness;
  *slice_bytes = slice * 32;
  return 0;
}
__attribute__((visibility("default"))) int groth16_witness_ready(Groth16CacheManager* cm, const char* key)
{
  if (!cm) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  if (!zp) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  zp->witness_resident = true;
  return 0;
}

__attribute__((visibility("default"))) int groth16_dist_stage1(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, void** d_send, void** d_recv, uint32_t* rows,
                                                               uint64_t* row_bytes, uint64_t* chunk_bytes)
{
  if (!cm || !d_send || !d_recv) return fail(ERR_ARG, "null argument");
  if (!groth16_dist_supported(cm, key)) return fail(ERR_ARG, "cache entry '%s' is not a strided shard of 2, 4 or 8 (or too small) — use groth16_commitments", key ? key : "");
  std::lock_guard<std::mutex> lk(cm->mu);
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  ZKeyCache* z = zp.get();
  IcicleDevice dev;
  memset(&dev, 0, sizeof dev);
  strcpy(dev.type, "HIP");
  dev.id = z->device_id;
  P_ICICLE(icicle_set_device(&dev));
  if (int rc = ensure_domain(cm, z)) return rc;
  const uint32_t n = z->domain_size, G = (uint32_t)z->shard_count, r = (uint32_t)z->shard_rank, m = n / G;
  if (wtns) {
    Wtns w;
    if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
    if (!Fr::eq(z->r, w.q)) return fail(ERR_FORMAT, "Curve of the witness does not match the curve of the proving key");
    if (w.n_witness != z->n_vars) return fail(ERR_FORMAT, "Invalid witness length. Circuit: %u, witness: %u", z->n_vars, w.n_witness);
    const hipStream_t lanes[3] = {z->s_qap, z->s_g2, z->s_g3};
    if (int rc = staged_upload(z->device_id, {{z->d_witness, (const uint8_t*)w.values, (size_t)z->n_vars * 32}}, lanes, 3)) return rc;
    z->witness_resident = true;
  } else if (!z->witness_resident)
    return fail(ERR_ARG, "no witness given and none resident on the device");
  hipStream_t gq = z->s_qap;
  int dom_log = 0;
  const fe* tw = ntt_domain_table(&dom_log);
  if (!z->d_dist_y) {
    P_HIP(hipMalloc((void**)&z->d_dist_y, (size_t)3 * m * 32));
    P_HIP(hipMalloc((void**)&z->d_dist_recv1, (size_t)3 * m * 32));
    P_HIP(hipMalloc((void**)&z->d_dist_send2, (size_t)3 * m * 32));
    P_HIP(hipMalloc((void**)&z->d_tw1, (size_t)m * 32));
    z->device_bytes += (size_t)10 * m * 32;
    P_HIP(qap_dist_tw1(tw, 1u << dom_log, n, G, r, z->d_tw1, gq));
  }
  P_HIP(qap_spmv_strided(z->d_witness, z->d_rowptr, z->d_cols, z->d_vals, n, G, r, z->d_dist_y, gq));
  NttFuse f;
  f.scale_tab = z->d_tw1;
  P_ICICLE(ntt_fused(z->d_dist_y, m, 3, true, gq, f));
  P_HIP(hipStreamSynchronize(gq)); // the exchange runs on the communicator's stream
  *d_send = z->d_dist_y;
  *d_recv = z->d_dist_recv1;
  if (rows) *rows = 3;
  if (row_bytes) *row_bytes = (uint64_t)m * 32;
  if (chunk_bytes) *chunk_bytes = (uint64_t)(m / G) * 32;
  return 0;
}

__attribute__((visibility("default"))) int groth16_dist_stage2(Groth16CacheManager* cm, const char* key, void** d_send, void** d_recv)
{
  if (!cm || !d_send || !d_recv) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  ZKeyCache* z = zp.get();
  if (!z || !z->d_dist_recv1) return fail(ERR_ARG, "groth16_dist_stage1 has not run for '%s'", key ? key : "");
  (void)hipSetDevice(z->device_id);
  const uint32_t n = z->domain_size, G = (uint32_t)z->shard_count, r = (uint32_t)z->shard_rank;
  int dom_log = 0;
  const fe* tw = ntt_domain_table(&dom_log);
  if (!tw || (1u << dom_log) < 2 * n) return fail((int)ICICLE_INVALID_ARGUMENT, "the NTT domain was released between the stages");
  P_HIP(qap_dist_mid(z->d_dist_recv1, z->d_dist_send2, tw, 1u << dom_log, n, G, r, z->s_qap));
  P_HIP(hipStreamSynchronize(z->s_qap));
  *d_send = z->d_dist_send2;
  *d_recv = z->d_fold; // exchange 2 delivers this rank's Z rows [row][m] where stage 3 (groth16_commitments) expects them
  z->dist_ready = true;
  return 0;
}

} // extern "C"

extern "C" {

__attribute__((visibility("default"))) int groth16_sum_commitments(const uint8_t* blocks, int count, uint8_t out[GROTH16_COMMITMENTS_BYTES])
{
  if (!blocks || !out || count < 1) return fail(ERR_ARG, "bad argument");
  uint8_t acc[GROTH16_COMMITMENTS_BYTES];
  memcpy(acc, blocks, sizeof acc);
  static const int off[5] = {0, 96, 192, 384, 480};
  for (int k = 1; k < count; k++) {
    const uint8_t* b = blocks + (size_t)k * GROTH16_COMMITMENTS_BYTES;
    for (int j = 0; j < 5; j++) {
      if (j == 2) bn254_g2_ecadd((const bn254_g2_projective_t*)(acc + off[j]), (const bn254_g2_projective_t*)(b + off[j]), (bn254_g2_projective_t*)(acc + off[j]));
      else bn254_ecadd((const bn254_projective_t*)(acc + off[j]), (const bn254_projective_t*)(b + off[j]), (bn254_projective_t*)(acc + off[j]));
    }
  }
  memcpy(out, acc, sizeof acc);
  return 0;
}

} // extern "C"

namespace {
// Uniform scalar in [0, r) from the kernel's CSPRNG (getrandom(2), /dev/urandom as fallback): 254 random bits, rejected
// while ≥ r (acceptance 0.756).  The reference draws r, s from an unseeded mt19937 (ScalarCfg::generate_random →
// utils/rand_gen.h:5) — zero-knowledge must not rest on a 32-bit-seeded, predictable generator, so the production path
// does not restate that; bn254_generate_scalars (test data, like the reference's) keeps the Mersenne twister.
bool secure_scalar(bn254_scalar_t* out)
{
  for (int tries = 0; tries < 256; tries++) {
    fe v;
    size_t got = 0;
    while (got < 32) {
      const ssize_t k = getrandom((uint8_t*)v.l + got, 32 - got, 0);
      if (k < 0) {
        if (errno == EINTR) continue;
        break;
      }
      got += (size_t)k;
    }
    if (got < 32) {
      FILE* f = fopen("/dev/urandom", "rb");
      if (!f) return false;
      const size_t k = fread(v.l, 1, 32, f);
      fclose(f);
      if (k != 32) return false;
    }
    v.l[7] &= 0x3fffffffu;
    if (Fr::is_canonical(v)) {
      memcpy(out, v.l, 32);
      return true;
    }
  }
  return false;
}
int compute_blinding(const ZKeyCache* z, const uint8_t* r_in, const uint8_t* s_in, Blinding* b)
{
  bn254_scalar_t rs[2];
  if ((!r_in && !secure_scalar(&rs[0])) || (!s_in && !secure_scalar(&rs[1]))) return fail(ERR_IO, "no entropy source for the blinding scalars"); // src/proof_helper.rs:276
  if (r_in) memcpy(&rs[0], r_in, 32);
  if (s_in) memcpy(&rs[1], s_in, 32);
  b->r = rs[0];
  b->s = rs[1];
  const bn254_projective_t* delta1 = (const bn254_projective_t*)&z->vk_delta_1;
  const bn254_g2_projective_t* delta2 = (const bn254_g2_projective_t*)&z->vk_delta_2;
  bn254_mul_scalar(delta1, &b->r, &b->d1r);
  bn254_mul_scalar(delta1, &b->s, &b->d1s);
  bn254_mul_scalar(&b->d1r, &b->s, &b->d1rs);
  bn254_g2_mul_scalar(delta2, &b->s, &b->d2s);
  return 0;
}
int assemble_impl(ZKeyCache* z, const void* wtns, size_t wtns_len, const uint8_t* points, const Blinding& bl, char* proof_json, size_t proof_cap, char* public_json, size_t public_cap, const EarlyTerms* et = nullptr);
} // namespace

extern "C" {

__attribute__((visibility("default"))) int groth16_assemble_proof(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, const uint8_t points[GROTH16_COMMITMENTS_BYTES],
                                                                  const uint8_t* r_in, const uint8_t* s_in, char* proof_json, size_t proof_cap, char* public_json, size_t public_cap)
{
  if (!cm || !wtns || !points) return fail(ERR_ARG, "null argument");
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  ZKeyCache* z = zp.get();
  if (!z) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  Blinding bl;
  if (int rc = compute_blinding(z, r_in, s_in, &bl)) return rc;
  return assemble_impl(z, wtns, wtns_len, points, bl, proof_json, proof_cap, public_json, public_cap);
}

} // extern "C"

namespace {
int assemble_impl(ZKeyCache* z, const void* wtns, size_t wtns_len, const uint8_t* points, const Blinding& bl, char* proof_json, size_t proof_cap, char* public_json, size_t public_cap, const EarlyTerms* et)
{
  Wtns w;
  if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
  if (w.n_witness != z->n_vars) return fail(ERR_FORMAT, "Invalid witness length");
  typedef bn254_projective_t P1;
  typedef bn254_g2_projective_t P2;
  P1 pi_a, pi_b1, pi_c, pi_h, t2;
  P2 pi_b;
  memcpy(&pi_a, points, 96);
  memcpy(&pi_b1, points + 96, 96);
  memcpy(&pi_b, points + 192, 192);
  memcpy(&pi_c, points + 384, 96);
  memcpy(&pi_h, points + 480, 96);
  const P1* alpha1 = (const P1*)&z->vk_alpha_1;
  const P1* beta1 = (const P1*)&z->vk_beta_1;
  const P2* beta2 = (const P2*)&z->vk_beta_2;
  // src/proof_helper.rs:280-283
  bn254_ecadd(&pi_a, alpha1, &pi_a);
  bn254_ecadd(&pi_a, &bl.d1r, &pi_a);           // pi_a = A + α1 + δ1·r
  bn254_g2_ecadd(&pi_b, beta2, &pi_b);
  bn254_g2_ecadd(&pi_b, &bl.d2s, &pi_b);        // pi_b = B2 + β2 + δ2·s
  bn254_ecadd(&pi_b1, beta1, &pi_b1);
  bn254_ecadd(&pi_b1, &bl.d1s, &pi_b1);         // pi_b1 = B1 + β1 + δ1·s
  bn254_ecadd(&pi_c, &pi_h, &pi_c);             // C + H
  if (et && et->done.load(std::memory_order_acquire) == 2) {
    bn254_ecadd(&pi_c, &et->ta, &pi_c); // computed by the tail threads of A and B1 while H was still running
    bn254_ecadd(&pi_c, &et->tb, &pi_c);
  } else {
    P1 ta, tb;
    std::thread th([&] { bn254_mul_scalar(&pi_a, &bl.s, &ta); }); // pi_a·s  ∥  pi_b1·r
    bn254_mul_scalar(&pi_b1, &bl.r, &tb);
    th.join();
    bn254_ecadd(&pi_c, &ta, &pi_c);
    bn254_ecadd(&pi_c, &tb, &pi_c);
  }
  bn254_ecsub(&pi_c, &bl.d1rs, &pi_c);          // − δ1·r·s
  (void)t2;
  bn254_affine_t a_aff, c_aff;
  bn254_g2_affine_t b_aff;
  bn254_to_affine(&pi_a, &a_aff);
  bn254_g2_to_affine(&pi_b, &b_aff);
  bn254_to_affine(&pi_c, &c_aff);
  auto dec = [](const void* p) {
    fe v;
    memcpy(v.l, p, 32);
    return to_decimal(v);
  };
  // serde_json::to_writer_pretty of a Value built with json!(proof): object keys sorted (BTreeMap), 2-space indent
  std::string pj = "{\n  \"curve\": \"bn128\",\n";
  pj += "  \"pi_a\": [\n    \"" + dec(&a_aff.x) + "\",\n    \"" + dec(&a_aff.y) + "\",\n    \"1\"\n  ],\n";
  pj += "  \"pi_b\": [\n    [\n      \"" + dec(&b_aff.x.c0) + "\",\n      \"" + dec(&b_aff.x.c1) + "\"\n    ],\n    [\n      \"" + dec(&b_aff.y.c0) + "\",\n      \"" + dec(&b_aff.y.c1) +
        "\"\n    ],\n    [\n      \"1\",\n      \"0\"\n    ]\n  ],\n";
  pj += "  \"pi_c\": [\n    \"" + dec(&c_aff.x) + "\",\n    \"" + dec(&c_aff.y) + "\",\n    \"1\"\n  ],\n";
  pj += "  \"protocol\": \"groth16\"\n}";
  // public signals: witness[1..=n_public] as decimal strings — src/proof_helper.rs:297-307
  std::string qj = z->n_public ? "[\n" : "[]";
  for (uint32_t i = 1; i <= z->n_public; i++) {
    qj += "  \"" + dec(w.values + (size_t)i * 32) + "\"";
    qj += i == z->n_public ? "\n]" : ",\n";
  }
  int need = 0;
  if (proof_json) {
    if (pj.size() + 1 > proof_cap) need = (int)pj.size() + 1;
    else memcpy(proof_json, pj.c_str(), pj.size() + 1);
  }
  if (public_json) {
    if (qj.size() + 1 > public_cap) need = need > (int)qj.size() + 1 ? need : (int)qj.size() + 1;
    else memcpy(public_json, qj.c_str(), qj.size() + 1);
  }
  if (need) return fail(need, "output buffer too small (need %d bytes)", need);
  return 0;
}
} // namespace

extern "C" {

// groth16_prove_mem with the option to re-use the witness already resident on the device (bench.py: inputs in HBM)
__attribute__((visibility("default"))) int groth16_prove_resident(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, int wtns_resident, const uint8_t* r, const uint8_t* s,
                                                                   char* proof_json, size_t proof_cap, char* public_json, size_t public_cap, Groth16Timings* tm);

__attribute__((visibility("default"))) int groth16_prove_mem(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, const uint8_t* r, const uint8_t* s, char* proof_json, size_t proof_cap,
                                                             char* public_json, size_t public_cap, Groth16Timings* tm)
{
  return groth16_prove_resident(cm, key, wtns, wtns_len, 0, r, s, proof_json, proof_cap, public_json, public_cap, tm);
}

__attribute__((visibility("default"))) int groth16_prove_resident(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, int wtns_resident, const uint8_t* r, const uint8_t* s,
                                                                   char* proof_json, size_t proof_cap, char* public_json, size_t public_cap, Groth16Timings* tm)
{
  uint8_t pts[GROTH16_COMMITMENTS_BYTES];
  const auto t0 = std::chrono::steady_clock::now();
  if (!cm || !wtns) return fail(ERR_ARG, "null argument");
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  ZKeyCache* z = zp.get();
  if (!z) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  // A sharded cache holds only this rank's point range: its commitments are PARTIAL sums and a proof assembled from them
  // alone would be well-formed but invalid.  The only correct sequence for shards is groth16_commitments → all-gather →
  // groth16_sum_commitments → groth16_assemble_proof (parallel.py).
  if (z->shard_count != 1) return fail(ERR_ARG, "cache entry '%s' is shard %d of %d: use groth16_commitments + groth16_sum_commitments + groth16_assemble_proof", key ? key : "", z->shard_rank, z->shard_count);
  // r, s and the commitment-independent blinding terms on a host thread while the GPU computes the commitments
  Blinding bl;
  EarlyTerms et;
  et.bl = &bl;
  int bl_rc = 0;
  std::thread th([&] {
    bl_rc = compute_blinding(z, r, s, &bl);
    et.bl_ready.store(true, std::memory_order_release);
  });
  int rc = commitments_impl(cm, key, wtns_resident ? nullptr : wtns, wtns_len, pts, tm, &et);
  th.join();
  if (rc) return rc;
  if (bl_rc) return fail(bl_rc, "no entropy source for the blinding scalars"); // (the message was set on the helper thread)
  const auto ta = std::chrono::steady_clock::now();
  rc = assemble_impl(z, wtns, wtns_len, pts, bl, proof_json, proof_cap, public_json, public_cap, &et);
  if (getenv("ICICLE_SNARK_TRACE_HOST")) fprintf(stderr, "[host] assemble %8.1f us (after %8.1f us)\n", ms_since(ta) * 1e3, std::chrono::duration<double, std::micro>(ta - t0).count());
  if (tm) tm->total_ms = ms_since(t0);
  return rc;
}

// groth16_prove — src/lib.rs:33-61
__attribute__((visibility("default"))) int groth16_prove(const char* witness_path, const char* zkey_path, const char* proof_path, const char* public_path, const char* device, Groth16CacheManager* cm)
{
  if (!witness_path || !zkey_path || !proof_path || !public_path || !device || !cm) return fail(ERR_ARG, "null argument");
  const auto t0 = std::chrono::steady_clock::now();
  // try_load_and_set_backend_device — src/lib.rs:25-31
  IcicleDevice dev;
  memset(&dev, 0, sizeof dev);
  strncpy(dev.type, device, sizeof dev.type - 1);
  dev.id = 0;
  if (strcmp(device, "CPU") != 0) P_ICICLE(icicle_load_backend_from_env_or_default());
  P_ICICLE(icicle_set_device(&dev)); // "CPU" (or anything but HIP/CUDA) fails here: no CPU fallback
  const std::string key = std::string(zkey_path) + "_" + device; // src/lib.rs:44
  if (!groth16_cache_contains(cm, key.c_str()))
    if (int rc = groth16_cache_load_file(cm, key.c_str(), zkey_path, 0, 0, 1)) return rc;
  MappedFile wf;
  if (int rc = wf.open_ro(witness_path)) return rc;
  std::vector<char> pj(4096), qj(256);
  {
    const std::shared_ptr<ZKeyCache> z = find(cm, key.c_str());
    if (!z) return fail(ERR_NOCACHE, "no cache entry '%s'", key.c_str());
    qj.resize(64 + (size_t)z->n_public * 84);
  }
  // the witness values go from the page cache straight into the upload workers' pinned buffers (pread) instead of being copied
  // out of the mapping, which is then only touched for the header and the public signals: −0.3 ms per prove at 1.6 M
  // constraints (page faults of a 51 MB mapping); ICICLE_SNARK_FILE_PREAD=0 copies from the mapping
  static const bool file_pread = !(getenv("ICICLE_SNARK_FILE_PREAD") && atoi(getenv("ICICLE_SNARK_FILE_PREAD")) == 0);
  if (file_pread) staged_copy_file_hint(wf.data, wf.len, wf.fd);
  const int prc = groth16_prove_mem(cm, key.c_str(), wf.data, wf.len, nullptr, nullptr, pj.data(), pj.size(), qj.data(), qj.size(), nullptr);
  if (file_pread) staged_copy_file_hint(nullptr, 0, -1);
  if (prc) return prc;
  for (int k = 0; k < 2; k++) {
    const char* path = k ? public_path : proof_path;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(ERR_IO, "cannot write %s", path);
    fputs(k ? qj.data() : pj.data(), f);
    fclose(f);
  }
  static const bool quiet = getenv("ICICLE_SNARK_QUIET") && atoi(getenv("ICICLE_SNARK_QUIET")) != 0;
  if (!quiet) {
    printf("proof took: %.3fms\n", ms_since(t0)); // src/lib.rs:58
    fflush(stdout);
  }
  return 0;
}

} // extern "C"
