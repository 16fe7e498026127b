// prover_internal.h — types and helpers shared by the translation units of the prover host
// (containers.cpp, cache.cpp, prover.cpp, assemble.cpp, multi.cpp).  Nothing here crosses the C ABI.
//
//   containers.cpp  snarkjs .zkey / .wtns containers        ← src/file_wrapper.rs:45-208, src/zkey.rs:47-85
//   cache.cpp       CacheManager::compute (ZKeyCache build) ← src/cache.rs:117-241
//   prover.cpp      construct_r1cs + groth16_commitments on ONE device, the C API        ← src/proof_helper.rs:31-241, src/lib.rs:33-61
//   assemble.cpp    blinding, affine conversion, JSON       ← src/proof_helper.rs:274-316, src/conversions.rs:30-56
//   multi.cpp       the same prove over a GROUP of devices in one process (device string "HIP:0-7"), SURVEY.md §8e
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <stdarg.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/groth16_prover.h"
#include "../common.h"
#include "../ec.h"
#include "../msm_plan.h"
#include "../ntt_fuse.h"
#include "../workers.h"
#include "cold_feed.h"
#include "qap.h"

namespace isnark {
const bn254::fe* ntt_domain_table(int* log_n); // ntt.hip

namespace prover {
using namespace bn254;

// ---- errors: one thread-local text per thread (groth16_last_error); worker threads hand theirs to the caller
int fail(int code, const char* fmt, ...);
const char* last_error_text();
void set_error_text(const char* text);
#define P_HIP(call)                                                                                                              \
  do {                                                                                                                           \
    hipError_t e__ = (call);                                                                                                     \
    if (e__ != hipSuccess) return ::isnark::prover::fail((int)ICICLE_UNKNOWN_ERROR, "%s: %s", #call, hipGetErrorString(e__));    \
  } while (0)
#define P_ICICLE(call)                                                                                                           \
  do {                                                                                                                           \
    eIcicleError e__ = (call);                                                                                                   \
    if (e__ != ICICLE_SUCCESS) return ::isnark::prover::fail((int)e__, "%s failed (%d): %s", #call, (int)e__, icicle_snark_last_error()); \
  } while (0)
enum { ERR_IO = -1, ERR_FORMAT = -2, ERR_ARG = -3, ERR_NOCACHE = -4 };

inline double ms_since(std::chrono::steady_clock::time_point t0)
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// ---- containers (containers.cpp)
struct Section {
  const uint8_t* p = nullptr;
  uint64_t size = 0;
  int count = 0;
};
int read_sections(const uint8_t* data, size_t len, const char* type, uint32_t max_version, std::vector<Section>& out);
int unique_section(const std::vector<Section>& s, size_t id, const Section** sec);
struct MappedFile {
  const uint8_t* data = nullptr;
  size_t len = 0;
  int fd = -1;
  ~MappedFile();
  int open_ro(const char* path);
};
struct Wtns {
  uint32_t n8 = 0, n_witness = 0;
  fe q;
  const uint8_t* values = nullptr; // n_witness × 32 B standard form
};
int parse_wtns(const uint8_t* data, size_t len, Wtns& w);

// ---- the device-resident cache of one zkey on ONE device (cache.cpp)
constexpr size_t PARTIALS_STRIDE = 64 * 16 * 256; // ≥ W·bpw·sizeof(XYZZ) for any geometry (W ≤ 64, bpw ≤ 16, G2 256 B)

struct Shard {
  uint32_t lo = 0, hi = 0; // [lo, hi) of the full base array
  void* d_points = nullptr; // internal encoding; in table mode W rows of len() points (row w = 2^(c·w)·P, msm_plan.h)
  uint32_t stride = 1, first = 0; // H of a power-of-two shard count: elements first + k·stride, k < len() (lo = 0, hi = len)
  uint32_t len() const { return hi - lo; }
};

// Deferred build of a key's fixed-base tables (cache.cpp).  The tables cost 0.26 s at 1.6 M constraints and buy 2–3 ms per
// prove: a key is therefore usable — in the classic layout, bases converted in place — as soon as its sections are on the
// device, and a worker thread builds the five tables on a low-priority stream behind (and beside) the first proofs.  The prove
// that finds them complete swaps pointers and geometry in, all five at once (adopt_tables; the caller holds the manager's mutex).
constexpr int TABLE_BUILD_GRACE_MS = 150;
struct TableBuild {
  std::thread th;
  std::atomic<int> state{0}; // 0 nothing pending, 1 building, 2 complete (fresh[] valid, not adopted yet), 3 failed / cancelled
  std::atomic<bool> cancel{false};
  std::atomic<bool> go{false}; // set at the end of the key's first prove: the build starts behind it, not beside it (or after TABLE_BUILD_GRACE_MS without one)
  std::atomic<bool> hold{false}; // cold pipeline: the base arrays are still being uploaded — nothing may read them before cold_prove clears this (neither `go` nor the grace time counts)
  void* fresh[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // A, B1, B2, C, H
  MsmGeom gw, gh;            // the table geometries the build works towards
  int dense_c = 0;           // digit width of the dense witness geometry (gw may be narrower: the first prove's witness was light)
  uint64_t pending_bytes = 0; // table bytes already counted in the entry's device_bytes while the first build is under way
  // A first build that takes a NARROWER witness digit than the dense one pending_bytes was sized for (more table rows) asks here:
  // `narrow_room` = bytes beyond pending_bytes it may take (set by whoever admitted the key: what the cache budget has left, or no
  // limit), `extra_bytes` = what it took (counted by evict_for_budget until adopt_tables has corrected device_bytes)
  std::atomic<uint64_t> narrow_room{0};
  std::atomic<uint64_t> extra_bytes{0};
  bool witness_only = false; // a re-build of the four witness tables with another digit width (the key follows its witnesses): H stays
  double build_ms = 0;       // wall clock of the build (beside whatever proves ran meanwhile)
};

// (ColdFeed: prover/cold_feed.h)
struct ZKeyCache {
  // header — src/zkey.rs:6-21
  uint32_t n8q = 0, n8r = 0, n_vars = 0, n_public = 0, domain_size = 0, n_coef = 0;
  fe q, r;
  G1::P vk_alpha_1, vk_beta_1, vk_delta_1; // standard form projective (host)
  G2::P vk_beta_2, vk_gamma_2, vk_delta_2;
  // device
  int device_id = 0, shard_rank = 0, shard_count = 1;
  bool in_group = false; // a shard of an in-process device group (multi.cpp), as opposed to the one shard of a rank-per-GPU process
  MsmGeom geom_w, geom_h; // window geometry of the witness MSMs (A, B1, B2, C) and of the H MSM; geom_w follows the witnesses (cache.cpp)
  int geom_w_default_c = 0;        // digit width of the dense geometry chosen at cache build
  uint32_t* h_stats = nullptr;     // pinned: offset and count of the last bucket of the witness sort = its entry count
  uint64_t witness_entries = 0;    // non-zero digits of the witness range in the most recent prove
  uint32_t proves_since_rebuild = 0;
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
  // d_fold holds the Z rows of this rank for the witness now resident: the next commitments call WITHOUT a new witness skips
  // its own inverse transform + fold.  Set only by the caller's confirmation that exchange 2 has delivered
  // (groth16_dist_exchange_done, or the in-process group prove); cleared by every witness upload and every stage 1.
  bool dist_ready = false;
  bool dist_stage2_done = false; // stage 2 has run for the resident witness (what groth16_dist_exchange_done checks)
  fe* d_skeys = nullptr;   // n: n⁻¹·g^i — 1/n and the coset keys folded into the inverse transform's last pass (ntt_fuse.h); built on first use
  uint8_t* d_partials = nullptr; // 5 × PARTIALS_STRIDE: per-window partial sums of the five MSMs
  uint8_t* h_partials = nullptr; // pinned mirror
  hipStream_t s_g1 = nullptr, s_g2 = nullptr, s_g3 = nullptr, s_g4 = nullptr, s_g5 = nullptr, s_qap = nullptr;
  hipEvent_t ev_witness = nullptr, ev_sort = nullptr, ev_sort_h = nullptr, ev[4] = {nullptr, nullptr, nullptr, nullptr}, ev_done[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // head / tail of a witness on its way in (prover.cpp): head resident (pinned source), the head's four accumulations done;
  // with timing: end of the head's chain, end of the upload
  hipEvent_t ev_lfork[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}, ev_ljoin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // large-bucket kernels on a side stream (msm_plan.h: LargeSide)
  hipEvent_t ev_head_in = nullptr, ev_head_done = nullptr, ev_t_head_start = nullptr, ev_t_head_end = nullptr, ev_t_witness = nullptr;
  int head_units = -1; // upload chunks of the witness that are sorted and accumulated while the rest is still on its way (follows the measured upload, prover.cpp); −1: not chosen yet
  uint64_t device_bytes = 0;
  bool witness_resident = false; // d_witness holds the witness of the last call (wtns == NULL reuses it)
  bool witness_event_set = false; // ev_witness was already recorded for the resident witness (behind a device-side all-gather)
  // device group: this shard's own 1/count of the witness is in place (recorded behind its PCIe upload, before the all-gather);
  // with slice_aligned — the shard's point range IS that slice — its witness sort and accumulations wait for nothing else
  hipEvent_t ev_own_slice = nullptr;
  bool own_slice_event_set = false, slice_aligned = false;
  Groth16Timings last_tm = {0, 0, 0, 0}; // phase timings of the most recent prove (groth16_last_timings)
  MsmProfile prof[6] = {};               // A, B1, B2, C, H of the most recent prove + [5] the digit sort of the witness HEAD (ev[0] → ev[4]; L = 0 without one): this entry's own slots (the shards of a group may share a device)
  uint64_t last_use = 0;                 // CacheManager LRU clock
  TableBuild tb;                         // deferred fixed-base tables (single-device keys)
  ColdFeed* feed = nullptr;              // set for the ONE prove that runs while the key's sections are still arriving (cold pipeline)

  ~ZKeyCache();
};

// elements per rank when the witness is uploaded in shard_count slices
inline uint64_t witness_slice_elems(uint32_t n_vars, int count) { return ((uint64_t)n_vars + count - 1) / count; }

// digit width of the witness MSMs adapted to the witnesses seen (cache.cpp)
int witness_digit_target(const ZKeyCache* z, uint64_t entries);
int rebuild_witness_tables(ZKeyCache* z, int c_new);
// the same in the background (round 5): a worker builds the four tables beside the proves of the key, which go on with the tables
// they have, and a later prove adopts them (adopt_tables).  Returns at once; nothing happens when a build is already under way.
void start_witness_rebuild(ZKeyCache* z, int c_new);

// deferred tables: 1 = the key proves with its tables (or has none coming: classic layout for good), 0 = still building.
// `wait`: block until the build has ended.  Swaps complete tables in; the caller holds the manager's mutex (no prove in flight).
int adopt_tables(ZKeyCache* z, bool wait);

// CacheManager::compute — src/cache.rs:117-241.  `defer_tables`: return once the key can prove in the classic layout and build
// the fixed-base tables on a worker thread (single-device keys; ICICLE_SNARK_DEFER_TABLES=0 builds them before returning)
// `cold` (single-device keys with deferred or no tables): return as soon as the buffers exist and an uploader task has been started —
// the sections, the CSR and the witness of `cold->wtns` arrive behind the stages of cold->feed; the caller waits for cold->task
// before it lets go of the zkey / witness memory.
struct ColdUpload {
  ColdFeed feed;
  HostTask task;                      // the uploader (a pooled worker; run inline when none can be had)
  // witness: standard-form values of the .wtns image the prove will run on
  const uint8_t* wtns_values = nullptr;
  size_t wtns_bytes = 0;
  // the two images when they are mapped files (staging workers pread() instead of copying out of the mapping); fd < 0: plain memory
  const void *zkey_base = nullptr, *wtns_base = nullptr;
  size_t zkey_len = 0, wtns_len = 0;
  int zkey_fd = -1, wtns_fd = -1;
  hipStream_t lanes[2] = {nullptr, nullptr};
  bool started = false;
};
void cold_upload_wait(ColdUpload* cu); // blocks until the uploader task has ended (no-op when it never started); returns its lanes to the pool
int build_cache(const uint8_t* data, size_t len, int device_id, int rank, int count, std::unique_ptr<ZKeyCache>& out, bool defer_tables = false, ColdUpload* cold = nullptr);
typedef CopyJob UploadJob;
int staged_upload(int device_id, const std::vector<UploadJob>& jobs, const hipStream_t* lanes_in = nullptr, int n_lanes = 0, StagedProgress* progress = nullptr);

// ---- blinding and proof assembly (assemble.cpp)
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
  // pi_a and pi_b complete (blinded, affine, as the decimal strings of proof.json) as soon as A's and B2's tails have run —
  // milliseconds before H's: only pi_c is left for assemble_impl after the last kernel
  std::string a_dec[2], b_dec[4];
  std::atomic<bool> a_ready{false}, b_ready{false};
};
// pi_a = A + α1 + δ1·r / pi_b = B2 + β2 + δ2·s → affine → decimal strings (assemble.cpp)
void early_pi_a(const ZKeyCache* z, const Blinding& bl, const bn254_projective_t* a_plus_alpha_d1r, EarlyTerms* et);
void early_pi_b(const ZKeyCache* z, const Blinding& bl, const bn254_g2_projective_t* b2_sum, EarlyTerms* et);
int compute_blinding(const ZKeyCache* z, const uint8_t* r_in, const uint8_t* s_in, Blinding* b);
int assemble_impl(const ZKeyCache* z, const void* wtns, size_t wtns_len, const uint8_t* points, const Blinding& bl, char* proof_json, size_t proof_cap, char* public_json, size_t public_cap,
                  const EarlyTerms* et = nullptr);

// ---- one device (prover.cpp)
struct DeviceGroup; // multi.cpp
} // namespace prover
} // namespace isnark

// The reference's CacheManager (src/cache.rs:110-115) maps "{zkey_path}_{device}" to a ZKeyCache.  Here an entry is either
// ONE device's cache or a GROUP of shard caches (one per device of a "HIP:a-b" device string) that prove together.
struct Groth16CacheManager {
  std::mutex mu;     // serialises cache builds and proves (one device pipeline per manager)
  std::mutex map_mu; // guards `cache` / `groups`; entries are shared_ptr so that an evict cannot free a key a prove still uses
  std::map<std::string, std::shared_ptr<isnark::prover::ZKeyCache>> cache;
  std::map<std::string, std::shared_ptr<isnark::prover::DeviceGroup>> groups;
  std::map<int, uint32_t> domain_n; // device → domain_size its NTT domain was last initialised for (get_cache, src/cache.rs:242-256)
  uint64_t clock = 0;               // LRU clock (last_use of the entries)
  uint64_t budget_bytes = 0;        // device-memory budget per device for cached keys (0 = none): least recently used keys are evicted
  std::thread warm;                 // prewarm_device on the device the manager was created for (joined by the first cache load and by free)
  ~Groth16CacheManager()
  {
    if (warm.joinable()) warm.join();
  }
};

namespace isnark {
namespace prover {
std::shared_ptr<ZKeyCache> find(Groth16CacheManager* cm, const char* key);
std::shared_ptr<DeviceGroup> find_group(Groth16CacheManager* cm, const char* key);
int set_active_device(int device_id);
int ensure_domain(Groth16CacheManager* cm, const ZKeyCache* z); // works on the calling thread's active device (= z->device_id)
int ensure_domain_for(Groth16CacheManager* cm, int device_id, uint32_t domain_size);
void evict_for_budget(Groth16CacheManager* cm, int device, uint64_t need);
uint64_t budget_room(Groth16CacheManager* cm, int device);
// the shard pipeline of one device; caller holds cm->mu.  wtns == NULL: the witness (and, with z->dist_ready, the Z rows) are resident
int shard_commitments(Groth16CacheManager* cm, ZKeyCache* z, const void* wtns, size_t wtns_len, uint8_t* out_points, Groth16Timings* tm, EarlyTerms* et);
// distributed front end, stages without the host synchronisation of the C API (the exchange is enqueued on z->s_qap)
int shard_upload_slice(ZKeyCache* z, const Wtns& w);
int shard_dist_stage1(Groth16CacheManager* cm, ZKeyCache* z);
int shard_dist_stage2(ZKeyCache* z);
bool shard_dist_supported(const ZKeyCache* z);

// ---- device groups (multi.cpp)
// "HIP", "HIP:3", "HIP:0-7", "HIP:0,2,5", "CUDA:0-3" (alias) → device ids (duplicates allowed: several shards on one device);
// a plain "HIP" / "CUDA" takes ICICLE_SNARK_DEVICES=<list> when set.  Returns 0 or an error code (text in last_error_text).
int parse_device_string(const char* device, std::vector<int>& ids);
int group_load(Groth16CacheManager* cm, const char* key, const uint8_t* zkey, size_t len, const std::vector<int>& devs);
int group_commitments(Groth16CacheManager* cm, DeviceGroup* g, const void* wtns, size_t wtns_len, uint8_t* out_points, Groth16Timings* tm);
const ZKeyCache* group_lead(const DeviceGroup* g);
void group_info(const DeviceGroup* g, Groth16CircuitInfo* info);
std::string group_describe(const DeviceGroup* g); // one line of JSON: shards, devices, transport, RCCL ranks (groth16_group_describe)
} // namespace prover
} // namespace isnark
