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
#include <climits>
#include <algorithm>
#include <fcntl.h>
#include <errno.h>
#include <sys/mman.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <unistd.h>

#include "prover_internal.h"
#include "../workers.h"

using namespace bn254;
using namespace isnark;
using namespace isnark::prover;

namespace {
// circuits with a domain (and witness) of up to this many elements start their witness MSMs right after the witness sort instead
// of behind the QAP front end
// (2^18 is 2-3 % better for the benchmark chain at 300-500 k constraints, 2^19 is 5 % better for the witness-light stand-in at 400 k)
constexpr uint32_t EARLY_MAX_DEFAULT = 1u << 19;
} // namespace

namespace isnark {
namespace prover {
// get_cache — src/cache.rs:242-256: (re)initialise the NTT domain of the entry's device for this key.  Sized 2·domain_size
// here (see file header); the reference sizes it from points_a.len() (a quirk).  Works on the calling thread's
// active device, which the caller has set to z->device_id.
int ensure_domain_for(Groth16CacheManager* cm, int device_id, uint32_t domain_size)
{
  static std::mutex dom_mu; // the prover threads of a device group come here concurrently (and may share a device)
  std::lock_guard<std::mutex> lk(dom_mu);
  auto it = cm->domain_n.find(device_id);
  if (it != cm->domain_n.end() && it->second == domain_size) {
    int lg = 0;
    if (ntt_domain_table(&lg) && (1u << lg) >= 2 * domain_size) return 0;
  }
  P_ICICLE(bn254_ntt_release_domain());
  bn254_scalar_t root;
  P_ICICLE(bn254_get_root_of_unity(2ull * domain_size, &root));
  NTTInitDomainConfig ic;
  memset(&ic, 0, sizeof ic);
  P_ICICLE(bn254_ntt_init_domain(&root, &ic));
  cm->domain_n[device_id] = domain_size;
  return 0;
}
int ensure_domain(Groth16CacheManager* cm, const ZKeyCache* z) { return ensure_domain_for(cm, z->device_id, z->domain_size); }

std::shared_ptr<ZKeyCache> find(Groth16CacheManager* cm, const char* key)
{
  std::lock_guard<std::mutex> lk(cm->map_mu);
  auto it = cm->cache.find(key ? key : "");
  if (it == cm->cache.end()) return nullptr;
  it->second->last_use = ++cm->clock;
  return it->second;
}
std::shared_ptr<DeviceGroup> find_group(Groth16CacheManager* cm, const char* key)
{
  std::lock_guard<std::mutex> lk(cm->map_mu);
  auto it = cm->groups.find(key ? key : "");
  return it == cm->groups.end() ? nullptr : it->second;
}

int set_active_device(int device_id)
{
  IcicleDevice dev;
  memset(&dev, 0, sizeof dev);
  strcpy(dev.type, "HIP");
  dev.id = device_id;
  P_ICICLE(icicle_set_device(&dev));
  return 0;
}

// Cache budget (groth16_cache_set_budget / ICICLE_SNARK_CACHE_BUDGET_MB): before a key of `need` bytes is built on `device`,
// the least recently used single-device entries of that device are evicted until used + need ≤ budget.  The fixed-base
// tables make an entry ≈ 10× the reference's (8.8 GB at 1.6 M constraints), so a CacheManager that holds several keys
// (src/cache.rs:110-114 never evicts) needs a bound.  Device groups are evicted only explicitly.  Caller holds cm->mu.
void evict_for_budget(Groth16CacheManager* cm, int device, uint64_t need)
{
  if (!cm->budget_bytes) return;
  for (;;) {
    std::shared_ptr<ZKeyCache> victim;
    {
      std::lock_guard<std::mutex> lk(cm->map_mu);
      uint64_t used = 0;
      auto lru = cm->cache.end();
      for (auto it = cm->cache.begin(); it != cm->cache.end(); ++it) {
        if (it->second->device_id != device) continue;
        used += it->second->device_bytes + it->second->tb.extra_bytes.load(std::memory_order_relaxed); // (+ what a narrower first table build takes beyond its estimate)
        if (lru == cm->cache.end() || it->second->last_use < lru->second->last_use) lru = it;
      }
      if (used + need <= cm->budget_bytes || lru == cm->cache.end()) return;
      victim = std::move(lru->second);
      cm->cache.erase(lru);
    }
    victim.reset(); // frees the device memory (outside the map lock)
  }
}

// bytes the budget of `device` has left once every cached key's device_bytes is counted (no budget: no limit) — what a first table
// build that wants more rows than the key was admitted with may take (cache.cpp: table_build_thread).  Caller holds cm->mu.
uint64_t budget_room(Groth16CacheManager* cm, int device)
{
  if (!cm->budget_bytes) return UINT64_MAX;
  std::lock_guard<std::mutex> lk(cm->map_mu);
  uint64_t used = 0;
  for (auto& kv : cm->cache)
    if (kv.second->device_id == device) used += kv.second->device_bytes + kv.second->tb.extra_bytes.load(std::memory_order_relaxed);
  return cm->budget_bytes > used ? cm->budget_bytes - used : 0;
}

} // namespace prover
} // namespace isnark

// ------------------------------------------------------------------------------------------------ C API
extern "C" {

__attribute__((visibility("default"))) const char* groth16_last_error(void) { return last_error_text(); }

extern "C" __attribute__((visibility("default"))) void groth16_cache_manager_prewarm(Groth16CacheManager* cm, int device_id);
__attribute__((visibility("default"))) Groth16CacheManager* groth16_cache_manager_new(void)
{
  Groth16CacheManager* cm = new Groth16CacheManager();
  if (const char* b = getenv("ICICLE_SNARK_CACHE_BUDGET_MB")) cm->budget_bytes = (uint64_t)atoll(b) << 20;
  // a device is already chosen (icicle_set_device before the manager, like src/lib.rs:25-31 before :44): what the first cache
  // load needs of it — six streams with their DMA queues, the pinned staging pool — is created now, on a helper thread
  const int dev = default_device_or_none();
  if (dev >= 0) groth16_cache_manager_prewarm(cm, dev);
  return cm;
}
// The same for a caller that knows its device before it has made it current (the REPL at start-up).  Returns at once; the
// first cache load of the manager waits for the helper thread.
__attribute__((visibility("default"))) void groth16_cache_manager_prewarm(Groth16CacheManager* cm, int device_id)
{
  if (!cm || device_id < 0) return;
  std::lock_guard<std::mutex> lk(cm->mu);
  if (cm->warm.joinable()) return;
  static const bool off = getenv("ICICLE_SNARK_PREWARM") && atoi(getenv("ICICLE_SNARK_PREWARM")) == 0;
  if (off) return;
  try {
    // six streams for the first key, two lanes for its cold upload: a lane created on demand costs the first cold prove of a
    // process 80 ms (profiles/r05_cold_path.txt)
    cm->warm = std::thread([device_id] {
      prewarm_device(device_id, 8);
      prewarm_modules(device_id);
    });
  } catch (...) {
  }
}
__attribute__((visibility("default"))) void groth16_cache_manager_free(Groth16CacheManager* cm) { delete cm; }
__attribute__((visibility("default"))) void groth16_cache_set_budget(Groth16CacheManager* cm, uint64_t bytes_per_device)
{
  if (!cm) return;
  std::lock_guard<std::mutex> lk(cm->mu);
  cm->budget_bytes = bytes_per_device;
}

__attribute__((visibility("default"))) int groth16_cache_contains(const Groth16CacheManager* cm, const char* key)
{
  Groth16CacheManager* m = const_cast<Groth16CacheManager*>(cm);
  return m && (find(m, key) || find_group(m, key)) ? 1 : 0;
}
__attribute__((visibility("default"))) void groth16_cache_evict(Groth16CacheManager* cm, const char* key)
{
  if (!cm) return;
  std::shared_ptr<ZKeyCache> victim; // destroyed outside the map lock; a prove in flight keeps its own reference
  std::shared_ptr<DeviceGroup> gvictim;
  {
    std::lock_guard<std::mutex> lk(cm->map_mu);
    auto it = cm->cache.find(key ? key : "");
    if (it != cm->cache.end()) {
      victim = std::move(it->second);
      cm->cache.erase(it);
    }
    auto ig = cm->groups.find(key ? key : "");
    if (ig != cm->groups.end()) {
      gvictim = std::move(ig->second);
      cm->groups.erase(ig);
    }
  }
}

// bytes a single-device entry of this zkey will hold, estimated from the container (for the budget): with tables ≈ 13 × the
// point sections, else the sections themselves; + the work buffers
static uint64_t estimate_entry_bytes(size_t zkey_len) { return (uint64_t)zkey_len * 11; }

__attribute__((visibility("default"))) int groth16_cache_load(Groth16CacheManager* cm, const char* key, const void* zkey, size_t zkey_len, int device_id, int shard_rank, int shard_count)
{
  if (!cm || !key || !zkey) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  if (find(cm, key) || find_group(cm, key)) return 0;
  if (cm->warm.joinable()) cm->warm.join(); // (what it creates is what the build below would create itself)
  evict_for_budget(cm, device_id, estimate_entry_bytes(zkey_len) / (uint64_t)(shard_count > 0 ? shard_count : 1));
  std::unique_ptr<ZKeyCache> z;
  // The NTT domain of the key (twiddle tables of 2·domain_size roots: 9–13 ms at 1.6 M constraints) is set up on a helper
  // thread WHILE the sections cross PCIe — the GPU has nothing else to do then — instead of inside the first prove.
  uint32_t dom_n = 0;
  {
    std::vector<Section> secs;
    const Section* s2 = nullptr;
    if (read_sections((const uint8_t*)zkey, zkey_len, "zkey", 2, secs) == 0 && unique_section(secs, 2, &s2) == 0 && s2->size >= 84) memcpy(&dom_n, s2->p + 80, 4);
    if (dom_n == 0 || (dom_n & (dom_n - 1)) || dom_n > (1u << 27)) dom_n = 0; // (a malformed header is build_cache's to report)
  }
  std::thread dom_th;
  if (dom_n && shard_count == 1) {
    try {
      dom_th = std::thread([cm, device_id, dom_n] {
        if (set_active_device(device_id) == 0) (void)ensure_domain_for(cm, device_id, dom_n); // a failure here is found again by the prove
      });
    } catch (...) {
    }
  }
  // the fixed-base tables of a single-device key are built behind its first proofs (cache.cpp: TableBuild)
  const int brc = build_cache((const uint8_t*)zkey, zkey_len, device_id, shard_rank, shard_count, z, /*defer_tables=*/true);
  if (dom_th.joinable()) dom_th.join();
  if (brc) return brc;
  std::shared_ptr<ZKeyCache> zp(z.release());
  {
    std::lock_guard<std::mutex> lm(cm->map_mu);
    zp->last_use = ++cm->clock;
    cm->cache[key] = zp;
  }
  zp->tb.narrow_room.store(budget_room(cm, device_id), std::memory_order_release);
  return 0;
}

__attribute__((visibility("default"))) int groth16_cache_load_file(Groth16CacheManager* cm, const char* key, const char* zkey_path, int device_id, int shard_rank, int shard_count)
{
  if (!cm || !key || !zkey_path) return fail(ERR_ARG, "null argument");
  if (groth16_cache_contains(cm, key)) return 0;
  MappedFile f;
  if (int rc = f.open_ro(zkey_path)) return rc;
  staged_copy_file_hint(f.data, f.len, f.fd); // sections 4-9 are pread() into the pinned staging buffers
  const int rc = groth16_cache_load(cm, key, f.data, f.len, device_id, shard_rank, shard_count);
  staged_copy_file_hint(nullptr, 0, -1);
  return rc;
}

// One key over a GROUP of devices (what the device string "HIP:0-7" of groth16_prove builds): shard r of n_devices lives on
// device_ids[r]; the same device may be named several times (several shards on one GPU: how a 1-GPU box tests the 8-way path).
__attribute__((visibility("default"))) int groth16_cache_load_devices(Groth16CacheManager* cm, const char* key, const void* zkey, size_t zkey_len, const int* device_ids, int n_devices)
{
  if (!cm || !key || !zkey || !device_ids || n_devices < 1) return fail(ERR_ARG, "bad argument");
  if (n_devices == 1) return groth16_cache_load(cm, key, zkey, zkey_len, device_ids[0], 0, 1);
  std::lock_guard<std::mutex> lk(cm->mu);
  if (find(cm, key) || find_group(cm, key)) return 0;
  return group_load(cm, key, (const uint8_t*)zkey, zkey_len, std::vector<int>(device_ids, device_ids + n_devices));
}

// "HIP:0-7" → {0,…,7}; returns the number of devices named (> cap: `ids` holds the first cap), or a negative error code
__attribute__((visibility("default"))) int groth16_parse_device(const char* device, int* ids, int cap)
{
  std::vector<int> v;
  if (int rc = parse_device_string(device, v)) return rc < 0 ? rc : -rc;
  for (int i = 0; i < (int)v.size() && i < cap; i++) ids[i] = v[i];
  return (int)v.size();
}

__attribute__((visibility("default"))) int groth16_last_timings(Groth16CacheManager* cm, const char* key, Groth16Timings* tm)
{
  if (!cm || !tm) return fail(ERR_ARG, "null argument");
  const std::shared_ptr<DeviceGroup> g = find_group(cm, key);
  const std::shared_ptr<ZKeyCache> zp = g ? nullptr : find(cm, key);
  if (!zp && !g) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  std::lock_guard<std::mutex> lk(cm->mu);
  *tm = g ? group_lead(g.get())->last_tm : zp->last_tm; // a group stores the slowest shard's phases in its lead entry
  return 0;
}

__attribute__((visibility("default"))) int groth16_cache_info(const Groth16CacheManager* cm, const char* key, Groth16CircuitInfo* info)
{
  if (!cm || !info) return fail(ERR_ARG, "null argument");
  Groth16CacheManager* m = const_cast<Groth16CacheManager*>(cm);
  if (const std::shared_ptr<DeviceGroup> g = find_group(m, key)) {
    group_info(g.get(), info);
    return 0;
  }
  const std::shared_ptr<ZKeyCache> zp = find(m, key);
  if (!zp) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  const ZKeyCache* z = zp.get();
  info->n_vars = z->n_vars;
  info->n_public = z->n_public;
  info->domain_size = z->domain_size;
  info->n_coef = z->n_coef;
  info->device_bytes = z->device_bytes;
  info->b_bases = z->B1.len();
  info->shards = 0;
  return 0;
}
// The same with the caller's struct size: fields beyond `info_size` are not written (a caller built against an older header
// keeps working when the struct grows)
__attribute__((visibility("default"))) int groth16_cache_info_sized(const Groth16CacheManager* cm, const char* key, void* info, size_t info_size)
{
  Groth16CircuitInfo full;
  memset(&full, 0, sizeof full);
  if (int rc = groth16_cache_info(cm, key, &full)) return rc;
  if (!info) return fail(ERR_ARG, "null argument");
  memcpy(info, &full, info_size < sizeof full ? info_size : sizeof full);
  return 0;
}

// Deferred tables of a single-device key: 1 = the key proves in its final layout (tables adopted, or none coming), 0 = the
// worker thread is still building them (proves meanwhile run the classic layout: same proofs, ≈ 20 % slower at 1.6 M
// constraints).  `wait` != 0 blocks until the build has ended and adopts the tables (bench.py, tests: a warm measurement
// starts from here).  Device groups build their tables at load: always 1.
__attribute__((visibility("default"))) int groth16_cache_tables_ready(Groth16CacheManager* cm, const char* key, int wait)
{
  if (!cm) return fail(ERR_ARG, "null argument");
  if (find_group(cm, key)) return 1;
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  if (!zp) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  if (zp->tb.state.load(std::memory_order_acquire) == 0) return 1;
  if (!wait && zp->tb.state.load(std::memory_order_acquire) == 1) return 0;
  if (wait) {
    zp->tb.go.store(true, std::memory_order_release); // somebody waits for the tables: no point in the build waiting for a first prove
    // wait for the build OUTSIDE the manager's mutex (0.26–1 s: every prove of every other key would queue behind it otherwise;
    // round-5 advisor) — the mutex is only taken for the pointer swap below
    while (zp->tb.state.load(std::memory_order_acquire) == 1) std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  std::lock_guard<std::mutex> lk(cm->mu);
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (int rc = set_active_device(zp->device_id)) return rc > 0 ? fail(ERR_ARG, "device %d of the key cannot be made active (icicle error %d)", zp->device_id, rc) : rc; // (1 means "ready" here)
  const int r = adopt_tables(zp.get(), wait != 0);
  if (prev >= 0) (void)set_active_device(prev);
  return r;
}

// What a device-group key runs on, as one line of JSON: {"shards", "devices", "distinct_devices", "transport": "pull" | "memcpy" |
// "rccl", "peer_access", "rccl_ranks" (0 unless the rccl transport moves the exchanges), "distributed_front_end", …}.  A
// single-device key answers {"shards": 0, "devices": [id]}.  Returns 0, or the size needed (incl. NUL) when `cap` is too small.
__attribute__((visibility("default"))) int groth16_group_describe(const Groth16CacheManager* cm, const char* key, char* out, size_t cap)
{
  if (!cm || !out) return fail(ERR_ARG, "null argument");
  Groth16CacheManager* m = const_cast<Groth16CacheManager*>(cm);
  std::string text;
  if (const std::shared_ptr<DeviceGroup> g = find_group(m, key)) text = group_describe(g.get());
  else if (const std::shared_ptr<ZKeyCache> zp = find(m, key))
    text = "{\"shards\": 0, \"devices\": [" + std::to_string(zp->device_id) + "], \"distinct_devices\": 1, \"transport\": \"none\", \"rccl_ranks\": 0}";
  else
    return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  if (text.size() + 1 > cap) return (int)text.size() + 1;
  memcpy(out, text.c_str(), text.size() + 1);
  return 0;
}

__attribute__((visibility("default"))) int groth16_commitments(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, uint8_t out_points[GROTH16_COMMITMENTS_BYTES], Groth16Timings* tm)
{
  if (!cm || !out_points) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  if (const std::shared_ptr<DeviceGroup> g = find_group(cm, key)) return group_commitments(cm, g.get(), wtns, wtns_len, out_points, tm);
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  if (!zp) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  return shard_commitments(cm, zp.get(), wtns, wtns_len, out_points, tm, nullptr);
}

} // extern "C"

namespace isnark {
namespace prover {
int shard_commitments(Groth16CacheManager* cm, ZKeyCache* z, const void* wtns, size_t wtns_len, uint8_t* out_points, Groth16Timings* tm, EarlyTerms* et)
{
  if (!wtns && !z->witness_resident) return fail(ERR_ARG, "no witness given and none resident on the device");
  const auto t0 = std::chrono::steady_clock::now();
  static const bool trace_host = getenv("ICICLE_SNARK_TRACE_HOST") != nullptr;
  auto mark = [&](const char* what) {
    if (trace_host) fprintf(stderr, "[host] %-12s %8.1f us\n", what, ms_since(t0) * 1e3);
  };
  if (int rc = set_active_device(z->device_id)) return rc;
  if (int rc = ensure_domain(cm, z)) return rc;
  mark("domain");
  // deferred fixed-base tables (cache.cpp): complete → this prove is the first to use them; still building → classic layout
  (void)adopt_tables(z, false);
  // the key follows its witnesses: a digit width at least two bits off the one the last witness called for (one bit when the key still
  // has its dense width) → the four witness
  // tables are re-built with that width, at most once every eight proves — by a worker thread BESIDE the proves of the key (round 5;
  // rounds 3–4 re-built them here, 0.1–0.3 s inside a prove), which go on with the tables they have until the new ones are complete
  // and adopt_tables above swaps them in (all four and geom_w together).  The worker is started at the END of the prove that counted
  // the digits (follow_witness below); ICICLE_SNARK_SYNC_REBUILD=1 re-builds here, inside the next prove, as rounds 3–4 did.
  auto follow_witness = [&](bool sync) -> int {
    if (!(z->geom_w.tab && z->witness_entries && z->proves_since_rebuild >= 1 && z->tb.state.load(std::memory_order_acquire) == 0)) return 0;
    const int c_t = witness_digit_target(z, z->witness_entries);
    // (from the dense width one bit narrower is taken at once — below ≈ 28 entries per bucket it pays, cache.cpp: witness_table_geometry —;
    //  every later move needs two bits and eight proves: no flapping)
    const bool from_default = z->geom_w.c == z->geom_w_default_c;
    if (!(((from_default && c_t < z->geom_w.c) || c_t <= z->geom_w.c - 2 || c_t >= z->geom_w.c + 2) && (from_default || z->proves_since_rebuild >= 8))) return 0;
    if (sync) {
      if (int rc = rebuild_witness_tables(z, c_t)) return rc;
    } else
      start_witness_rebuild(z, c_t);
    z->proves_since_rebuild = 0;
    return 0;
  };
  const bool sync_rebuild = getenv("ICICLE_SNARK_SYNC_REBUILD") && atoi(getenv("ICICLE_SNARK_SYNC_REBUILD")) != 0;
  if (sync_rebuild)
    if (int rc = follow_witness(true)) return rc;
  const uint32_t n = z->domain_size, nv = z->n_vars, npub = z->n_public;
  hipStream_t g1 = z->s_g1, g2 = z->s_g2, g3 = z->s_g3, gq = z->s_qap;
  double h2d_host_ms = 0;
  SortPlan plan_w, plan_head, plan_h; // witness (all of it, or its tail when the head is sorted apart), head of the witness, H scalars
  // bucket arrays of the four witness MSMs (A, B1, B2, C): owned here because a head and a tail accumulation share them
  WsScoped<uint8_t> bk[4];
  // Declared after the plans, so it runs before their destructors, and before the first enqueue of this call: on an error
  // return the kernels already enqueued may still read the plans' workspace (which ~SortPlan hands back to the arena) or
  // the caller's pinned witness buffer — drain the six streams first.
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
  const uint32_t wlo = z->A.lo, wlen = z->A.len(), skip = npub + 1;
  const uint32_t early_max = EARLY_MAX_DEFAULT;
  // Witness MSMs of small circuits (domain up to 2^19) leave the GPU far from full: they start right
  // after the witness sort instead of waiting for the QAP (200 k constraints: 3.71 → 3.51 ms, 400 k: 6.69 → 6.24 ms; the QAP
  // itself slows down — 1.1 → 3.9 ms at 400 k — which is why the large ones are held back: 800 k: 9.87 → 10.27 ms).
  // A shard of an in-process device group whose point range is its own witness slice (cache.cpp: slice_aligned) starts its witness
  // sort behind its own PCIe upload and its accumulations behind that sort: the all-gather of the witness, the distributed front
  // end and its two all-to-alls — xGMI round trips during which the GPU would otherwise wait — then run beside them; only H needs
  // the front end.  (On ONE GPU with every shard aliased to it this is neutral: there is no exchange latency to hide.)
  const bool own_slice_first = z->in_group && z->slice_aligned && z->own_slice_event_set && wtns == nullptr;
  const bool early = wlen <= early_max && (n <= early_max || own_slice_first); // (rank-per-GPU shards keep the full-size inverse transform: neutral there)
  MsmProfile* prof[5]; // A, B1, B2, C, H — this entry's own slots (published to the device's ring at the end by the lead shard)
  for (int k = 0; k < 5; k++) {
    prof[k] = &z->prof[k];
    msm_profile_own_init(prof[k]);
  }
  msm_profile_own_init(&z->prof[5]); // digit sort of the witness head (roofline.scatter of bench.py adds it to the tail's)
  z->prof[5].L = 0;
  z->prof[5].valid = true; // (published either way so that the ring keeps its order; L = 0 says "no head in this prove")
  // (tab > 1 = table mode with exactly this digit width: a key adapted to its witnesses, cache.cpp)
  const bool adapted_w = z->geom_w.tab && z->geom_w.c != z->geom_w_default_c;
  const uint32_t skip_below = skip > wlo ? skip - wlo : 0; // C ignores witness[0..=n_public]
  const Shard* sh4[4] = {&z->A, &z->B1, &z->B2, &z->C};
  hipStream_t st4[4] = {g1, z->s_g4, g2, z->s_g5}; // A, B1, B2, C: separate streams, so that one MSM's latency-bound reduction overlaps another's accumulation
  // accumulation of witness[first … first + pl.L) into the bucket array of MSM k (0 A, 1 B1, 2 B2, 3 C) on stream st: the sort
  // entries index scalars relative to `first`, so the table pointer moves with it (C's bases start at wire n_public + 1)
  // The large-bucket kernels of an MSM on a SIDE stream beside its accumulation (msm_plan.h: LargeSide): B2's on the front end's
  // stream and H's on the H-sort stream — both idle by then.  ICICLE_SNARK_LARGE_SIDE: bit k = MSM k (A, B1, B2, C, H); default B2 + H.
  static const int large_side_mask = getenv("ICICLE_SNARK_LARGE_SIDE") ? atoi(getenv("ICICLE_SNARK_LARGE_SIDE")) : 0x14;
  auto large_side = [&](int k, hipStream_t side_stream) {
    LargeSide ls;
    if ((large_side_mask >> k) & 1) {
      ls.stream = side_stream;
      ls.fork = z->ev_lfork[k];
      ls.join = z->ev_ljoin[k];
    }
    return ls;
  };
  auto accumulate = [&](int k, const SortPlan& pl, uint32_t first, bool into, hipStream_t st, MsmProfile* p, bool resident = false, const LargeSide* side = nullptr) -> int {
    const size_t esz = k == 2 ? 128 : 64;
    uint32_t sb = 0;
    size_t base_off = first;
    if (k == 3) {
      if (first >= skip_below) base_off = first - skip_below;
      else {
        base_off = 0;
        sb = skip_below - first;
      }
    }
    const void* pts = (const uint8_t*)sh4[k]->d_points + base_off * esz;
    if (k == 2) P_ICICLE(msm_g2_accumulate(&pl, pts, 2, sb, st, bk[k].p, into, p, sh4[k]->len(), resident, side));
    else P_ICICLE(msm_g1_accumulate(&pl, pts, 2, sb, st, bk[k].p, into, p, sh4[k]->len(), resident, side));
    return 0;
  };

  // ---- the witness arrives.  HEAD / TAIL: over PCIe the witness takes 1.4 ms (host buffer) to 2.7 ms (file) at 1.6 M constraints
  // and nothing of construct_r1cs can start before all of it is there (the spmv reads arbitrary wires) — but the witness MSMs are
  // sums over wires: the first `head` wires are sorted and accumulated into the four bucket arrays while the rest is still on
  // its way, and the tail's accumulation continues those buckets after the front end (msm_plan.h: `into`).  The GPU, idle during
  // the upload before, takes 8–20 % of the witness accumulations (what the upload of the machine affords) off the critical path.
  uint32_t head = 0;
  const uint32_t head_unit = (uint32_t)(STAGED_CHUNK_BYTES / 32); // the head is a whole number of upload chunks
  bool pinned_src = false, head_forced = false;
  if (wtns) {
    Wtns w;
    if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
    // src/proof_helper.rs:253-262
    if (!Fr::eq(z->r, w.q)) return fail(ERR_FORMAT, "Curve of the witness does not match the curve of the proving key");
    if (w.n_witness != z->n_vars) return fail(ERR_FORMAT, "Invalid witness length. Circuit: %u, witness: %u", z->n_vars, w.n_witness);
    // a new witness invalidates whatever the distributed stages left behind for the previous one
    z->witness_resident = false;
    z->dist_ready = z->dist_stage2_done = false;
    z->witness_event_set = false;
    // (ICICLE_SNARK_HEAD_PCT: the head's share in percent, 0 = off; ICICLE_SNARK_HEAD_MIN: smallest witness that is split —
    //  the tests set it to 0 so that circuits the oracle proves in seconds take the path)
    static const int head_pct_env = getenv("ICICLE_SNARK_HEAD_PCT") ? atoi(getenv("ICICLE_SNARK_HEAD_PCT")) : -1;
    static const long head_min_env = getenv("ICICLE_SNARK_HEAD_MIN") ? atol(getenv("ICICLE_SNARK_HEAD_MIN")) : -1;
    const uint32_t head_min = head_min_env >= 0 ? (uint32_t)head_min_env : (1u << 19);
    head_forced = head_pct_env >= 0;
    // Not for a key adapted to light witnesses (cache.cpp: mostly 0 / 1 wires, a fifth of the digits of a dense witness): its
    // accumulations are short and the second sort, the second round of large-bucket kernels and the transforms slowed by the
    // head's last workgroups cost more than the head takes off the MSM phase (stand-ins of BASELINE configs 4 / 5: 5.6 → 7.0 ms
    // and 8.6 → 9.3 ms with a head; benchmark/1600k: 16.8 → 16.3 ms, 3200k: 30.7 → 29.4 ms).
    const bool head_ok = z->geom_w.tab && z->shard_count == 1 && wlo == 0 && wlen == nv && nv >= head_min;
    if (head_ok && head_pct_env >= 0 && (!early || head_min_env >= 0)) {
      head = (uint32_t)((double)nv * (head_pct_env / 100.0) / head_unit + 0.5) * head_unit; // forced share (tests, sweeps)
    } else if (head_ok && !early && !adapted_w) {
      if (z->head_units < 0) z->head_units = (int)((double)nv * 0.10 / head_unit + 0.5); // first prove of the key: a tenth
      head = (uint32_t)z->head_units * head_unit;
    }
    if (head < head_unit || head >= nv || nv - head < head_unit) head = 0;
    if (z->feed) head = 0;
    const auto tu = std::chrono::steady_clock::now();
    pinned_src = !z->feed && is_pinned_host(w.values, z->device_id);
    P_HIP(hipEventRecord(z->ev[0], gq));
    if (z->feed) {
      // cold pipeline (prover_internal.h: ColdFeed): the uploader task that is still sending the key's sections has the witness
      // as its second stage — wait until that stage has been POSTED (its event recorded), then order the front end behind it
      if (int rc = z->feed->wait(ColdFeed::WITNESS)) return fail(rc, "%s", z->feed->err.c_str());
      P_HIP(hipStreamWaitEvent(gq, z->feed->ev[ColdFeed::WITNESS], 0));
      mark("feed: witness");
    } else if (head) {
      // the head's kernels go to g2; the staging lanes of the upload are three of the streams with nothing to do before the whole
      // witness is there (QAP, H sort, C's own) — no extra stream, no extra hardware queue
      const hipStream_t lanes[3] = {gq, g3, z->s_g5}; // (two to five lanes measure the same: profiles/r04_upload_lanes.txt)
      StagedProgress prog;
      prog.head_bytes = (size_t)head * 32;
      int up_rc = 0, wait_rc = 0; // up_rc and up_err belong to the uploader task until it has been waited for
      std::string up_err;
      HostTask uploader; // (a pooled worker, workers.h; waited for on every path below)
      bool uploader_started = false;
      const void* hint_base;
      size_t hint_len;
      int hint_fd;
      staged_copy_file_hint_get(&hint_base, &hint_len, &hint_fd);
      if (pinned_src) {
        // the caller's buffer is pinned: two DMAs straight from it on the QAP stream, an event between them
        P_HIP(hipMemcpyAsync(z->d_witness, w.values, (size_t)head * 32, hipMemcpyHostToDevice, gq));
        P_HIP(hipEventRecord(z->ev_head_in, gq));
        P_HIP(hipMemcpyAsync(z->d_witness + head, (const uint8_t*)w.values + (size_t)head * 32, (size_t)(nv - head) * 32, hipMemcpyHostToDevice, gq));
        P_HIP(hipStreamWaitEvent(g2, z->ev_head_in, 0));
      } else {
        uploader.fn = [&] {
          staged_copy_file_hint(hint_base, hint_len, hint_fd);
          up_rc = staged_upload(z->device_id, {{z->d_witness, (const uint8_t*)w.values, (size_t)nv * 32}}, lanes, 3, &prog);
          if (up_rc) up_err = last_error_text(); // (the text lives in this worker's thread-local slot: hand it to the caller)
          staged_copy_file_hint(nullptr, 0, -1);
          prog.done.store(true, std::memory_order_release);
          prog.notify();
        };
        // no worker to be had (thread limit of the container): the upload runs here, the head follows it instead of overlapping it
        WorkerPool::get().run_or_inline(&uploader);
        uploader_started = true;
        prog.wait_head(); // blocks (condition variable) until every lane has recorded the event behind its last chunk of the head
        mark("head in");
        const int tot = prog.lanes_total.load(std::memory_order_acquire);
        for (int t = 0; t < tot; t++)
          if (prog.ev[t] && hipStreamWaitEvent(g2, prog.ev[t], 0) != hipSuccess) wait_rc = fail((int)ICICLE_UNKNOWN_ERROR, "hipStreamWaitEvent");
      }
      // head: digit sort, then the four accumulations (zero-initialising their bucket arrays) — ONE chain on g2, each kernel
      // launched with no more workgroups than the GPU holds (`resident`).  Four concurrent accumulations with ordinary
      // grids were measured first: their queued workgroups kept the hardware pipes busy dispatching, the barrier packets
      // behind the staging DMAs (the events the upload workers wait for before they re-use a pinned buffer) were not
      // processed until the kernels ended, and the upload stalled for their whole length (70 % of the bytes in 1.8 ms, the
      // rest 1.1 ms late: profiles/r04_head_concurrent_timeline.txt).
      int hrc = 0;
      auto enqueue_head = [&]() -> int {
        (void)hipEventRecord(z->ev_t_head_start, g2); // (timing, with ev_t_head_end and ev_t_witness: steers head_frac)
        MsmProfile* ph = &z->prof[5];
        (void)hipEventRecord(ph->ev[0], g2);
        P_ICICLE(msm_sort_run(z->d_witness, head, 0, 0, 0, g2, &plan_head, z->geom_w.c, 0, 1, adapted_w ? (uint64_t)((double)z->witness_entries * head / nv) + 1 : 0));
        for (int e = 1; e < 5; e++) (void)hipEventRecord(ph->ev[e], g2);
        ph->has_sort_end = ph->valid = true;
        ph->L = head; ph->nbuckets = plan_head.nbuckets; ph->c = plan_head.g.c; ph->W = plan_head.g.W; ph->is_g2 = 0;
        // entry count of the head's sort (offset + count of its last bucket) for the key's digit-width rule — on g2, in stream
        // order behind the sort that writes them (round-4 advisor: on the tail's stream nothing ordered the copies behind it)
        if (plan_head.nbuckets) {
          P_HIP(hipMemcpyAsync(&z->h_stats[2], plan_head.offsets + plan_head.nbuckets - 1, 4, hipMemcpyDeviceToHost, g2));
          P_HIP(hipMemcpyAsync(&z->h_stats[3], plan_head.counts + plan_head.nbuckets - 1, 4, hipMemcpyDeviceToHost, g2));
        }
        if (plan_head.g.tab != z->geom_w.tab || plan_head.g.c != z->geom_w.c || plan_head.nbuckets != z->geom_w.NB)
          return fail((int)ICICLE_UNKNOWN_ERROR, "window geometry of the cached tables does not match the sort of the witness head");
        for (int k = 0; k < 4; k++) P_HIP(bk[k].alloc(msm_bucket_bytes(&plan_head, k == 2), st4[k]));
        for (int k : {2, 0, 1, 3})
          if (int rc = accumulate(k, plan_head, 0, false, g2, nullptr, true)) return rc;
        P_HIP(hipEventRecord(z->ev_head_done, g2));
        (void)hipEventRecord(z->ev_t_head_end, g2); // (timing: where the head's chain ends relative to the upload)
        return 0;
      };
      if (!wait_rc) hrc = enqueue_head();
      mark("head enq");
      if (uploader_started && uploader.queued) WorkerPool::wait(&uploader);
      mark("upload");
      if (up_rc) return fail(up_rc, "%s", up_err.c_str());
      if (wait_rc) return wait_rc;
      if (hrc) return hrc;
      if (pinned_src) P_HIP(hipEventRecord(z->ev_witness, gq));
    } else if (pinned_src) {
      // the caller's buffer is pinned (hipHostMalloc / hipHostRegister) and mapped for this device: one DMA straight from it,
      // in stream order with everything that waits for ev_witness — no staging copy, no host wait (51 MB: 0.9 instead of 1.35 ms)
      P_HIP(hipMemcpyAsync(z->d_witness, w.values, (size_t)nv * 32, hipMemcpyHostToDevice, gq));
      P_HIP(hipEventRecord(z->ev_witness, gq));
    } else {
      // witness → device through the parallel pinned-staging uploader of the cold path (three workers on prover streams, 2 MB
      // chunks): a single memcpy into one pinned buffer + one DMA took 4 ms for the 51 MB of benchmark/1600k.
      // three lanes: 51 MB in 1.4 ms, stable; with six, one upload in four stalled for ~15 ms on the GPU box (host threads
      // of this call, the MSM tails and the runtime's own compete for the container's CPU quota)
      const hipStream_t lanes[3] = {gq, g2, g3};
      if (int rc = staged_upload(z->device_id, {{z->d_witness, (const uint8_t*)w.values, (size_t)nv * 32}}, lanes, 3)) return rc;
    }
    h2d_host_ms = ms_since(tu);
    mark("witness in");
    if (!pinned_src) P_HIP(hipEventRecord(z->ev_witness, gq)); // (the staged upload has returned: every byte is there)
    z->witness_event_set = true;
  } else
    P_HIP(hipEventRecord(z->ev[0], gq));
  z->witness_resident = true;
  // resident witness: nothing to wait for.  Group prove: ev_witness was recorded behind the witness all-gather on the exchange's
  // stream (multi.cpp) and witness_event_set says so
  if (!z->witness_event_set) P_HIP(hipEventRecord(z->ev_witness, gq));
  z->witness_event_set = false;
  (void)hipEventRecord(z->ev_t_witness, gq);

  // ---- ONE digit sort of the witness range (shared by A, B1, B2, C) — of its tail when the head was sorted above — on g2
  // without a head (the G2 bucket stages follow it there) and on g3 with one (g2 still carries B2's head accumulation).
  // Enqueued BEHIND the front end for the large circuits (round 5): the front end is their critical chain and its first kernel
  // used to sit behind the nine launches of the sort on the host, ≈ 50 µs after the last byte of the witness had landed; the
  // small circuits and the shards that start from their own slice begin their accumulations right behind the sort: sort first.
  hipStream_t gs = head ? g3 : g2;
  MsmProfile* psort = prof[2]; // the witness sort is timed with the profile of the G2 MSM
  auto enqueue_witness_sort = [&]() -> int {
    P_HIP(hipStreamWaitEvent(gs, own_slice_first ? z->ev_own_slice : z->ev_witness, 0));
    (void)hipEventRecord(psort->ev[0], gs);
    {
      const uint32_t tail_len = wlen - head;
      const uint64_t hint = adapted_w ? (uint64_t)((double)z->witness_entries * tail_len / wlen) + 1 : 0;
      P_ICICLE(msm_sort_run(z->d_witness + wlo + head, tail_len, 0, 0, 0, gs, &plan_w, z->geom_w.tab ? z->geom_w.c : 0, 0, 1, hint)); // (table mode: the KEY's digit width, whatever msm_geometry's rule says for this length)
    }
    if (plan_w.g.tab != z->geom_w.tab || plan_w.g.c != z->geom_w.c || (head && plan_w.nbuckets != plan_head.nbuckets))
      return fail((int)ICICLE_UNKNOWN_ERROR, "window geometry of the cached tables does not match the witness sort");
    (void)hipEventRecord(psort->ev[4], gs); // end of the witness digit sort (roofline.scatter)
    psort->has_sort_end = true;
    P_HIP(hipEventRecord(z->ev_sort, gs));
    // entry counts of the sorts (offset + count of the last bucket), read at the end of the prove: they steer the digit width of
    // the key's witness tables (cache.cpp: rebuild_witness_tables)
    z->h_stats[0] = z->h_stats[1] = 0;
    if (!head) z->h_stats[2] = z->h_stats[3] = 0; // (with a head: written by the copies behind the head's sort on g2)
    if (plan_w.nbuckets) {
      P_HIP(hipMemcpyAsync(&z->h_stats[0], plan_w.offsets + plan_w.nbuckets - 1, 4, hipMemcpyDeviceToHost, gs));
      P_HIP(hipMemcpyAsync(&z->h_stats[1], plan_w.counts + plan_w.nbuckets - 1, 4, hipMemcpyDeviceToHost, gs));
    }
    mark("wsort");
    return 0;
  };
  if (early)
    if (int rc = enqueue_witness_sort()) return rc;

  // ---- stream gq: construct_r1cs (src/proof_helper.rs:31-170) on the device
  P_HIP(hipStreamWaitEvent(gq, z->ev_witness, 0));
  P_HIP(hipEventRecord(z->ev[1], gq));
  // the distributed stages left this rank's Z rows in d_fold — honoured only for the witness they were computed from (no new
  // witness in this call) and only once the caller has confirmed that exchange 2 delivered (groth16_dist_exchange_done)
  const bool dist_ready = !wtns && z->dist_ready && z->H.stride > 1;
  z->dist_ready = z->dist_stage2_done = false;
  if (z->feed) { // cold pipeline: the coefficient records have landed and the CSR has been built from them
    if (int rc = z->feed->wait(ColdFeed::COEF)) return fail(rc, "%s", z->feed->err.c_str());
    P_HIP(hipStreamWaitEvent(gq, z->feed->ev[ColdFeed::COEF], 0));
  }
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
  if (z->H.stride > 1) {
    // strided H shard: coset keys, the fold over the shard count and the twist in one pass, then a size-n/G transform
    const uint32_t m = z->H.len();
    if (!dist_ready) {
      P_ICICLE(bn254_ntt((const bn254_scalar_t*)z->d_vec, (int)n, kInverse, &nc, (bn254_scalar_t*)z->d_vec)); // :116
      P_HIP(qap_coset_fold3(z->d_vec, tw, (1u << dom_log) / (2 * n), n, z->H.stride, z->H.first, z->d_fold, gq));
    }
    if (ntt_fusable(m)) {
      NttFuse f;
      f.fused_out = z->d_fold + m;
      P_ICICLE(ntt_fused(z->d_fold, m, 3, false, gq, f));
    } else {
      P_ICICLE(bn254_ntt((const bn254_scalar_t*)z->d_fold, (int)m, kForward, &nc, (bn254_scalar_t*)z->d_fold));
      P_HIP(qap_final(z->d_fold, m, gq));
    }
    d_hscalars = z->d_fold + m;
  } else if (ntt_fusable(n)) {
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

  if (!early)
    if (int rc = enqueue_witness_sort()) return rc;
  auto fill = [](MsmProfile* p, const SortPlan& pl, int g2flag) {
    p->L = pl.L; p->nbuckets = pl.nbuckets; p->c = pl.g.c; p->W = pl.g.W; p->is_g2 = g2flag;
  };
  uint8_t* DP = z->d_partials;


  // ---- groth16_commitments — src/proof_helper.rs:198-206.  A, B1, B2, C share the witness sort and run on four streams.
  // Held back until the QAP front end is done (large circuits): the accumulations fill every CU with milliseconds-long
  // workgroups, and the NTT passes measured 8× slower when they had to wait for those to retire (rocprof: 2.9 ms vs 0.35 ms per pass).
  if (!head)
    for (int k = 0; k < 4; k++) P_HIP(bk[k].alloc(msm_bucket_bytes(&plan_w, k == 2), st4[k]));
  const int slot4[4] = {0, 1, 3, 2}; // ticket slots of the plan: A 0, B1 1, B2 3, C 2
  auto bucket_stages = [&](int k) -> int {
    MsmProfile* p = prof[k];
    hipStream_t st = st4[k];
    fill(p, plan_w, k == 2);
    P_HIP(hipStreamWaitEvent(st, z->ev_sort, 0));
    if (!early) P_HIP(hipStreamWaitEvent(st, z->ev[2], 0));
    if (head && k != 2) P_HIP(hipStreamWaitEvent(st, z->ev_head_done, 0)); // the heads were accumulated on g2
    if (z->feed) { // cold pipeline: this MSM's bases have landed and are in the bucket kernels' encoding
      if (int rc = z->feed->wait(ColdFeed::SEC_A + k)) return fail(rc, "%s", z->feed->err.c_str());
      P_HIP(hipStreamWaitEvent(st, z->feed->ev[ColdFeed::SEC_A + k], 0));
    }
    if (p != psort) (void)hipEventRecord(p->ev[0], st);
    // side streams idle in the MSM phase: the front end's for B2, A and B1; H-sort's (behind H's sort) for C
    const LargeSide ls = large_side(k, k == 3 ? g3 : gq);
    if (int rc = accumulate(k, plan_w, wlo ? 0 : head, head != 0, st, p, false, ls.stream ? &ls : nullptr)) return rc;
    P_ICICLE(k == 2 ? msm_g2_reduce(&plan_w, st, bk[k].p, DP + k * PARTIALS_STRIDE, slot4[k]) : msm_g1_reduce(&plan_w, st, bk[k].p, DP + k * PARTIALS_STRIDE, slot4[k]));
    (void)hipEventRecord(p->ev[3], st);
    p->valid = true;
    return 0;
  };
  // ---- stream g3: digit sort of the H scalars (atomics / memory bound) overlaps the ALU-bound A, B1, C stages
  auto enqueue_h_sort = [&]() -> int {
    P_HIP(hipStreamWaitEvent(g3, z->ev[2], 0));
    (void)hipEventRecord(prof[4]->ev[0], g3);
    // (`crowded`: H's sort runs beside the four witness accumulations of a large circuit)
    P_ICICLE(msm_sort_run(d_hscalars, z->H.len(), 0, 0, 0, g3, &plan_h, z->geom_h.tab ? z->geom_h.c : 0, 0, 1, 0, /*crowded=*/!early));
    if (plan_h.g.tab != z->geom_h.tab || plan_h.g.c != z->geom_h.c) return fail((int)ICICLE_UNKNOWN_ERROR, "window geometry of the cached tables does not match the H sort");
    (void)hipEventRecord(prof[4]->ev[4], g3);
    prof[4]->has_sort_end = true;
    P_HIP(hipEventRecord(z->ev_sort_h, g3));
    mark("hsort");
    return 0;
  };
  // (cold pipeline: the host is about to wait for B2's section — everything that needs no section is enqueued first)
  if (z->feed)
    if (int rc = enqueue_h_sort()) return rc;
  if (int rc = bucket_stages(2)) return rc; // commitment_b (G2) — src/proof_helper.rs:206: the longest chain first
  mark("g2");
  if (!z->feed)
    if (int rc = enqueue_h_sort()) return rc;

  for (int k : {0, 1, 3})
    if (int rc = bucket_stages(k)) return rc;
  mark("abc");

  // H: behind one of the witness MSMs for the large circuits (measured at 1.6 M constraints: five concurrent accumulations
  // are slower than four followed by one, 17.7 vs 17.4 ms) — behind B1, whose accumulation is the first of the three G1
  // ones to start and to finish (behind A: +0.1 ms, behind C: +0.4 ms); on g3 right behind its own sort for the small ones and for
  // multi-GPU shards, where the GPU is far from full and only the length of the chains counts (200 k: 4.3 → 3.9 ms).  Letting H
  // wait only for B1's ACCUMULATION kernel instead of B1's whole chain fills a ≈ 1 ms gap in the timeline and still makes the
  // prove slower (16.35–16.48 against 16.0–16.2 ms; HISTORY.md §4 lists this and the other schedules that were measured).
  const int h_behind = 1; // B1
  const bool h_chain = z->H.len() > (1u << 19);
  hipStream_t gh = h_chain ? st4[h_behind] : g3;
  if (h_chain) P_HIP(hipStreamWaitEvent(gh, z->ev_sort_h, 0));
  fill(prof[4], plan_h, 0);
  if (z->feed) {
    if (int rc = z->feed->wait(ColdFeed::SEC_H)) return fail(rc, "%s", z->feed->err.c_str());
    P_HIP(hipStreamWaitEvent(gh, z->feed->ev[ColdFeed::SEC_H], 0));
  }
  {
    const LargeSide ls = large_side(4, g3);
    P_ICICLE(msm_g1_partials(&plan_h, z->H.d_points, 2, 0, gh, DP + 4 * PARTIALS_STRIDE, prof[4], z->H.len(), 0, ls.stream && gh != g3 ? &ls : nullptr));
  }
  (void)hipEventRecord(prof[4]->ev[3], gh);
  prof[4]->valid = true;
  mark("h");
  // Each MSM's partial sums go to pinned memory on ITS OWN stream as soon as its reduction is done, and a host
  // thread per MSM waits for that copy and runs the Horner tail — the tails of the early finishers (B2, A, B1, C)
  // overlap the GPU work still in flight; only the last one (H) is exposed.
  uint32_t Ww = 0, bw1 = 0, Wb = 0, bw2 = 0, Wh = 0, bh = 0;
  const size_t by1 = msm_partials_bytes(&plan_w, false, &Ww, &bw1), by2 = msm_partials_bytes(&plan_w, true, &Wb, &bw2), byh = msm_partials_bytes(&plan_h, false, &Wh, &bh);
  const size_t sizes[5] = {by1, by1, by2, by1, byh};
  hipStream_t st5[5] = {st4[0], st4[1], st4[2], st4[3], gh};
  if (h_chain) {
    // the copy of the MSM in front of H must not wait for H (same stream): its partials were complete at its ev[3], copy them on g3 instead
    P_HIP(hipStreamWaitEvent(g3, prof[h_behind]->ev[3], 0));
    st5[h_behind] = g3;
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
    const int cw = plan_w.g.c, ch = plan_h.g.c;
    hipEvent_t* evd = z->ev_done;
    const int dev = z->device_id;
    const MsmGeom gw = plan_w.g, gh = plan_h.g;
    auto g1tail = [&](int k, uint32_t W, uint32_t bpw, int c, size_t off) {
      (void)hipSetDevice(dev);
      (void)hipEventSynchronize(evd[k]);
      const MsmGeom& gg = k == 4 ? gh : gw;
      if (gg.tab) msm_g1_host_tail_tab(HP + k * PARTIALS_STRIDE, W, (bn254_projective_t*)(out_points + off));
      else msm_g1_host_tail(HP + k * PARTIALS_STRIDE, W, 1, c, gg.wide, (bn254_projective_t*)(out_points + off));
      if (et && k < 2) {
        while (!et->bl_ready.load(std::memory_order_acquire)) std::this_thread::yield();
        bn254_projective_t p;
        memcpy(&p, out_points + off, sizeof p);
        bn254_ecadd(&p, (const bn254_projective_t*)(k == 0 ? &z->vk_alpha_1 : &z->vk_beta_1), &p);
        bn254_ecadd(&p, k == 0 ? &et->bl->d1r : &et->bl->d1s, &p);
        bn254_mul_scalar(&p, k == 0 ? &et->bl->s : &et->bl->r, k == 0 ? &et->ta : &et->tb);
        et->done.fetch_add(1, std::memory_order_release);
        if (k == 0) early_pi_a(z, *et->bl, &p, et); // p = pi_a
      }
    };
    // the four other tails on pooled workers (workers.h), H's — the last to arrive — on this thread
    HostTask tt[4];
    tt[0].fn = [&] { g1tail(0, Ww, bw1, cw, (size_t)0); };
    tt[1].fn = [&] { g1tail(1, Ww, bw1, cw, (size_t)96); };
    tt[3].fn = [&] { g1tail(3, Ww, bw1, cw, (size_t)384); };
    tt[2].fn = [&] {
      (void)hipSetDevice(dev);
      (void)hipEventSynchronize(evd[2]);
      if (gw.tab) msm_g2_host_tail_tab(HP + 2 * PARTIALS_STRIDE, Wb, (bn254_g2_projective_t*)(out_points + 192));
      else msm_g2_host_tail(HP + 2 * PARTIALS_STRIDE, Wb, 1, cw, gw.wide, (bn254_g2_projective_t*)(out_points + 192));
      if (et) {
        while (!et->bl_ready.load(std::memory_order_acquire)) std::this_thread::yield();
        bn254_g2_projective_t b2;
        memcpy(&b2, out_points + 192, sizeof b2);
        early_pi_b(z, *et->bl, &b2, et);
      }
    };
    bool pooled[4];
    for (int k = 0; k < 4; k++) pooled[k] = WorkerPool::get().submit(&tt[k]);
    g1tail(4, Wh, bh, ch, 480);
    for (int k = 0; k < 4; k++) {
      if (pooled[k]) WorkerPool::wait(&tt[k]);
      else tt[k].fn(); // (no worker was to be had: after H's, on this thread)
    }
  }
  mark("tails");
  // every tail thread has waited for its MSM's ev_done (recorded behind the last operation of that MSM's chain), so all six
  // streams are drained except for ev[3] on g1, which waits for the five of them: one synchronisation instead of six
  P_HIP(hipStreamSynchronize(g1));
  mark("drained");
  drain.armed = false;
  // (the stats copies sit on g2 / g3 in front of work whose ev_done a tail thread has waited for)
  z->witness_entries = (uint64_t)z->h_stats[0] + z->h_stats[1] + z->h_stats[2] + z->h_stats[3];
  z->proves_since_rebuild++;
  z->tb.go.store(true, std::memory_order_release); // deferred tables: the key's first proof is out, the build may start
  if (!sync_rebuild) (void)follow_witness(false);  // (table mode: the digits this prove counted may call for another width)
  msm_sort_release(&plan_w);
  msm_sort_release(&plan_head);
  msm_sort_release(&plan_h);
  // HIP-event profile of the five MSMs → the ring icicle_snark_msm_profile reads (bench.py: back = 4 … 0 = A, B1, B2, C, H); in
  // a device group only the lead shard publishes
  // (the head's sort first, so that A … H keep their places: back = 5 is the head sort, L = 0 when the witness was not split)
  if (z->shard_rank == 0 || !z->in_group) {
    msm_profile_publish(&z->prof[5], 1);
    msm_profile_publish(z->prof, 5);
  }
  if (head) {
    // The head's share follows the machine: its chain should end just before the last byte of the witness lands — what is left
    // of it then runs beside the transforms of the front end and slows them (long-lived accumulation workgroups hold the
    // registers the transform workgroups need), what ends earlier leaves the GPU idle.  gap > 0: the head ended `gap` ms
    // before the upload; a larger head both starts later (its bytes land later) and runs longer.
    float work = 0, gap = 0, up = 0;
    if (hipEventElapsedTime(&work, z->ev_t_head_start, z->ev_t_head_end) == hipSuccess && hipEventElapsedTime(&gap, z->ev_t_head_end, z->ev_t_witness) == hipSuccess &&
        hipEventElapsedTime(&up, z->ev[0], z->ev_t_witness) == hipSuccess && work > 0 && up > 0) {
      // Measured at 1.6 M constraints (profiles/r04_head_sweep.txt): ending 0.3 ms AFTER the upload is better than ending with
      // it (12 % of the witness against 8 %: 16.1 against 16.4 ms file to file, 16.9–17.3 without a head) — the transforms
      // lose less to the last of the head's workgroups than the MSM phase gains — and the optimum is flat beyond that.
      const double f = (double)head / nv, per_unit = work / f + up; // ms by which the head's end moves per unit of share
      const double margin = -0.30;
      double f_new = f + 0.8 * ((double)gap - margin) / per_unit; // (0.8: per_unit is measured, the step lands near the target at once; the hysteresis below absorbs the jitter of the upload)
      if (f_new > 0.40) f_new = 0.40;
      // the share moves in whole upload chunks and only when it is off by most of one: a head of another size is another set of
      // workspace blocks (a fresh hipMalloc inside a prove when it grows)
      const double target = f_new * nv / head_unit, cur = (double)head / head_unit;
      if (!head_forced && (target > cur + 0.75 || target < cur - 0.75)) {
        z->head_units = (int)(target + 0.5);
        if (z->head_units < 1) z->head_units = 1;
      }
      if (trace_host) fprintf(stderr, "[host] head %.1f %% of the witness: chain %.3f ms, ended %.3f ms before the upload (%.3f ms) -> next %.1f %%\n", 100 * f, work, gap, up, 100.0 * (head_forced ? head : (uint32_t)z->head_units * head_unit) / nv);
    } else
      (void)hipGetLastError();
  }
  {
    float a = 0, b = 0, c = 0;
    (void)hipEventElapsedTime(&a, z->ev[0], z->ev[1]);
    (void)hipEventElapsedTime(&b, z->ev[1], z->ev[2]);
    (void)hipEventElapsedTime(&c, z->ev[2], z->ev[3]);
    z->last_tm.h2d_ms = h2d_host_ms > a ? h2d_host_ms : a; // staged upload: the host waited for it; pinned source: the DMA in front of ev[1]
    z->last_tm.qap_ms = b;
    z->last_tm.msm_ms = c;
    z->last_tm.total_ms = ms_since(t0);
    if (tm) *tm = z->last_tm;
  }
  mark("published");
  return 0;
}

// ---- distributed front end: the stages of one shard, enqueued on z->s_qap without host synchronisation --------------------
bool shard_dist_supported(const ZKeyCache* z)
{
  const uint32_t G = (uint32_t)z->shard_count;
  if (z->H.stride <= 1 || (G != 2 && G != 4 && G != 8)) return false;
  const uint32_t m = z->domain_size / G;
  return m % G == 0 && ntt_fusable(m);
}

static int check_witness(const ZKeyCache* z, const Wtns& w)
{
  // src/proof_helper.rs:253-262
  if (!Fr::eq(z->r, w.q)) return fail(ERR_FORMAT, "Curve of the witness does not match the curve of the proving key");
  if (w.n_witness != z->n_vars) return fail(ERR_FORMAT, "Invalid witness length. Circuit: %u, witness: %u", z->n_vars, w.n_witness);
  return 0;
}

// this shard's 1/shard_count of the witness → its place in d_witness (returns when the bytes have arrived)
int shard_upload_slice(ZKeyCache* z, const Wtns& w)
{
  if (int rc = check_witness(z, w)) return rc;
  z->witness_resident = false; // until the all-gather has completed the buffer (groth16_witness_ready / the group prove)
  z->dist_ready = z->dist_stage2_done = false;
  z->witness_event_set = false;
  const uint64_t slice = witness_slice_elems(z->n_vars, z->shard_count);
  const uint64_t lo = std::min<uint64_t>(z->n_vars, slice * (uint64_t)z->shard_rank), hi = std::min<uint64_t>(z->n_vars, lo + slice);
  if (hi > lo) {
    const hipStream_t lanes[3] = {z->s_qap, z->s_g2, z->s_g3};
    if (int rc = staged_upload(z->device_id, {{z->d_witness + lo, (const uint8_t*)w.values + lo * 32, (size_t)(hi - lo) * 32}}, lanes, 3)) return rc;
  }
  return 0;
}

// stage 1: spmv of the rows c ≡ rank (mod G) and their size-m inverse transform → d_dist_y (what exchange 1 sends)
int shard_dist_stage1(Groth16CacheManager* cm, ZKeyCache* z)
{
  if (int rc = ensure_domain(cm, z)) return rc;
  const uint32_t n = z->domain_size, G = (uint32_t)z->shard_count, r = (uint32_t)z->shard_rank, m = n / G;
  z->dist_ready = z->dist_stage2_done = false;
  hipStream_t gq = z->s_qap;
  int dom_log = 0;
  const fe* tw = ntt_domain_table(&dom_log);
  if (!z->d_tw1) {
    // all four buffers or none: a failed allocation must not leave a half-initialised entry behind for the next call
    fe* b[4] = {nullptr, nullptr, nullptr, nullptr};
    const size_t sz[4] = {(size_t)3 * m * 32, (size_t)3 * m * 32, (size_t)3 * m * 32, (size_t)m * 32};
    for (int k = 0; k < 4; k++)
      if (hipMalloc((void**)&b[k], sz[k]) != hipSuccess) {
        (void)hipGetLastError();
        for (int j = 0; j < k; j++) (void)hipFree(b[j]);
        return fail((int)ICICLE_ALLOCATION_FAILED, "distributed front end: cannot allocate the exchange buffers (%zu bytes)", sz[k]);
      }
    if (hipError_t e = qap_dist_tw1(tw, 1u << dom_log, n, G, r, b[3], gq)) {
      for (fe* p : b) (void)hipFree(p);
      return fail((int)ICICLE_UNKNOWN_ERROR, "qap_dist_tw1: %s", hipGetErrorString(e));
    }
    z->d_dist_y = b[0];
    z->d_dist_recv1 = b[1];
    z->d_dist_send2 = b[2];
    z->d_tw1 = b[3];
    z->device_bytes += (size_t)10 * m * 32;
  }
  P_HIP(qap_spmv_strided(z->d_witness, z->d_rowptr, z->d_cols, z->d_vals, n, G, r, z->d_dist_y, gq));
  NttFuse f;
  f.scale_tab = z->d_tw1;
  P_ICICLE(ntt_fused(z->d_dist_y, m, 3, true, gq, f));
  return 0;
}

// stage 2: what exchange 1 delivered (d_dist_recv1) → d_dist_send2 (what exchange 2 sends; it delivers into d_fold)
int shard_dist_stage2(ZKeyCache* z)
{
  if (!z->d_dist_recv1) return fail(ERR_ARG, "stage 1 of the distributed front end has not run for this entry");
  const uint32_t n = z->domain_size, G = (uint32_t)z->shard_count, r = (uint32_t)z->shard_rank;
  int dom_log = 0;
  const fe* tw = ntt_domain_table(&dom_log);
  if (!tw || (1u << dom_log) < 2 * n) return fail((int)ICICLE_INVALID_ARGUMENT, "the NTT domain was released between the stages");
  P_HIP(qap_dist_mid(z->d_dist_recv1, z->d_dist_send2, tw, 1u << dom_log, n, G, r, z->s_qap));
  z->dist_stage2_done = true;
  return 0;
}

} // namespace prover
} // namespace isnark

extern "C" {

// ---- distributed front end over the C API (one process per GPU: include/groth16_prover.h) --------------------------------
__attribute__((visibility("default"))) int groth16_dist_supported(Groth16CacheManager* cm, const char* key)
{
  if (!cm) return 0;
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  return zp && shard_dist_supported(zp.get()) ? 1 : 0;
}

// Multi-GPU witness distribution (see include/groth16_prover.h)
__attribute__((visibility("default"))) int groth16_upload_witness_slice(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, void** d_witness, uint64_t* slice_bytes)
{
  if (!cm || !wtns || !d_witness || !slice_bytes) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  ZKeyCache* z = zp.get();
  if (!z) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  if (int rc = set_active_device(z->device_id)) return rc;
  Wtns w;
  if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
  if (int rc = shard_upload_slice(z, w)) return rc;
  *d_witness = z->d_witness;
  *slice_bytes = witness_slice_elems(z->n_vars, z->shard_count) * 32;
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
  if (int rc = set_active_device(z->device_id)) return rc;
  if (wtns) {
    Wtns w;
    if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
    if (int rc = check_witness(z, w)) return rc;
    z->witness_resident = false;
    z->dist_ready = z->dist_stage2_done = false;
    z->witness_event_set = false;
    const hipStream_t lanes[3] = {z->s_qap, z->s_g2, z->s_g3};
    if (int rc = staged_upload(z->device_id, {{z->d_witness, (const uint8_t*)w.values, (size_t)z->n_vars * 32}}, lanes, 3)) return rc;
    z->witness_resident = true;
  } else if (!z->witness_resident)
    return fail(ERR_ARG, "no witness given and none resident on the device");
  if (int rc = shard_dist_stage1(cm, z)) return rc;
  P_HIP(hipStreamSynchronize(z->s_qap)); // the caller's exchange runs on its communicator's stream
  const uint32_t G = (uint32_t)z->shard_count, m = z->domain_size / G;
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
  if (int rc = set_active_device(z->device_id)) return rc;
  if (int rc = shard_dist_stage2(z)) return rc;
  P_HIP(hipStreamSynchronize(z->s_qap));
  *d_send = z->d_dist_send2;
  *d_recv = z->d_fold; // exchange 2 delivers this rank's Z rows [row][m] where stage 3 (groth16_commitments) expects them
  return 0;
}

// The caller confirms that exchange 2 has delivered into the receive buffer of groth16_dist_stage2: only then does the next
// groth16_commitments(wtns = NULL) skip its own inverse transform + fold and finish from d_fold.  Without this call (a failed
// exchange, a retry with a fresh witness) the commitments call computes everything itself, replicated.
__attribute__((visibility("default"))) int groth16_dist_exchange_done(Groth16CacheManager* cm, const char* key)
{
  if (!cm) return fail(ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(cm->mu);
  const std::shared_ptr<ZKeyCache> zp = find(cm, key);
  if (!zp) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  if (!zp->dist_stage2_done || !zp->witness_resident) return fail(ERR_ARG, "groth16_dist_stage2 has not run for the resident witness of '%s'", key ? key : "");
  zp->dist_ready = true;
  return 0;
}

} // extern "C"

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
  const std::shared_ptr<DeviceGroup> grp = find_group(cm, key);
  const std::shared_ptr<ZKeyCache> zp = grp ? nullptr : find(cm, key);
  const ZKeyCache* z = grp ? group_lead(grp.get()) : zp.get();
  if (!z) return fail(ERR_NOCACHE, "no cache entry '%s'", key ? key : "");
  // A sharded cache holds only this rank's point range: its commitments are PARTIAL sums and a proof assembled from them
  // alone would be well-formed but invalid.  Shards prove together as a device group (groth16_cache_load_devices, one
  // process) or through groth16_commitments → all-gather → groth16_sum_commitments → groth16_assemble_proof (one process per GPU).
  if (!grp && z->shard_count != 1) return fail(ERR_ARG, "cache entry '%s' is shard %d of %d: use groth16_commitments + groth16_sum_commitments + groth16_assemble_proof", key ? key : "", z->shard_rank, z->shard_count);
  // r, s and the commitment-independent blinding terms on a host thread while the GPU computes the commitments
  Blinding bl;
  EarlyTerms et;
  et.bl = &bl;
  int bl_rc = 0;
  std::string bl_err;
  HostTask bt;
  bt.fn = [&] {
    bl_rc = compute_blinding(z, r, s, &bl);
    if (bl_rc) bl_err = last_error_text();
    et.bl_ready.store(true, std::memory_order_release);
  };
  WorkerPool::get().run_or_inline(&bt); // (inline when no worker is to be had: before the commitments instead of beside them)
  int rc;
  {
    std::lock_guard<std::mutex> lk(cm->mu);
    // a device group sums the shards' commitments first: the early C-term products (EarlyTerms) need the complete A and B1
    rc = grp ? group_commitments(cm, grp.get(), wtns_resident ? nullptr : wtns, wtns_len, pts, tm)
             : shard_commitments(cm, zp.get(), wtns_resident ? nullptr : wtns, wtns_len, pts, tm, &et);
  }
  if (bt.queued) WorkerPool::wait(&bt);
  if (rc) return rc;
  if (bl_rc) return fail(bl_rc, "%s", bl_err.empty() ? "no entropy source for the blinding scalars" : bl_err.c_str());
  const auto ta = std::chrono::steady_clock::now();
  rc = assemble_impl(z, wtns, wtns_len, pts, bl, proof_json, proof_cap, public_json, public_cap, grp ? nullptr : &et);
  if (getenv("ICICLE_SNARK_TRACE_HOST")) fprintf(stderr, "[host] assemble %8.1f us (after %8.1f us)\n", ms_since(ta) * 1e3, std::chrono::duration<double, std::micro>(ta - t0).count());
  if (tm) tm->total_ms = ms_since(t0);
  return rc;
}

} // extern "C"

// proof.json and public.json (src/lib.rs:52-57), the second on a pooled worker beside the first: creating and closing a file
// costs 20–80 µs on the boxes measured, and both are inside the reference's timed region
static int write_json_pair(const char* proof_path, const char* proof_text, const char* public_path, const char* public_text)
{
  auto put = [](const char* path, const char* text) -> bool {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) return false;
    const size_t n = strlen(text);
    size_t done = 0;
    while (done < n) {
      const ssize_t w = write(fd, text + done, n - done);
      if (w <= 0) break;
      done += (size_t)w;
    }
    return close(fd) == 0 && done == n;
  };
  bool ok_public = false;
  HostTask t;
  t.fn = [&] { ok_public = put(public_path, public_text); };
  WorkerPool::get().run_or_inline(&t);
  const bool ok_proof = put(proof_path, proof_text);
  if (t.queued) WorkerPool::wait(&t);
  if (!ok_proof) return fail(ERR_IO, "cannot write %s", proof_path);
  if (!ok_public) return fail(ERR_IO, "cannot write %s", public_path);
  return 0;
}

// ---- cold pipeline (prover_internal.h: ColdFeed): groth16_prove on a key that is not cached, one device ---------------------------
// The cache entry is built with its sections still on their way (build_cache with a ColdUpload: an uploader task sends coefficients,
// witness and point sections in the order the prove needs them), inserted, and the prove is enqueued behind the stages of the feed:
// upload and first proof overlap.  Returns 1 when the pipeline does not apply (the caller then loads and proves as before: a witness
// that does not fit the key is diagnosed there), 0 on success with the JSON texts filled, an error code otherwise (the key is evicted
// again when its upload failed).
// returns COLD_DECLINED when the pipeline does not apply (the caller then loads and proves one after the other) — a value no
// eIcicleError (≥ 0) and no ERR_* code of this library (small negatives) can take (round-5 advisor: `1` is ICICLE_INVALID_DEVICE)
static constexpr int COLD_DECLINED = INT_MIN;
static int cold_prove(Groth16CacheManager* cm, const std::string& key, const MappedFile& zf, const MappedFile& wf, int device_id, std::vector<char>& pj, std::vector<char>& qj)
{
  const bool off = getenv("ICICLE_SNARK_COLD_PIPELINE") && atoi(getenv("ICICLE_SNARK_COLD_PIPELINE")) == 0; // read per call: tests toggle it
  if (off) return COLD_DECLINED;
  // the witness must fit the key BEFORE anything of it is sent: n_vars of the header (src/zkey.rs:47-85) against the .wtns header
  Wtns w;
  if (parse_wtns(wf.data, wf.len, w)) return COLD_DECLINED;
  uint32_t n_vars = 0, dom_n = 0, n_public = 0;
  {
    std::vector<Section> secs;
    const Section* s2 = nullptr;
    if (read_sections(zf.data, zf.len, "zkey", 2, secs) != 0 || unique_section(secs, 2, &s2) != 0 || s2->size < 84) return COLD_DECLINED;
    memcpy(&n_vars, s2->p + 72, 4);
    memcpy(&n_public, s2->p + 76, 4);
    memcpy(&dom_n, s2->p + 80, 4);
    if (w.n_witness != n_vars || dom_n == 0 || (dom_n & (dom_n - 1)) || dom_n > (1u << 27) || memcmp(w.q.l, s2->p + 40, 32) != 0) return COLD_DECLINED;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::unique_lock<std::mutex> lk(cm->mu);
  if (find(cm, key.c_str()) || find_group(cm, key.c_str())) return COLD_DECLINED; // (somebody else loaded it meanwhile)
  if (cm->warm.joinable()) cm->warm.join();
  evict_for_budget(cm, device_id, (uint64_t)zf.len * 11);
  std::thread dom_th; // the NTT domain of the key, set up while the sections cross PCIe (as in groth16_cache_load)
  try {
    dom_th = std::thread([cm, device_id, dom_n, t0] {
      if (set_active_device(device_id) == 0) (void)ensure_domain_for(cm, device_id, dom_n);
      if (getenv("ICICLE_SNARK_TRACE_COLD")) fprintf(stderr, "[cold] NTT domain ready at      %8.2f ms\n", ms_since(t0));
    });
  } catch (...) {
  }
  ColdUpload cu;
  cu.wtns_values = w.values;
  cu.wtns_bytes = (size_t)w.n_witness * 32;
  cu.zkey_base = zf.data; cu.zkey_len = zf.len; cu.zkey_fd = zf.fd;
  cu.wtns_base = wf.data; cu.wtns_len = wf.len; cu.wtns_fd = wf.fd;
  std::unique_ptr<ZKeyCache> zu;
  const int brc = build_cache(zf.data, zf.len, device_id, 0, 1, zu, /*defer_tables=*/true, &cu);
  if (brc) {
    cold_upload_wait(&cu);
    if (dom_th.joinable()) dom_th.join();
    return brc;
  }
  std::shared_ptr<ZKeyCache> zp(zu.release());
  {
    std::lock_guard<std::mutex> lm(cm->map_mu);
    zp->last_use = ++cm->clock;
    cm->cache[key] = zp;
  }
  zp->tb.narrow_room.store(budget_room(cm, device_id), std::memory_order_release);
  ZKeyCache* z = zp.get();
  const bool piped = z->feed != nullptr;
  // from here on as groth16_prove_resident, under the manager's mutex throughout
  uint8_t pts[GROTH16_COMMITMENTS_BYTES];
  Blinding bl;
  EarlyTerms et;
  et.bl = &bl;
  int bl_rc = 0;
  std::string bl_err;
  HostTask bt;
  bt.fn = [&] {
    bl_rc = compute_blinding(z, nullptr, nullptr, &bl);
    if (bl_rc) bl_err = last_error_text();
    et.bl_ready.store(true, std::memory_order_release);
  };
  WorkerPool::get().run_or_inline(&bt);
  if (dom_th.joinable()) dom_th.join(); // (shard_commitments checks the domain first: it is there, or it is set up now)
  int rc = shard_commitments(cm, z, wf.data, wf.len, pts, nullptr, &et);
  if (bt.queued) WorkerPool::wait(&bt);
  // the uploader has to have ended before the mappings go away — and before anybody else proves with this entry
  cold_upload_wait(&cu);
  z->feed = nullptr;
  const int up_rc = cu.feed.rc;
  if (!up_rc) z->tb.hold.store(false, std::memory_order_release); // the sections are complete: the deferred table build may read them
  const std::string up_err = cu.feed.err;
  lk.unlock();
  if (up_rc) {
    groth16_cache_evict(cm, key.c_str()); // its sections never arrived completely
    return fail(up_rc, "%s", up_err.c_str());
  }
  if (rc) return rc;
  if (bl_rc) return fail(bl_rc, "%s", bl_err.empty() ? "no entropy source for the blinding scalars" : bl_err.c_str());
  qj.resize(64 + (size_t)n_public * 84);
  rc = assemble_impl(z, wf.data, wf.len, pts, bl, pj.data(), pj.size(), qj.data(), qj.size(), &et);
  if (getenv("ICICLE_SNARK_TRACE_HOST")) fprintf(stderr, "[host] cold prove (%s) %8.1f us\n", piped ? "pipelined" : "not pipelined", ms_since(t0) * 1e3);
  return rc;
}

extern "C" {

// groth16_prove — src/lib.rs:33-61.  `device`: the reference's device type string ("CUDA" there, id 0: src/lib.rs:25-31);
// here "HIP" (alias "CUDA"), optionally with the devices to prove on: "HIP:1", "HIP:0-7", "HIP:0,2,4,6" — more than one
// device = the MSMs sharded by point range over a device group in this process.
__attribute__((visibility("default"))) int groth16_prove(const char* witness_path, const char* zkey_path, const char* proof_path, const char* public_path, const char* device, Groth16CacheManager* cm)
{
  if (!witness_path || !zkey_path || !proof_path || !public_path || !device || !cm) return fail(ERR_ARG, "null argument");
  const auto t0 = std::chrono::steady_clock::now();
  // try_load_and_set_backend_device — src/lib.rs:25-31
  std::vector<int> devs;
  if (int rc = parse_device_string(device, devs)) return rc; // "CPU" (or anything but HIP/CUDA) fails here: no CPU fallback
  P_ICICLE(icicle_load_backend_from_env_or_default());
  if (int rc = set_active_device(devs[0])) return rc;
  const std::string key = std::string(zkey_path) + "_" + device; // src/lib.rs:44
  static const bool trace_host0 = getenv("ICICLE_SNARK_TRACE_HOST") != nullptr;
  static const bool quiet0 = getenv("ICICLE_SNARK_QUIET") && atoi(getenv("ICICLE_SNARK_QUIET")) != 0;
  if (!groth16_cache_contains(cm, key.c_str()) && devs.size() == 1) {
    // no cache entry: the key's sections and the witness cross PCIe while the first proof is being computed (cold_prove)
    MappedFile zf, wf;
    if (int rc = zf.open_ro(zkey_path)) return rc;
    if (int rc = wf.open_ro(witness_path)) return rc;
    std::vector<char> pj(4096), qj(256);
    const int crc = cold_prove(cm, key, zf, wf, devs[0], pj, qj);
    if (crc == 0) {
      if (int rc = write_json_pair(proof_path, pj.data(), public_path, qj.data())) return rc;
      if (trace_host0) fprintf(stderr, "[host] prove: files written (cold)   %8.1f us\n", ms_since(t0) * 1e3);
      if (!quiet0) {
        printf("proof took: %.3fms\n", ms_since(t0)); // src/lib.rs:58
        fflush(stdout);
      }
      return 0;
    }
    if (crc != COLD_DECLINED) return crc;
    // (declined: the pipeline does not apply — load and prove one after the other, below)
  }
  if (!groth16_cache_contains(cm, key.c_str())) {
    if (devs.size() == 1) {
      if (int rc = groth16_cache_load_file(cm, key.c_str(), zkey_path, devs[0], 0, 1)) return rc;
    } else {
      MappedFile zf;
      if (int rc = zf.open_ro(zkey_path)) return rc;
      if (int rc = groth16_cache_load_devices(cm, key.c_str(), zf.data, zf.len, devs.data(), (int)devs.size())) return rc;
    }
  }
  static const bool trace_host = getenv("ICICLE_SNARK_TRACE_HOST") != nullptr;
  if (trace_host) fprintf(stderr, "[host] prove: cache key found        %8.1f us\n", ms_since(t0) * 1e3);
  MappedFile wf;
  if (int rc = wf.open_ro(witness_path)) return rc;
  if (trace_host) fprintf(stderr, "[host] prove: witness mapped         %8.1f us\n", ms_since(t0) * 1e3);
  std::vector<char> pj(4096), qj(256);
  {
    Groth16CircuitInfo info;
    if (int rc = groth16_cache_info(cm, key.c_str(), &info)) return rc;
    qj.resize(64 + (size_t)info.n_public * 84);
  }
  // the witness values go from the page cache straight into the upload workers' pinned buffers (pread) instead of being copied
  // out of the mapping, which is then only touched for the header and the public signals: −0.3 ms per prove at 1.6 M
  // constraints (page faults of a 51 MB mapping)
  staged_copy_file_hint(wf.data, wf.len, wf.fd);
  const int prc = groth16_prove_mem(cm, key.c_str(), wf.data, wf.len, nullptr, nullptr, pj.data(), pj.size(), qj.data(), qj.size(), nullptr);
  staged_copy_file_hint(nullptr, 0, -1);
  if (prc) return prc;
  if (trace_host) fprintf(stderr, "[host] prove: proof assembled        %8.1f us\n", ms_since(t0) * 1e3);
  if (int rc = write_json_pair(proof_path, pj.data(), public_path, qj.data())) return rc;
  if (trace_host) fprintf(stderr, "[host] prove: files written          %8.1f us\n", ms_since(t0) * 1e3);
  static const bool quiet = getenv("ICICLE_SNARK_QUIET") && atoi(getenv("ICICLE_SNARK_QUIET")) != 0;
  if (!quiet) {
    printf("proof took: %.3fms\n", ms_since(t0)); // src/lib.rs:58
    fflush(stdout);
  }
  return 0;
}

} // extern "C"
