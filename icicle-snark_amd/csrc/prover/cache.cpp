// cache.cpp — CacheManager::compute (src/cache.rs:117-241): zkey sections 4-9 → the device-resident ZKeyCache of one
// device (or of one shard of a device group).
#include <algorithm>
#include <fcntl.h>
#include <errno.h>
#include <sys/mman.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <unistd.h>

#include "prover_internal.h"

using namespace bn254;
using namespace isnark;
using namespace isnark::prover;

namespace isnark {
namespace prover {

ZKeyCache::~ZKeyCache()
  {
    // (an eviction must not leave the caller's thread on another device)
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    (void)hipSetDevice(device_id);
    // a table build still under way stops at its next slice; complete tables nobody adopted go with the key
    tb.cancel.store(true);
    if (tb.th.joinable()) tb.th.join();
    for (void* t : tb.fresh)
      if (t) (void)hipFree(t);
    if (s_qap) (void)hipStreamSynchronize(s_qap);
    if (s_g1) (void)hipStreamSynchronize(s_g1);
    if (s_g2) (void)hipStreamSynchronize(s_g2);
    if (s_g3) (void)hipStreamSynchronize(s_g3);
    if (s_g4) (void)hipStreamSynchronize(s_g4);
    if (s_g5) (void)hipStreamSynchronize(s_g5);
    for (void* p : {(void*)d_rowptr, (void*)d_cols, (void*)d_vals, A.d_points, B1.d_points, B2.d_points, C.d_points, H.d_points, (void*)d_witness, (void*)d_vec, (void*)d_fold, (void*)d_skeys, (void*)d_dist_y, (void*)d_dist_recv1, (void*)d_dist_send2, (void*)d_tw1, (void*)d_partials})
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
    if (ev_sort_h) (void)hipEventDestroy(ev_sort_h);
    for (hipEvent_t e : {ev_own_slice, ev_head_in, ev_head_done, ev_t_head_start, ev_t_head_end, ev_t_witness})
      if (e) (void)hipEventDestroy(e);
    for (auto e : ev)
      if (e) (void)hipEventDestroy(e);
    for (auto e : ev_done)
      if (e) (void)hipEventDestroy(e);
    for (auto e : ev_lfork)
      if (e) (void)hipEventDestroy(e);
    for (auto e : ev_ljoin)
      if (e) (void)hipEventDestroy(e);
    for (auto& p : prof) msm_profile_own_destroy(&p);
  }

namespace {
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
} // namespace

// ---- cold path: host → device ingest ------------------------------------------------------------
// The zkey arrives as pageable memory (an mmap of the file, or the caller's buffer).  A pageable hipMemcpy is a
// single-threaded staging copy; isnark::staged_copy (runtime.cpp) runs up to eight workers that copy 2 MB chunks into
// their own pair of pinned buffers and enqueue the DMAs on their own streams, so page faults / memcpy of one chunk
// overlap the DMA of the others.  `lanes`: streams to enqueue the DMAs on (the per-prove witness upload passes the
// prover's own streams, idle at that point); nullptr: short-lived streams of the call (cold path).
int staged_upload(int device_id, const std::vector<UploadJob>& jobs, const hipStream_t* lanes_in, int n_lanes, StagedProgress* progress)
{
  const hipError_t e = staged_copy(device_id, jobs.data(), jobs.size(), true, lanes_in, n_lanes, /*own_temp_streams=*/lanes_in == nullptr, progress);
  if (e != hipSuccess) return fail((int)ICICLE_COPY_FAILED, "host to device upload: %s", hipGetErrorString(e));
  return 0;
}

namespace {
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
} // namespace

// ---- digit width of a key's witness tables (A, B1, B2, C) for DENSE scalars --------------------------------------------------------
// msm_geometry's table rule takes the widest digit the sort entry holds (c = 20 from 2^19 wires on): fewest additions.  For the four
// witness MSMs that is not the fastest below ≈ 28 entries per bucket: their bucket sets are reduced four times over (the G2 one
// alone is a millisecond of latency-bound work) while the accumulations they save are throughput-bound — measured (round 6,
// profiles/r06_witness_digit_width.txt): 800 k constraints c = 19 instead of 20: 8.35 → 8.19 ms, 400 k c = 18 instead of 19: 4.72 → 4.39 ms;
// 1600 k (40 entries per bucket at c = 20) and 200 k (46 at c = 17) stay; below c = 17 the bucket count no longer fills the GPU (one
// thread per bucket: 100 k with c = 16 is 0.35 ms slower).  H keeps the widest digit (narrower measured slower at 400 k and 800 k).
MsmGeom witness_table_geometry(uint32_t len)
{
  MsmGeom g = msm_geometry(len, 0, 1);
  if (!g.tab) return g;
  int c = g.c;
  while (c > 17 && (((uint64_t)len * (uint64_t)(254 / c + 1)) >> (c - 1)) < 28) c--;
  if (c != g.c) {
    const MsmGeom h = msm_geometry(len, 0, c);
    if (h.tab && h.c == c) return h;
  }
  return g;
}

// ---- which physical stream plays which role ----------------------------------------------------------------------------------------
// The six streams of a key sit on hardware queues, and queue k on pipe k mod 4 — in whatever order the process happened to use its
// streams first (pooled streams, prewarm threads, the stream of a table build, other keys).  Two pairs of the six share a pipe, and a
// kernel with more workgroups than the GPU holds delays everything else on its pipe by 0.5–1 ms while it dispatches (microbench.hip:
// probe_stream_pipes; profiles/r05_pipe_probe.txt).  Left to chance, the witness digit sort was bimodal from one key instance to the
// next — 1.05 or 2.3–2.8 ms inside the prove: its stream on the front end's pipe or not — and with it the start of the accumulations.
// So the pipes are MEASURED (once per process and stream) and the roles dealt the way six consecutive queues fall: the front end shares
// its pipe with B1 + H's stream and A's with C's — streams that idle while their mate is busy —, the sort / B2 stream and the
// H-sort / tail-sort stream have a pipe each.  Measured over five boxes (profiles/r06_stream_roles_by_pipe.txt): the sort 1.03–1.12 ms
// in every run, the prove −0.2 … ±0 ms; "everything apart" (the four accumulation chains on four pipes, the front end beside A) is
// +0.1–0.25 ms and the sort ON the front end's pipe +0.5 ms (it then ends after the front end and the accumulations wait for it).
namespace {
void assign_stream_roles(ZKeyCache* z, bool allow_measure)
{
  static const bool off = getenv("ICICLE_SNARK_PIPE_ROLES") && atoi(getenv("ICICLE_SNARK_PIPE_ROLES")) == 0;
  if (off) return;
  hipStream_t phys[6] = {z->s_qap, z->s_g1, z->s_g2, z->s_g3, z->s_g4, z->s_g5};
  int cls[6];
  // (the shards of a device group do not measure: with all of them on one device — the aliased test groups — 48 streams share 12
  //  queues and the classes mean nothing, and on G devices G probes would sit in the group's load; a shard whose streams the prewarm
  //  thread has measured — rank-per-GPU processes, the group's first device — still gets its roles)
  if (!stream_pipe_classes(phys, 6, cls, allow_measure)) return;
  // roles: 0 front end, 1 A, 2 B2 (+ witness sort / head chain), 3 H sort (+ tail sort of a split witness), 4 B1 + H, 5 C
  auto cost = [&](const int* perm) {
    auto same = [&](int a, int b) { return cls[perm[a]] == cls[perm[b]] ? 1 : 0; };
    int c = 1000 * (same(0, 2) + same(0, 3) + same(2, 3)) + 200 * (same(2, 1) + same(2, 4) + same(2, 5) + same(3, 1) + same(3, 4) + same(3, 5)) + 100 * (1 - same(0, 4)) +
            100 * (1 - same(1, 5));
    for (int r = 0; r < 6; r++) c += perm[r] != r ? 1 : 0; // among equals: the fewest moves
    return c;
  };
  int perm[6] = {0, 1, 2, 3, 4, 5}, best[6] = {0, 1, 2, 3, 4, 5};
  int best_cost = cost(perm);
  while (std::next_permutation(perm, perm + 6)) {
    const int c = cost(perm);
    if (c < best_cost) {
      best_cost = c;
      memcpy(best, perm, sizeof best);
    }
  }
  z->s_qap = phys[best[0]];
  z->s_g1 = phys[best[1]];
  z->s_g2 = phys[best[2]];
  z->s_g3 = phys[best[3]];
  z->s_g4 = phys[best[4]];
  z->s_g5 = phys[best[5]];
  if (getenv("ICICLE_SNARK_TRACE_COLD") || getenv("ICICLE_SNARK_VERBOSE"))
    fprintf(stderr, "[icicle-snark-hip] stream pipes %d %d %d %d %d %d -> roles (qap A B2 Hsort B1 C) take streams %d %d %d %d %d %d (cost %d)\n", cls[0], cls[1], cls[2], cls[3], cls[4], cls[5], best[0],
            best[1], best[2], best[3], best[4], best[5], best_cost);
}
} // namespace

// CacheManager::compute — src/cache.rs:117-241
// ---- deferred fixed-base tables (prover_internal.h: TableBuild) --------------------------------------------------------------
namespace {
void table_build_thread(ZKeyCache* z)
{
  TableBuild& tb = z->tb;
  // behind the key's first proof, not beside it: a prove next to the build takes 24 instead of 19 ms at 1.6 M constraints, and
  // the first one is the one a caller without a cache waits for.  A key nobody proves with gets its tables after the grace time.
  // (cold pipeline: not before the key's sections have landed — cold_prove clears `hold` behind the upload, or evicts the key)
  while (tb.hold.load(std::memory_order_acquire) && !tb.cancel.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  const int grace_ms = getenv("ICICLE_SNARK_TABLE_GRACE_MS") ? atoi(getenv("ICICLE_SNARK_TABLE_GRACE_MS")) : TABLE_BUILD_GRACE_MS;
  for (int waited = 0; !tb.witness_only && waited < grace_ms && !tb.go.load(std::memory_order_acquire) && !tb.cancel.load(); waited++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  const auto t0 = std::chrono::steady_clock::now();
  static const bool trace_tb = getenv("ICICLE_SNARK_TRACE_TABLES") != nullptr;
  bool ok = hipSetDevice(z->device_id) == hipSuccess;
  if (trace_tb) fprintf(stderr, "[tables] thread: device set at %.1f ms\n", ms_since(t0));
  // The key's first prove (classic layout) has counted the non-zero digits of its witness: a witness of 0 / 1 wires and small values
  // wants narrower digits than the dense default (witness_digit_target; HISTORY.md §9-2a) — build the four witness tables with that width
  // at once instead of building the dense ones and rebuilding them INSIDE a later prove (0.1–0.3 s).  The classic count (16-bit digits)
  // is an upper bound of the table-mode one, so the width chosen here is the rule's or one bit above it; the rule keeps following the
  // witnesses afterwards.  (witness_entries was written before `go` was released.)
  const MsmGeom gw_dense = tb.gw;
  bool narrowed = false;
  if (ok && !tb.witness_only && tb.go.load(std::memory_order_acquire) && z->witness_entries) {
    int lg = 0;
    while (((uint64_t)1 << lg) * 32 < z->witness_entries) lg++;
    int c_t = lg + 1 < 13 ? 13 : lg + 1;
    if (c_t < tb.gw.c) { // (one bit is worth taking since the dense width itself follows the load: witness_table_geometry)
      const MsmGeom g = msm_geometry(z->A.len(), 0, c_t);
      // a narrower digit = more rows per table than build_cache sized (`pending_bytes`, the budget's admission, the device's memory):
      // taken only when the extra bytes fit what the cache budget has left (narrow_room) and the device has them free beside the
      // slices' temporaries — as start_witness_rebuild checks for a re-build (round-5 advisor)
      if (g.tab && g.c == c_t && g.W > tb.gw.W) {
        const uint64_t per_row = (uint64_t)z->A.len() * 64 + (uint64_t)z->B1.len() * 64 + (uint64_t)z->B2.len() * 128 + (uint64_t)z->C.len() * 64;
        const uint64_t extra = (uint64_t)(g.W - tb.gw.W) * per_row;
        size_t free_b = 0, total_b = 0;
        const bool mem_ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)g.W * per_row + (uint64_t)tb.gh.W * z->H.len() * 64 + (1ull << 30) <= free_b;
        (void)hipGetLastError();
        if (mem_ok && extra <= tb.narrow_room.load(std::memory_order_acquire)) {
          tb.gw = g;
          tb.extra_bytes.store(extra, std::memory_order_release);
          narrowed = true;
        }
      } else if (g.tab && g.c == c_t)
        tb.gw = g;
    }
  }
  hipStream_t s = nullptr;
  if (ok) {
    // the lowest stream priority: the build fills what the proves of the key (and of other keys) leave free
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) {
      (void)hipGetLastError();
      ok = false;
    }
  }
  if (trace_tb) fprintf(stderr, "[tables] thread: stream created at %.1f ms\n", ms_since(t0));
  struct Job { const Shard* sh; bool g2; const MsmGeom* g; };
  // H first: the longest of the five builds' G1 arrays; B2 (the G2 array, 60 % of the G1 four together) last
  const Job jobs[5] = {{&z->A, false, &tb.gw}, {&z->B1, false, &tb.gw}, {&z->B2, true, &tb.gw}, {&z->C, false, &tb.gw}, {&z->H, false, &tb.gh}};
  // (a narrowed first build that fails — out of memory after all — is tried once more with the dense geometry the key was admitted with)
  for (int attempt = 0; attempt < 2; attempt++) {
    for (int k : {4, 0, 1, 3, 2}) {
      if (k == 4 && tb.witness_only) continue; // (a re-build for another witness density: row 0 of the current tables is the source)
      if (k == 4 && tb.fresh[4]) continue;     // (second attempt: H's table is there already)
      if (!ok || tb.cancel.load()) {
        ok = false;
        break;
      }
      const Job& j = jobs[k];
      // the bases are in the internal encoding already (form 2); the proves of the key only read them
      const eIcicleError e = j.g2 ? msm_g2_build_table_sliced(j.sh->d_points, j.sh->len(), 2, *j.g, s, &tb.fresh[k], &tb.cancel)
                                  : msm_g1_build_table_sliced(j.sh->d_points, j.sh->len(), 2, *j.g, s, &tb.fresh[k], &tb.cancel);
      if (e != ICICLE_SUCCESS) {
        (void)hipGetLastError();
        ok = false;
        if (getenv("ICICLE_SNARK_VERBOSE") && !tb.cancel.load()) fprintf(stderr, "[icicle-snark-hip] deferred tables: build failed (%s)%s\n", icicle_snark_last_error(), narrowed && attempt == 0 ? "; retrying with the dense digit width" : "; the key keeps the classic layout");
      }
    }
    if (ok || !narrowed || attempt == 1 || tb.cancel.load() || !s) break;
    for (int k : {0, 1, 2, 3}) {
      if (tb.fresh[k]) (void)hipFree(tb.fresh[k]);
      tb.fresh[k] = nullptr;
    }
    tb.gw = gw_dense;
    tb.extra_bytes.store(0, std::memory_order_release);
    ok = true;
  }
  if (trace_tb) fprintf(stderr, "[tables] thread: arrays built at %.1f ms\n", ms_since(t0));
  if (s) (void)hipStreamDestroy(s);
  if (trace_tb) fprintf(stderr, "[tables] thread: stream destroyed at %.1f ms\n", ms_since(t0));
  if (!ok)
    for (void*& t : tb.fresh) {
      if (t) (void)hipFree(t);
      t = nullptr;
    }
  tb.build_ms = ms_since(t0);
  if (getenv("ICICLE_SNARK_TRACE_COLD")) fprintf(stderr, "[cold] deferred tables %s after %.1f ms\n", ok ? "complete" : "abandoned", tb.build_ms);
  tb.state.store(ok ? 2 : 3, std::memory_order_release);
}
} // namespace

int adopt_tables(ZKeyCache* z, bool wait)
{
  TableBuild& tb = z->tb;
  int st = tb.state.load(std::memory_order_acquire);
  if (st == 0) return 1;
  if (st == 1) {
    if (!wait) return 0;
    if (tb.th.joinable()) tb.th.join();
    st = tb.state.load(std::memory_order_acquire);
  }
  if (tb.th.joinable()) tb.th.join();
  if (st == 2) {
    // all five at once: pointers and both geometries change together, so no sort of one geometry ever meets tables of another.
    // (synchronising frees: the caller holds the manager's mutex, nothing of this key is in flight and the build has ended)
    Shard* sh5[5] = {&z->A, &z->B1, &z->B2, &z->C, &z->H};
    for (int k = 0; k < 5; k++) {
      if (k == 4 && tb.witness_only) continue;
      const MsmGeom& g = k == 4 ? tb.gh : tb.gw;
      const int64_t w_old = tb.witness_only ? z->geom_w.W : 1; // (tables of another width replace tables; the first build replaces plain arrays)
      (void)hipFree(sh5[k]->d_points);
      sh5[k]->d_points = tb.fresh[k];
      tb.fresh[k] = nullptr;
      z->device_bytes += (int64_t)sh5[k]->len() * ((int64_t)g.W - w_old) * (k == 2 ? 128 : 64);
    }
    z->geom_w = tb.gw;
    if (!tb.witness_only) {
      z->geom_h = tb.gh;
      z->geom_w_default_c = tb.dense_c; // (the witness tables may have been built narrower already: see table_build_thread)
    }
    // what the classic proves measured (entries of 16-bit digits) says nothing about the table digits: the witness-driven
    // digit width (rebuild_witness_tables) starts over from the first table-mode prove
    z->witness_entries = 0;
    z->proves_since_rebuild = 0;
  }
  // the first build's tables were counted in device_bytes from the load on (cache budget); the loop above has added what was really
  // built, an abandoned build (st == 3) has added nothing
  z->device_bytes -= tb.pending_bytes;
  tb.pending_bytes = 0;
  tb.extra_bytes.store(0, std::memory_order_release); // (device_bytes is exact again)
  tb.witness_only = false;
  tb.state.store(0, std::memory_order_release);
  return 1;
}

void start_witness_rebuild(ZKeyCache* z, int c_new)
{
  TableBuild& tb = z->tb;
  if (!z->geom_w.tab || c_new == z->geom_w.c || tb.state.load(std::memory_order_acquire) != 0) return;
  if (tb.th.joinable()) tb.th.join();
  const MsmGeom g = c_new == z->geom_w_default_c ? witness_table_geometry(z->A.len()) : msm_geometry(z->A.len(), 0, c_new);
  if (!g.tab || (c_new != z->geom_w_default_c && g.c != c_new)) return; // the entry encoding does not fit this width: keep what there is
  // memory: the four new tables beside the old ones + a slice's temporaries; when the device cannot hold that the key keeps its width
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  const uint64_t need = (uint64_t)g.W * ((uint64_t)z->A.len() * 64 + (uint64_t)z->B1.len() * 64 + (uint64_t)z->B2.len() * 128 + (uint64_t)z->C.len() * 64) + (1ull << 30);
  if (need > free_b) return;
  tb.gw = g;
  tb.gh = z->geom_h;
  tb.witness_only = true;
  tb.cancel.store(false);
  tb.go.store(true);
  tb.state.store(1, std::memory_order_release);
  try {
    tb.th = std::thread(table_build_thread, z);
  } catch (...) {
    tb.witness_only = false;
    tb.state.store(0, std::memory_order_release); // no thread to be had: the key keeps its width
  }
}

// ---- cold pipeline (prover_internal.h: ColdFeed / ColdUpload) ------------------------------------------------------------------
namespace {
struct ColdPlan { // what the uploader task works through; owned by the task (heap), deleted when it ends
  ColdUpload* cu;
  ZKeyCache* z;
  int device_id;
  uint32_t* d_records;
  size_t rec_bytes;
  const uint8_t* rec_src;
  uint32_t n_coef;
  UploadJob sec[5]; // A, B1, B2, C, H
};
void cold_upload_task(ColdPlan* pl)
{
  std::unique_ptr<ColdPlan> own(pl);
  ColdUpload* cu = pl->cu;
  ColdFeed& F = cu->feed;
  ZKeyCache* z = pl->z;
  const bool trace = getenv("ICICLE_SNARK_TRACE_COLD") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto bail = [&](int code, const char* fmt, const char* detail) {
    char buf[256];
    snprintf(buf, sizeof buf, fmt, detail);
    F.fail_with(code, buf);
    if (pl->d_records) (void)hipFree(pl->d_records);
    F.finish();
  };
  if (hipSetDevice(pl->device_id) != hipSuccess) return bail((int)ICICLE_INVALID_DEVICE, "cold upload: %s", "hipSetDevice");
  hipStream_t su = cu->lanes[0]; // kernels and events of the stages; a stage's data has LANDED when staged_copy returns
  auto upload = [&](const UploadJob& j, bool from_wtns) -> bool {
    if (!j.n) return true;
    if (from_wtns) staged_copy_file_hint(cu->wtns_base, cu->wtns_len, cu->wtns_fd);
    else staged_copy_file_hint(cu->zkey_base, cu->zkey_len, cu->zkey_fd);
    // two workers, one per lane: more of them (4, 6, 8 over the same two streams) were measured and are SLOWER beside a running
    // prove — 31 → 34–53 ms process-warm, 55 → 69–80 ms for the first key of a process (profiles/r05_cold_path.txt)
    const hipError_t e = staged_copy(pl->device_id, &j, 1, true, cu->lanes, 2, false, nullptr);
    staged_copy_file_hint(nullptr, 0, -1);
    if (e != hipSuccess) {
      bail((int)ICICLE_COPY_FAILED, "cold upload: %s", hipGetErrorString(e));
      return false;
    }
    return true;
  };
  auto stage_done = [&](int i) -> bool {
    if (hipEventRecord(F.ev[i], su) != hipSuccess) {
      bail((int)ICICLE_UNKNOWN_ERROR, "cold upload: %s", "hipEventRecord");
      return false;
    }
    F.post(i);
    if (trace) fprintf(stderr, "[cold] stage %d posted at    %8.2f ms\n", i, ms_since(t0));
    return true;
  };
  // coefficients → CSR (the range check of the records is part of it: a key that fails it never reaches the front end)
  if (!upload({pl->d_records, pl->rec_src, pl->rec_bytes}, false)) return;
  if (trace) fprintf(stderr, "[cold] records uploaded at      %8.2f ms\n", ms_since(t0));
  {
    uint32_t first_bad = 0;
    const hipError_t e = qap_build_csr(pl->d_records, pl->n_coef, z->domain_size, z->n_vars, z->d_rowptr, z->d_cols, z->d_vals, &first_bad, su);
    if (e != hipSuccess) return bail((int)ICICLE_UNKNOWN_ERROR, "cold upload: CSR build: %s", hipGetErrorString(e));
    if (first_bad != 0xffffffffu) {
      char num[32];
      snprintf(num, sizeof num, "%u", first_bad);
      return bail(ERR_FORMAT, "zkey: coefficient %s out of range", num);
    }
    // (the records are freed at the END of the task: a hipFree of 141 MB in front of the witness upload made the next
    // hipMemcpyAsync of a process's first cold prove wait 8–15 ms for the runtime)
  }
  if (!stage_done(ColdFeed::COEF)) return;
  // the witness of the prove that is waiting for all this
  if (!upload({z->d_witness, cu->wtns_values, cu->wtns_bytes}, true)) return;
  if (!stage_done(ColdFeed::WITNESS)) return;
  // the point sections in the order the prove enqueues their MSMs (B2 — the longest chain — first), each converted in place from the
  // file's Montgomery form to the bucket kernels' encoding as it lands
  const int order[5] = {2, 0, 1, 3, 4};
  Shard* sh5[5] = {&z->A, &z->B1, &z->B2, &z->C, &z->H};
  for (int k : order) {
    if (!upload(pl->sec[k], false)) return;
    const eIcicleError e = k == 2 ? msm_g2_points_to_internal(sh5[k]->d_points, sh5[k]->len(), 1, su) : msm_g1_points_to_internal(sh5[k]->d_points, sh5[k]->len(), 1, su);
    if (e != ICICLE_SUCCESS) return bail((int)e, "cold upload: %s", "points to internal form");
    if (!stage_done(ColdFeed::SEC_A + k)) return;
  }
  if (hipStreamSynchronize(su) != hipSuccess) return bail((int)ICICLE_SYNCHRONIZATION_FAILED, "cold upload: %s", "hipStreamSynchronize");
  if (trace) fprintf(stderr, "[cold] upload task done after %8.2f ms\n", ms_since(t0));
  (void)hipFree(pl->d_records);
  pl->d_records = nullptr;
  F.finish();
}
} // namespace

void cold_upload_wait(ColdUpload* cu)
{
  if (!cu) return;
  if (cu->started && cu->task.queued) WorkerPool::wait(&cu->task);
  cu->started = false;
  for (hipStream_t& st : cu->lanes) {
    if (st) (void)icicle_destroy_stream(st); // (drained; back to the pool)
    st = nullptr;
  }
  for (hipEvent_t& e : cu->feed.ev) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
}

int build_cache(const uint8_t* data, size_t len, int device_id, int rank, int count, std::unique_ptr<ZKeyCache>& out, bool defer_tables, ColdUpload* cold)
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
  // six streams; the library asks the runtime for eight hardware queues so that they do not share one (runtime.cpp).
  // Stream priorities were tried (QAP chain high, G2 low, …): every variant was 1-2 ms slower than equal priorities.
  // Created BEFORE the ingest, which uses them as its lanes: a fresh stream costs ≈ 4 ms to create and its first host→device
  // copy sets up its DMA queue (another 3 ms) — the eight short-lived upload streams of rounds 1–4 paid both on every cold
  // load (≈ 35 ms of a 57 ms upload at 1.6 M constraints), and the key's own streams then paid them again.
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_qap)); // QAP front end (its own hardware queue; a higher stream priority made no difference)
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g1));
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g2));
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g3));
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g4));
  P_ICICLE(icicle_create_stream((icicleStreamHandle*)&z->s_g5));
  assign_stream_roles(z.get(), /*allow_measure=*/count <= 1);
  lap("six streams");
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
  // The range of shard `rank` is the witness SLICE that rank uploads itself (witness_slice_elems: ⌈n_vars / count⌉ wires from
  // rank·slice; groth16_upload_witness_slice, multi.cpp) whenever that leaves no shard empty: its digit sort and its four witness
  // accumulations then need nothing from the other devices and run while the in-place all-gather, the distributed front end and its
  // two all-to-alls are still under way (prover.cpp: own_slice_first).  Otherwise the even split ⌊n_vars·rank / count⌋.
  const uint64_t slice = witness_slice_elems(z->n_vars, count);
  z->slice_aligned = count > 1 && slice * (uint64_t)(count - 1) < z->n_vars;
  const uint32_t wlo = z->slice_aligned ? (uint32_t)(slice * (uint64_t)rank) : (uint32_t)((uint64_t)z->n_vars * rank / count);
  const uint32_t whi = z->slice_aligned ? (uint32_t)std::min<uint64_t>(z->n_vars, slice * (uint64_t)(rank + 1)) : (uint32_t)((uint64_t)z->n_vars * (rank + 1) / count);
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
  // bases: the file's Montgomery form (R = 2^256) → the bucket kernels' internal encoding (R' = 2^261), once.  Table mode
  // (msm_plan.h; ICICLE_SNARK_TABLES=0 disables it): every base array becomes W rows 2^(c·w)·P so that all digits of a
  // scalar share one bucket set — 13 instead of 16 mixed additions per scalar at 1.6 M constraints for 13× the base memory.
  // (decided BEFORE the upload since round 5: the cold pipeline below only applies when no table has to be built in here)
  const int tables_env = getenv("ICICLE_SNARK_TABLES") ? atoi(getenv("ICICLE_SNARK_TABLES")) : 1;
  bool tables = tables_env != 0;
  // Above 2^22 points the 32-bit sort entry has no room for 20-bit digits beside the point index (msm_sort.hip: tab_low_bits):
  // the tables would fall back to c = 19 / 14 digits, and measured at 6.4 M constraints (domain 2^23) that is no faster than the
  // classic layout with c = 16 / 16 digits — 65.4 against 64.3 ms resident — for 37.7 instead of 4.4 GB of device memory and a
  // second of table build (tests/test_gpu_fullsize.py::test_prove_at_domain_2p23_…): such keys stay classic
  // (ICICLE_SNARK_TABLES=2 builds the tables all the same).
  if (tables_env < 2 && (z->A.len() > (1u << 22) || z->H.len() > (1u << 22))) tables = false;
  z->geom_w = tables ? witness_table_geometry(z->A.len()) : msm_geometry(z->A.len(), 0, 0);
  z->geom_h = msm_geometry(z->H.len(), 0, tables ? 1 : 0);
  if (tables) {
    // the tables need W× the base memory plus the temporaries of the largest build (projective rows + inversion
    // scratch of the G2 set); keep the classic layout when the device cannot hold them next to what is already there
    size_t free_b = 0, total_b = 0;
    release_cached_device_memory(); // blocks parked by icicle_free count as free
    P_HIP(hipMemGetInfo(&free_b, &total_b));
    const uint64_t ww = (uint64_t)z->geom_w.W, wh = (uint64_t)z->geom_h.W, wb = ww;
    const uint64_t need = ww * ((uint64_t)z->A.len() * 64 + (uint64_t)z->C.len() * 64) + wb * (uint64_t)z->B1.len() * (64 + 128) + wh * (uint64_t)z->H.len() * 64 +
                          wb * (uint64_t)z->B2.len() * (192 + 64) + ((uint64_t)n * 128 + (uint64_t)z->n_vars * 32 + (64u << 20));
    if (need > free_b) {
      tables = false;
      z->geom_w = msm_geometry(z->A.len(), 0, 0);
      z->geom_h = msm_geometry(z->H.len(), 0, 0);
    }
  }
  const bool defer_env = !(getenv("ICICLE_SNARK_DEFER_TABLES") && atoi(getenv("ICICLE_SNARK_DEFER_TABLES")) == 0);
  const bool defer = tables && defer_tables && defer_env && count == 1;
  // cold pipeline: the caller's prove starts while the sections are still on their way (nothing in here needs their contents then)
  const bool pipeline = cold != nullptr && count == 1 && !h_strided && (defer || !tables);
  if (!pipeline) {
    {
      const hipStream_t lanes[6] = {z->s_qap, z->s_g1, z->s_g2, z->s_g3, z->s_g4, z->s_g5};
      if (int rc = staged_upload(device_id, jobs, lanes, 6)) return rc;
    }
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
  }
  {
    if (defer) {
      // the key proves in the classic layout until the worker thread (started at the end of this function) has built the
      // tables of these geometries; adopt_tables swaps them in
      z->tb.gw = z->geom_w;
      z->tb.gh = z->geom_h;
      z->tb.dense_c = z->geom_w.c;
      // the tables count towards the entry's size from now on (the cache budget admits and evicts keys by device_bytes: a key must not
      // look small while its tables are still being built)
      z->tb.pending_bytes = (uint64_t)(z->geom_w.W - 1) * ((uint64_t)z->A.len() * 64 + (uint64_t)z->B1.len() * 64 + (uint64_t)z->B2.len() * 128 + (uint64_t)z->C.len() * 64) +
                            (uint64_t)(z->geom_h.W - 1) * (uint64_t)z->H.len() * 64;
      z->device_bytes += z->tb.pending_bytes;
      z->geom_w = msm_geometry(z->A.len(), 0, 0);
      z->geom_h = msm_geometry(z->H.len(), 0, 0);
      z->tb.state.store(1);
    }
    struct Job { Shard* sh; bool g2; const MsmGeom* g; };
    const Job jobs5[5] = {{&z->A, false, &z->geom_w}, {&z->B1, false, &z->geom_w}, {&z->B2, true, &z->geom_w}, {&z->C, false, &z->geom_w}, {&z->H, false, &z->geom_h}};
    for (const Job& j : jobs5) {
      if (pipeline) break; // (the uploader task converts every section as it lands)
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
  P_HIP(hipHostMalloc((void**)&z->h_partials, 5 * PARTIALS_STRIDE + 64)); // + the sort statistics read back per prove (h_stats)
  z->h_stats = reinterpret_cast<uint32_t*>(z->h_partials + 5 * PARTIALS_STRIDE);
  z->h_stats[0] = z->h_stats[1] = 0;
  z->geom_w_default_c = z->geom_w.c;
  z->device_bytes += (size_t)z->n_vars * 32 + (size_t)n * 96;
  {
    // the first host→device copy on a stream sets up its DMA queue (milliseconds, measured 20 ms over six streams): the
    // ingest above has done that for every stream it used as a lane; a key of a few chunks leaves some untouched
    const hipStream_t all[6] = {z->s_qap, z->s_g1, z->s_g2, z->s_g3, z->s_g4, z->s_g5};
    for (hipStream_t st : all) P_HIP(hipMemcpyAsync(z->d_partials, z->h_partials, 4096, hipMemcpyHostToDevice, st));
    for (hipStream_t st : all) P_HIP(hipStreamSynchronize(st));
  }
  P_HIP(hipEventCreateWithFlags(&z->ev_witness, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_sort, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_sort_h, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_own_slice, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_head_in, hipEventDisableTiming));
  P_HIP(hipEventCreateWithFlags(&z->ev_head_done, hipEventDisableTiming));
  P_HIP(hipEventCreate(&z->ev_t_head_start));
  P_HIP(hipEventCreate(&z->ev_t_head_end));
  P_HIP(hipEventCreate(&z->ev_t_witness));
  for (auto& e : z->ev) P_HIP(hipEventCreate(&e));
  for (auto& e : z->ev_done) P_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : z->ev_lfork) P_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : z->ev_ljoin) P_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  lap("work buffers, events");
  out = std::move(z);
  if (pipeline) {
    // the uploader task (a pooled worker; inline when none can be had: then everything has arrived when this returns)
    ZKeyCache* zz = out.get();
    for (auto& e : cold->feed.ev) P_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& st : cold->lanes) P_ICICLE(icicle_create_stream((icicleStreamHandle*)&st));
    ColdPlan* pl = new ColdPlan();
    pl->cu = cold;
    pl->z = zz;
    pl->device_id = device_id;
    pl->d_records = d_records;
    pl->rec_bytes = rec_bytes;
    pl->rec_src = s4->p + 4;
    pl->n_coef = n_coef;
    const Section* psec[5] = {s5, s6, s7, s8, s9};
    Shard* sh5[5] = {&zz->A, &zz->B1, &zz->B2, &zz->C, &zz->H};
    for (int k = 0; k < 5; k++) {
      const size_t esz = k == 2 ? 128 : 64;
      pl->sec[k] = {sh5[k]->d_points, psec[k]->p + (size_t)sh5[k]->lo * esz, (size_t)sh5[k]->len() * esz};
    }
    free_records.p = nullptr; // the records belong to the task now (freed behind the CSR build)
    zz->feed = &cold->feed;
    zz->tb.hold.store(true, std::memory_order_release); // the deferred table build reads the base arrays: not before they are complete
    cold->task.fn = [pl] { cold_upload_task(pl); };
    cold->started = true;
    WorkerPool::get().run_or_inline(&cold->task);
    lap("cold upload task started");
  }
  // (the entry does not move any more: the worker keeps a pointer to it and ~ZKeyCache joins the worker)
  if (out->tb.state.load() == 1) {
    try {
      out->tb.th = std::thread(table_build_thread, out.get());
    } catch (...) {
      // no thread to be had: the key keeps the classic layout — and gives back what was counted for tables that will not come
      out->tb.state.store(0);
      out->device_bytes -= out->tb.pending_bytes;
      out->tb.pending_bytes = 0;
      out->tb.hold.store(false, std::memory_order_release);
    }
  }
  return 0;
}

// ---- digit width of the witness MSMs adapted to the witnesses a key really sees ------------------------------------------------
// The tables fix the digit width c at cache build for DENSE scalars: c = 20 at a million wires, 2^19 buckets per MSM, whose
// reduction is 1 M point additions per MSM whatever the witness.  Witnesses of real circuits are mostly 0/1 wires and small values:
// a tenth to a fifth of the non-zero digits of a dense witness, so the four witness MSMs (A, B1, B2, C) spend more in reducing
// empty-ish buckets than in filling them — the G2 reduction heads the critical chain (HISTORY.md §9-2a).  After a prove the
// entry count of the witness digit sort is known; if it calls for a narrower digit (target ≈ 32 entries per bucket; one bit is
// enough from the dense width, two bits afterwards: prover.cpp, follow_witness), the four tables are rebuilt from their own row 0 with that width — once, ≈ 0.1–0.3 s, like a cache build — and the
// next proves sort with it and size the large-bucket threshold from the observed count.  Measured on the stand-in keys of
// BASELINE configs 4 / 5: c = 20 → 18, prove 5.5 → 4.7 ms (1.0 M constraints) and 8.15 → 7.3 ms (1.4 M); c ≤ 16 is slower again
// (one thread per bucket: chains of hundreds).  A later, denser witness moves the key back the same way.  H stays dense.
int witness_digit_target(const ZKeyCache* z, uint64_t entries)
{
  if (!entries) return z->geom_w.c;
  int lg = 0;
  while (((uint64_t)1 << lg) * 32 < entries) lg++; // 2^lg ≥ entries / 32 buckets
  int c = lg + 1;
  if (c > z->geom_w_default_c) c = z->geom_w_default_c;
  if (c < 13) c = 13;
  return c;
}
int rebuild_witness_tables(ZKeyCache* z, int c_new)
{
  if (!z->geom_w.tab || c_new == z->geom_w.c) return 0;
  static std::mutex rebuild_mu; // the shards of a device group may share a device (and its null stream)
  std::lock_guard<std::mutex> lk(rebuild_mu);
  const MsmGeom g = c_new == z->geom_w_default_c ? witness_table_geometry(z->A.len()) : msm_geometry(z->A.len(), 0, c_new);
  if (!g.tab || (c_new != z->geom_w_default_c && g.c != c_new)) return 0; // the entry encoding does not fit this width: keep what there is
  // ALL four new tables are built next to the old ones, then pointers and geometry are swapped together: a failure at any
  // point (allocation, launch) frees what was built and leaves the key exactly as it was — the old tables with the old
  // geom_w — so shard_commitments never indexes tables of one geometry with digits of another (round-3 advisor finding:
  // the one-by-one swap left A and B1 in the new geometry when B2's build failed).  Memory: the four new tables plus the
  // temporaries of the largest build (projective rows + inversion scratch of the G2 set); when the device cannot hold that
  // next to the old tables the key simply keeps its width.
  struct Job { Shard* sh; bool g2; };
  const Job jobs[4] = {{&z->A, false}, {&z->B1, false}, {&z->B2, true}, {&z->C, false}};
  size_t free_b = 0, total_b = 0;
  release_cached_device_memory();
  P_HIP(hipMemGetInfo(&free_b, &total_b));
  uint64_t need = (uint64_t)z->B2.len() * g.W * (192 + 64) + (64u << 20);
  for (const Job& j : jobs) need += (uint64_t)j.sh->len() * g.W * (j.g2 ? 128 : 64);
  if (need > free_b) return 0;
  void* fresh[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < 4; k++) {
    const Job& j = jobs[k];
    // row 0 of the old table = the bases themselves, in the internal encoding (form 2)
    const eIcicleError e = j.g2 ? msm_g2_build_table(j.sh->d_points, j.sh->len(), 2, g, nullptr, &fresh[k]) : msm_g1_build_table(j.sh->d_points, j.sh->len(), 2, g, nullptr, &fresh[k]);
    if (e != ICICLE_SUCCESS) {
      (void)hipGetLastError();
      for (void* t : fresh)
        if (t) (void)hipFree(t);
      // not an error of the prove: the key keeps its tables and its width (out of memory is the expected cause)
      if (getenv("ICICLE_SNARK_VERBOSE")) fprintf(stderr, "[icicle-snark-hip] witness tables keep c = %d (rebuild for c = %d failed: %s)\n", z->geom_w.c, c_new, icicle_snark_last_error());
      return 0;
    }
  }
  for (int k = 0; k < 4; k++) {
    const Job& j = jobs[k];
    (void)hipFree(j.sh->d_points); // (a synchronising free: nothing of a prove is in flight here)
    j.sh->d_points = fresh[k];
    z->device_bytes += (int64_t)j.sh->len() * ((int64_t)g.W - (int64_t)z->geom_w.W) * (j.g2 ? 128 : 64);
  }
  z->geom_w = g;
  return 0;
}

} // namespace prover
} // namespace isnark
