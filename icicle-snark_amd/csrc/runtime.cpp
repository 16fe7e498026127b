// runtime.cpp — device / stream / memory C API (replaces icicle/src/runtime.cpp + the CUDA device
// API icicle/backend/cuda/src/cuda_device_api.cu for device type "HIP").
//
// * one registered device type, "HIP" (alias "CUDA"); the active device is thread-local with a
//   process-wide default (icicle/src/device_api.cpp:87-116);
// * streams are hipStream_t (non-blocking); allocations are plain hipMalloc, kernel temporaries come
//   from a per-stream workspace arena (common.h);
// * every allocation made through this API is recorded so that icicle_is_active_device_memory()
//   also answers for interior pointers (icicle/include/icicle/memory_tracker.h:11-55) — the Rust
//   DeviceSlice checks this on every slice (wrappers/rust/icicle-runtime/src/memory.rs:120-125).
#include <atomic>
#include <map>
#include <thread>
#include <execinfo.h>
#include <mutex>
#include <signal.h>
#include <vector>
#include <stdarg.h>
#include <string.h>
#include <unistd.h>
#include <chrono>

#include "common.h"
#include "msm_plan.h"
#include "workers.h"

// The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share
// a queue serialise.  The prover overlaps five streams (HISTORY.md §4); with four queues two of them collide and a
// benchmark/1600k prove measured 28.5 ms instead of 25.5 ms.  Ask for twelve unless the user has set the knob — a key's six
// streams, the two lanes of a cold upload (kept in the stream pool afterwards), the staging engine's lane and a table build's
// stream are ten; with eight queues the two pooled lanes made two of a key's streams share one (stand-in of 1.0 M constraints:
// 6.3 instead of 5.7 ms per prove; 1600k: no difference between 8, 12 and 16).  This runs when the library is loaded, before its
// first HIP call initialises the runtime.
// ICICLE_SNARK_BACKTRACE=1: print the native stack of a crashing thread (the HIP runtime's callback threads carry no
// Python frames) before the default action takes over
static void isnark_segv(int sig)
{
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void isnark_runtime_env()
{
  setenv("GPU_MAX_HW_QUEUES", "12", 0);
  if (getenv("ICICLE_SNARK_BACKTRACE")) {
    signal(SIGSEGV, isnark_segv);
    signal(SIGBUS, isnark_segv);
  }
}

namespace isnark {

static thread_local char g_err[512] = "";
void set_last_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  if (getenv("ICICLE_SNARK_VERBOSE")) fprintf(stderr, "[icicle-snark-hip] %s\n", g_err);
}

static std::mutex g_mu;
static std::map<uintptr_t, std::pair<size_t, int>> g_allocs; // base -> (size, device id)
static int g_default_device = -1;                             // -1: none chosen yet
static thread_local int t_device = -1;

static bool type_ok(const char* t) { return strncmp(t, "HIP", 64) == 0 || strncmp(t, "CUDA", 64) == 0; }

static int active_device()
{
  if (t_device >= 0) return t_device;
  return g_default_device;
}

eIcicleError require_device()
{
  int d = active_device();
  if (d < 0) {
    // the reference falls back to a default device on a fresh thread (device_api.cpp:104-116)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
      set_last_error("no HIP device available (this library has no CPU fallback)");
      return ICICLE_INVALID_DEVICE;
    }
    d = 0;
    g_default_device = 0;
  }
  if (t_device != d) {
    HIP_TRY(hipSetDevice(d), ICICLE_INVALID_DEVICE);
    t_device = d;
  }
  return ICICLE_SUCCESS;
}

static void track(void* p, size_t size)
{
  std::lock_guard<std::mutex> lk(g_mu);
  g_allocs[(uintptr_t)p] = {size, t_device};
}
static bool untrack(void* p)
{
  std::lock_guard<std::mutex> lk(g_mu);
  return g_allocs.erase((uintptr_t)p) > 0;
}
// returns device id owning ptr or -1
static int identify(const void* p)
{
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_allocs.upper_bound((uintptr_t)p);
  if (it == g_allocs.begin()) return -1;
  --it;
  if ((uintptr_t)p < it->first + it->second.first) return it->second.second;
  return -1;
}

bool is_tracked_device_ptr(const void* p) { return identify(p) >= 0; }

// ---- pageable-memory copy engine (see common.h) ------------------------------------------------------------------
constexpr size_t STAGED_CHUNK = STAGED_CHUNK_BYTES;
struct StagedPool { // pinned staging, events and the lane stream of ONE device (an event only records on a stream of the device
  std::mutex mu;     // it was created on; one pool and one lock per device also lets the prover threads of several GPUs upload at once)
  uint8_t* pinned = nullptr;
  hipEvent_t events[STAGED_LANES][2] = {};
  hipEvent_t head_ev[STAGED_LANES] = {}; // StagedProgress: behind a lane's last chunk of the head
  std::vector<hipStream_t> lanes; // persistent lane stream(s)
};
constexpr int STAGED_DEVICES = 16;
static StagedPool g_staged[STAGED_DEVICES];

bool is_pinned_host(const void* host_ptr, int device_id)
{
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, host_ptr) != hipSuccess) {
    (void)hipGetLastError(); // an unregistered (pageable) pointer is reported as an error: clear it
    return false;
  }
  if (a.type != hipMemoryTypeHost) return false;
  // device_id ≥ 0: the DMA will be issued by that device — take the direct path only for memory pinned in its context or
  // pinned portably (hipHostMallocPortable / hipHostRegisterPortable); anything else goes through the staging copy
  if (device_id >= 0 && a.device != device_id && !(a.allocationFlags & hipHostMallocPortable)) return false;
  return true;
}

// A caller whose pageable source is a mapped FILE can say so: the workers then pread() the chunk straight into their pinned
// buffer instead of copying it out of the mapping (no page faults on the mapping; same number of copies)
static thread_local const uint8_t* t_file_base = nullptr;
static thread_local size_t t_file_len = 0;
static thread_local int t_file_fd = -1;
void staged_copy_file_hint(const void* base, size_t len, int fd)
{
  t_file_base = (const uint8_t*)base;
  t_file_len = len;
  t_file_fd = fd;
}
void staged_copy_file_hint_get(const void** base, size_t* len, int* fd)
{
  *base = t_file_base;
  *len = t_file_len;
  *fd = t_file_fd;
}

// pinned staging buffers and events of a device's pool (once; the caller holds P.mu and has made the device current)
static hipError_t staged_pool_init(StagedPool& P)
{
  if (P.pinned) return hipSuccess;
  hipError_t e0;
  uint8_t* pinned = nullptr;
  if ((e0 = hipHostMalloc((void**)&pinned, STAGED_LANES * 2 * STAGED_CHUNK, hipHostMallocPortable)) != hipSuccess) return e0;
  for (int t = 0; t < STAGED_LANES; t++)
    for (int k = 0; k < 2; k++)
      if (!P.events[t][k] && (e0 = hipEventCreateWithFlags(&P.events[t][k], hipEventDisableTiming)) != hipSuccess) return e0;
  for (int t = 0; t < STAGED_LANES; t++)
    if (!P.head_ev[t] && (e0 = hipEventCreateWithFlags(&P.head_ev[t], hipEventDisableTiming)) != hipSuccess) return e0;
  P.pinned = pinned;
  return hipSuccess;
}

hipError_t staged_copy(int device_id, const CopyJob* jobs, size_t njobs, bool to_device, const hipStream_t* lanes_in, int n_lanes, bool own_temp_streams, StagedProgress* progress)
{
  const uint8_t* const file_base = t_file_base;
  const size_t file_len = t_file_len;
  const int file_fd = t_file_fd;
  if (device_id < 0 || device_id >= STAGED_DEVICES) return hipErrorInvalidDevice;
  StagedPool& P = g_staged[device_id];
  std::lock_guard<std::mutex> lk(P.mu);
  hipError_t e0 = hipSetDevice(device_id);
  if (e0 != hipSuccess) return e0;
  if ((e0 = staged_pool_init(P)) != hipSuccess) return e0;
  // chunk size: 2 MB.  Smaller chunks for transfers of a few tens of MB (shorter pipeline fill) were measured on the 51 MB
  // witness of benchmark/1600k and are slower — 1 MB: +0.05 ms, 512 KB: +0.3 ms, 256 KB: +0.8 ms per prove (per-DMA cost);
  // two to five staging lanes make no difference either (17.3–17.5 ms): ≈ 1.3 ms for 51 MB is what this path costs
  const size_t CH = STAGED_CHUNK;
  std::vector<CopyJob> chunks;
  for (size_t j = 0; j < njobs; j++)
    for (size_t off = 0; off < jobs[j].n; off += CH)
      chunks.push_back({(uint8_t*)jobs[j].dst + off, (const uint8_t*)jobs[j].src + off, jobs[j].n - off < CH ? jobs[j].n - off : CH});
  if (chunks.empty()) return hipSuccess;
  hipStream_t streams[STAGED_LANES] = {};
  int max_lanes = STAGED_LANES;
  std::vector<hipStream_t> temp;
  if (lanes_in) {
    max_lanes = n_lanes < STAGED_LANES ? n_lanes : STAGED_LANES;
    for (int t = 0; t < max_lanes; t++) streams[t] = lanes_in[t];
  } else if (own_temp_streams) {
    // cold path: short-lived streams (idle streams would keep hardware-queue slots the prover's own streams need)
    for (int t = 0; t < STAGED_LANES; t++) {
      hipStream_t st;
      if ((e0 = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) {
        for (hipStream_t x : temp) (void)hipStreamDestroy(x);
        return e0;
      }
      temp.push_back(st);
      streams[t] = st;
    }
  } else {
    // per-call copies of the C ABI: four workers fill their pinned buffers in parallel and enqueue the DMAs on ONE
    // persistent stream of the engine — every extra idle stream takes one of the GPU_MAX_HW_QUEUES hardware queues away
    // from the prover's own streams (with four lane streams alive, two of the prover's six shared a queue)
    std::vector<hipStream_t>& v = P.lanes;
    max_lanes = 4;
    if (v.empty()) {
      hipStream_t st;
      if ((e0 = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return e0;
      v.push_back(st);
    }
    for (int t = 0; t < max_lanes; t++) streams[t] = v[0];
  }
  std::atomic<size_t> next{0};
  std::atomic<int> err{(int)hipSuccess};
  static const bool trace_slow = getenv("ICICLE_SNARK_TRACE_HOST") != nullptr;
  // chunks of the head (StagedProgress): the first chunks of job 0
  const size_t head_chunks = progress && to_device && njobs ? ((progress->head_bytes < jobs[0].n ? progress->head_bytes : jobs[0].n) + CH - 1) / CH : 0;
  auto worker = [&](int t) {
    bool head_reported = progress == nullptr;
    // behind this lane's last chunk of the head (or with nothing of it): record, then count this lane in
    auto report_head = [&]() {
      if (head_reported) return;
      head_reported = true;
      if (hipEventRecord(P.head_ev[t], streams[t]) != hipSuccess) err = (int)hipErrorUnknown;
      progress->ev[t] = P.head_ev[t];
      progress->lanes_reported.fetch_add(1, std::memory_order_release);
      progress->notify();
    };
    struct ReportOnExit {
      decltype(report_head)& f;
      ~ReportOnExit() { f(); }
    } on_exit{report_head};
    if (hipSetDevice(device_id) != hipSuccess) {
      err = (int)hipErrorInvalidDevice;
      return;
    }
    uint8_t* buf[2] = {P.pinned + (size_t)t * 2 * STAGED_CHUNK, P.pinned + ((size_t)t * 2 + 1) * STAGED_CHUNK};
    if (to_device) {
      bool used[2] = {false, false};
      for (int k = 0;; k ^= 1) {
        const size_t i = next.fetch_add(1);
        if (i >= head_chunks) report_head(); // chunks are handed out in order: this lane has enqueued its last chunk of the head
        if (i >= chunks.size() || err.load() != (int)hipSuccess) break;
        hipError_t e = hipSuccess;
        const auto tA = trace_slow ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        if (used[k]) e = hipEventSynchronize(P.events[t][k]); // the DMA that last read this buffer is done
        const auto tB = trace_slow ? std::chrono::steady_clock::now() : tA;
        if (e == hipSuccess) {
          const uint8_t* src = (const uint8_t*)chunks[i].src;
          bool done = false;
          static const bool use_pread = !(getenv("ICICLE_SNARK_FILE_PREAD") && atoi(getenv("ICICLE_SNARK_FILE_PREAD")) == 0);
          if (use_pread && file_fd >= 0 && src >= file_base && src + chunks[i].n <= file_base + file_len) {
            size_t got = 0;
            while (got < chunks[i].n) {
              const ssize_t r = pread(file_fd, buf[k] + got, chunks[i].n - got, (off_t)(src - file_base + got));
              if (r <= 0) break;
              got += (size_t)r;
            }
            done = got == chunks[i].n;
          }
          if (!done) memcpy(buf[k], src, chunks[i].n);
          const auto tC = trace_slow ? std::chrono::steady_clock::now() : tA;
          e = hipMemcpyAsync(chunks[i].dst, buf[k], chunks[i].n, hipMemcpyHostToDevice, streams[t]);
          if (trace_slow) {
            // ICICLE_SNARK_TRACE_HOST: which step of a chunk took more than 2 ms (buffer wait / read out of the page cache / enqueue)
            const auto tD = std::chrono::steady_clock::now();
            const double w = std::chrono::duration<double, std::milli>(tB - tA).count(), r = std::chrono::duration<double, std::milli>(tC - tB).count(),
                         q = std::chrono::duration<double, std::milli>(tD - tC).count();
            if (w > 2 || r > 2 || q > 2) fprintf(stderr, "[host] staged copy: lane %d chunk %zu of %zu: buffer wait %.2f ms, read %.2f ms, enqueue %.2f ms\n", t, i, chunks.size(), w, r, q);
          }
        }
        if (e == hipSuccess) e = hipEventRecord(P.events[t][k], streams[t]);
        used[k] = true;
        if (e != hipSuccess) {
          err = (int)e;
          break;
        }
      }
    } else {
      // device → host: the DMA of chunk i into one buffer overlaps the memcpy of chunk i − 1 out of the other
      size_t pending = (size_t)-1;
      int pk = 0;
      for (int k = 0;; k ^= 1) {
        const size_t i = next.fetch_add(1);
        const bool have = i < chunks.size() && err.load() == (int)hipSuccess;
        hipError_t e = hipSuccess;
        if (have) {
          e = hipMemcpyAsync(buf[k], chunks[i].src, chunks[i].n, hipMemcpyDeviceToHost, streams[t]);
          if (e == hipSuccess) e = hipEventRecord(P.events[t][k], streams[t]);
        }
        if (pending != (size_t)-1) {
          hipError_t e2 = hipEventSynchronize(P.events[t][pk]);
          if (e2 == hipSuccess) memcpy(chunks[pending].dst, buf[pk], chunks[pending].n);
          else e = e2;
          pending = (size_t)-1;
        }
        if (e != hipSuccess) {
          err = (int)e;
          break;
        }
        if (!have) break;
        pending = i;
        pk = k;
      }
    }
    hipError_t e = hipStreamSynchronize(streams[t]);
    if (e != hipSuccess) err = (int)e;
  };
  const int nt = chunks.size() < (size_t)max_lanes ? (int)chunks.size() : max_lanes;
  if (progress) {
    progress->lanes_total.store(nt, std::memory_order_release);
    progress->notify();
  }
  // lanes 1 … nt − 1 on pooled workers (workers.h; a lane the pool cannot take runs here, after lane 0), lane 0 on this thread
  HostTask tasks[STAGED_LANES];
  for (int t = 1; t < nt; t++) {
    tasks[t].fn = [&worker, t] { worker(t); };
    if (!WorkerPool::get().submit(&tasks[t])) tasks[t].queued = false;
  }
  if (nt > 0) worker(0);
  for (int t = 1; t < nt; t++) {
    if (tasks[t].queued) WorkerPool::wait(&tasks[t]);
    else worker(t);
  }
  for (hipStream_t x : temp) (void)hipStreamDestroy(x);
  return (hipError_t)err.load();
}

void release_cached_device_memory(); // below: the allocation cache of icicle_malloc / icicle_free
void ws_release_idle(int dev, bool lock = true);
} // namespace isnark
// stream pool and allocation cache (defined with the C API below), needed by the workspace arena's out-of-memory path
static std::mutex g_sp_mu;
static std::map<int, std::vector<hipStream_t>> g_stream_pool;
static std::mutex g_ac_mu;
static void alloc_cache_flush(int dev);
namespace isnark {

// ---- workspace arena (see common.h) -------------------------------------------------------------
struct WsBlock {
  void* ptr;
  size_t size;
  bool in_use;
};
static std::mutex g_ws_mu;
// keyed by (device, stream): the null stream exists on every device
typedef std::pair<int, hipStream_t> WsKey;
static std::map<WsKey, std::vector<WsBlock>> g_ws;
constexpr size_t WS_ORPHAN_MAX = (size_t)8 << 30;
static std::map<int, std::vector<WsBlock>> g_ws_orphans; // device → idle blocks whose stream was destroyed (ws_release_stream)
static WsKey ws_key(hipStream_t s)
{
  int d = 0;
  (void)hipGetDevice(&d);
  return WsKey(d, s);
}

hipError_t ws_alloc(void** p, size_t bytes, hipStream_t s)
{
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~(size_t)255;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  auto& v = g_ws[ws_key(s)];
  int best = -1;
  for (size_t i = 0; i < v.size(); i++)
    if (!v[i].in_use && v[i].size >= bytes && (best < 0 || v[i].size < v[best].size)) best = (int)i;
  // do not hand a huge block to a tiny request: keep fragmentation bounded
  if (best >= 0 && v[best].size <= 4 * bytes + (1 << 20)) {
    v[best].in_use = true;
    *p = v[best].ptr;
    return hipSuccess;
  }
  {
    // a block left behind by a destroyed stream of this device (idle: that stream was synchronised)
    std::vector<WsBlock>& orphans = g_ws_orphans[ws_key(s).first];
    int ob = -1;
    for (size_t i = 0; i < orphans.size(); i++)
      if (orphans[i].size >= bytes && orphans[i].size <= 4 * bytes + (1 << 20) && (ob < 0 || orphans[i].size < orphans[ob].size)) ob = (int)i;
    if (ob >= 0) {
      v.push_back({orphans[ob].ptr, orphans[ob].size, true});
      *p = orphans[ob].ptr;
      orphans.erase(orphans.begin() + ob);
      return hipSuccess;
    }
  }
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) {
    // drop cached free blocks of this stream (and the allocation cache of the C API) and retry once
    (void)hipGetLastError();
    for (auto it = v.begin(); it != v.end();) {
      if (!it->in_use) { (void)hipFree(it->ptr); it = v.erase(it); } else ++it;
    }
    ws_release_idle(ws_key(s).first, /*lock=*/false); // g_ws_mu is held here
    {
      std::lock_guard<std::mutex> la(g_ac_mu);
      alloc_cache_flush(ws_key(s).first);
    }
    e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return e;
  }
  v.push_back({q, bytes, true});
  *p = q;
  return hipSuccess;
}

hipError_t ws_free(void* p, hipStream_t s)
{
  std::lock_guard<std::mutex> lk(g_ws_mu);
  // the freeing thread's active device need not be the block's (a non-null stream belongs to exactly one device)
  for (auto& kv : g_ws) {
    if (kv.first.second != s) continue;
    for (auto& b : kv.second)
      if (b.ptr == p) {
        b.in_use = false;
        return hipSuccess;
      }
  }
  return hipErrorInvalidValue;
}

// Idle workspace of device `dev` back to the driver: the orphan list and the blocks cached for streams parked in the stream
// pool (a parked stream was synchronised by icicle_destroy_stream and nothing runs on it until it is handed out again).
// Called when an allocation fails and before the table-mode memory check of a cache build.
void ws_release_idle(int dev, bool lock)
{
  std::vector<hipStream_t> parked;
  {
    std::lock_guard<std::mutex> lp(g_sp_mu);
    auto it = g_stream_pool.find(dev);
    if (it != g_stream_pool.end()) parked = it->second;
  }
  std::unique_lock<std::mutex> lk(g_ws_mu, std::defer_lock);
  if (lock) lk.lock();
  for (auto& b : g_ws_orphans[dev]) (void)hipFree(b.ptr);
  g_ws_orphans[dev].clear();
  for (hipStream_t st : parked) {
    auto it = g_ws.find(WsKey(dev, st));
    if (it == g_ws.end()) continue;
    for (auto b = it->second.begin(); b != it->second.end();) {
      if (!b->in_use) {
        (void)hipFree(b->ptr);
        b = it->second.erase(b);
      } else ++b;
    }
  }
}

// The stream goes away (the caller has synchronised it): its idle blocks move to the device's orphan list, from which
// ws_alloc serves the next stream — the reference's host creates and destroys its streams in every prove
// (src/proof_helper.rs:32,186-187,235-236), and returning ~300 MB of MSM workspace to the driver each time cost
// 20 ms of hipMalloc / hipFree per prove.
void ws_release_stream(hipStream_t s)
{
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (auto it = g_ws.begin(); it != g_ws.end();) {
    if (it->first.second == s && (s != nullptr || it->first == ws_key(s))) {
      std::vector<WsBlock>& orphans = g_ws_orphans[it->first.first];
      size_t held = 0;
      for (auto& b : orphans) held += b.size;
      for (auto& b : it->second) {
        if (!b.in_use && held + b.size <= WS_ORPHAN_MAX) {
          orphans.push_back({b.ptr, b.size, false});
          held += b.size;
        } else (void)hipFree(b.ptr);
      }
      it = g_ws.erase(it);
    } else ++it;
  }
}

// ---- fixed-base tables of caller-owned base arrays (msm_plan.h) -------------------------------------------------------------------
struct BaseTable {
  int dev;
  uintptr_t ptr;
  size_t bytes;
  uint32_t n;
  bool g2;
  int form;
  uint32_t sightings;
  void* table; // nullptr: seen, not built
  size_t table_bytes;
  MsmGeom g;
  hipEvent_t built;
  uint64_t last_use;
  unsigned long long* sums; // device: hash sum of the bases at build time (msm_plan.h)
  int pins;                 // calls that hold the table between lookup and the end of their enqueueing
  uint64_t id;
  hipEvent_t used = nullptr; // recorded behind the last kernel of the most recent call that used the table (base_table_unpin), on that call's
                             // stream: a call on ANOTHER stream waits for it before its guarded refresh may rewrite rows those kernels read
};
static std::mutex g_bt_mu;
static std::vector<BaseTable> g_bt;
static std::vector<BaseTable> g_bt_parked; // retired while pinned: freed by the last unpin
static uint64_t g_bt_clock = 0, g_bt_next_id = 1;
static bool base_tables_enabled()
{
  static const bool on = !(getenv("ICICLE_SNARK_MSM_TABLES") && atoi(getenv("ICICLE_SNARK_MSM_TABLES")) == 0);
  return on;
}
static size_t base_tables_budget()
{
  static const size_t v = getenv("ICICLE_SNARK_MSM_TABLE_MB") ? (size_t)atoll(getenv("ICICLE_SNARK_MSM_TABLE_MB")) << 20 : (size_t)16 << 30;
  return v;
}
static void base_table_free(BaseTable& t) // hipFree waits for whatever enqueued work still reads the table
{
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != t.dev) (void)hipSetDevice(t.dev);
  if (t.table) (void)hipFree(t.table);
  if (t.sums) (void)hipFree(t.sums);
  if (t.built) (void)hipEventDestroy(t.built);
  if (t.used) (void)hipEventDestroy(t.used);
  if (cur != t.dev) (void)hipSetDevice(cur);
  t.table = nullptr;
  t.sums = nullptr;
  t.built = nullptr;
  t.used = nullptr;
}
// caller holds g_bt_mu.  A pinned table (a call is between its lookup and the end of its enqueueing: kernels that read the
// table may not be in any stream yet, so a free here could run before them) is parked and freed by the last unpin.
static void base_table_drop(BaseTable& t)
{
  if (!t.table) return;
  if (t.pins > 0) {
    g_bt_parked.push_back(t);
    t.table = nullptr;
    t.sums = nullptr;
    t.built = nullptr;
    t.used = nullptr;
    t.pins = 0;
    t.id = g_bt_next_id++; // the parked copy keeps the id the pin holders know
    return;
  }
  base_table_free(t);
}
void base_table_unpin(uint64_t id, hipStream_t s)
{
  if (!id) return;
  std::lock_guard<std::mutex> lk(g_bt_mu);
  for (BaseTable& t : g_bt)
    if (t.id == id) {
      // the call's last kernel is enqueued: whoever refreshes the table from another stream orders itself behind this point.  The
      // marks CHAIN: the unpinning stream first waits for the previous mark (another caller's stream, perhaps still reading the
      // table), so the one event a refresher waits for is behind the kernels of every stream that used the table, not just
      // behind those of the last one to unpin (round-5 advisor: two concurrent callers)
      if (t.table) {
        if (t.used) (void)hipStreamWaitEvent(s, t.used, 0);
        if (t.used || hipEventCreateWithFlags(&t.used, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(t.used, s);
      }
      (void)hipGetLastError();
      if (t.pins > 0) t.pins--;
      return;
    }
  for (size_t i = 0; i < g_bt_parked.size(); i++)
    if (g_bt_parked[i].id == id) {
      if (--g_bt_parked[i].pins <= 0) {
        base_table_free(g_bt_parked[i]);
        g_bt_parked.erase(g_bt_parked.begin() + i);
      }
      return;
    }
}
void note_device_write(const void* p, size_t bytes)
{
  if (!p || !bytes) return;
  std::lock_guard<std::mutex> lk(g_bt_mu);
  if (g_bt.empty()) return;
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  for (size_t i = 0; i < g_bt.size();) {
    if (g_bt[i].ptr < hi && lo < g_bt[i].ptr + g_bt[i].bytes) {
      base_table_drop(g_bt[i]);
      g_bt.erase(g_bt.begin() + i);
    } else i++;
  }
}
BaseTableState base_table_lookup(const void* bases, size_t bytes, uint32_t n, bool g2, int form, size_t table_bytes, BaseTableRef* ref)
{
  if (!base_tables_enabled()) return BASE_TABLE_NONE;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_bt_mu);
  for (BaseTable& t : g_bt) {
    if (t.dev != dev || t.ptr != (uintptr_t)bases || t.n != n || t.g2 != g2 || t.form != form) continue;
    t.last_use = ++g_bt_clock;
    if (t.table) {
      ref->table = t.table;
      ref->g = t.g;
      ref->built = t.built;
      ref->sums = t.sums;
      ref->id = t.id;
      ref->used = t.used;
      t.pins++;
      return BASE_TABLE_HIT;
    }
    t.sightings++;
    if (t.sightings < 2) return BASE_TABLE_NONE;
    // second MSM over this array: worth a table if it fits the budget (least recently used tables make room)
    if (table_bytes > base_tables_budget()) return BASE_TABLE_NONE;
    for (;;) {
      size_t used = 0;
      size_t lru = (size_t)-1;
      for (size_t i = 0; i < g_bt.size(); i++) {
        if (g_bt[i].dev != dev || !g_bt[i].table) continue;
        used += g_bt[i].table_bytes;
        if (lru == (size_t)-1 || g_bt[i].last_use < g_bt[lru].last_use) lru = i;
      }
      if (used + table_bytes <= base_tables_budget() || lru == (size_t)-1) break;
      base_table_drop(g_bt[lru]);
      g_bt[lru].sightings = 0;
    }
    return BASE_TABLE_BUILD;
  }
  // first sighting: remember the array (bounded: the oldest table-less entry makes room)
  if (g_bt.size() >= 64) {
    size_t old = (size_t)-1;
    for (size_t i = 0; i < g_bt.size(); i++)
      if (!g_bt[i].table && (old == (size_t)-1 || g_bt[i].last_use < g_bt[old].last_use)) old = i;
    if (old == (size_t)-1) return BASE_TABLE_NONE;
    g_bt.erase(g_bt.begin() + old);
  }
  g_bt.push_back({dev, (uintptr_t)bases, bytes, n, g2, form, 1u, nullptr, 0, MsmGeom(), nullptr, ++g_bt_clock, nullptr, 0, g_bt_next_id++, nullptr});
  return BASE_TABLE_NONE;
}
// the table (and the hash sum of its bases in sums[0]) are complete in the order of stream s.  On return *ref names the table,
// pinned for the publishing call — also when the entry went away in the meantime (a write to the bases between lookup and
// publish): the table then serves this one call and is freed by its unpin.
void base_table_publish(const void* bases, uint32_t n, bool g2, int form, void* table, size_t table_bytes, const MsmGeom& g, unsigned long long* sums, hipStream_t s, BaseTableRef* ref)
{
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(ev, s);
  ref->table = table;
  ref->g = g;
  ref->built = nullptr; // built on the caller's own stream
  ref->sums = sums;
  std::lock_guard<std::mutex> lk(g_bt_mu);
  for (BaseTable& t : g_bt) {
    if (t.dev != dev || t.ptr != (uintptr_t)bases || t.n != n || t.g2 != g2 || t.form != form || t.table) continue;
    t.table = table;
    t.table_bytes = table_bytes;
    t.g = g;
    t.built = ev;
    t.sums = sums;
    t.pins = 1;
    ref->id = t.id;
    return;
  }
  BaseTable orphan = {dev, (uintptr_t)bases, 0, n, g2, form, 0u, table, table_bytes, g, ev, 0, sums, 1, g_bt_next_id++, nullptr};
  ref->id = orphan.id;
  g_bt_parked.push_back(orphan);
}

} // namespace isnark

using namespace isnark;

ISNARK_API const char* icicle_snark_last_error(void) { return g_err; }

ISNARK_API eIcicleError icicle_load_backend(const char*, bool) { return ICICLE_SUCCESS; }
ISNARK_API eIcicleError icicle_load_backend_from_env_or_default(void) { return ICICLE_SUCCESS; }

ISNARK_API eIcicleError icicle_is_device_available(const IcicleDevice* dev)
{
  if (!dev || !type_ok(dev->type)) return ICICLE_INVALID_DEVICE;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || dev->id < 0 || dev->id >= n) return ICICLE_INVALID_DEVICE;
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_set_device(const IcicleDevice* dev)
{
  if (!dev || !type_ok(dev->type)) {
    set_last_error("device type '%.63s' is not registered (only HIP)", dev ? dev->type : "(null)");
    return ICICLE_INVALID_DEVICE;
  }
  ICICLE_TRY(icicle_is_device_available(dev));
  HIP_TRY(hipSetDevice(dev->id), ICICLE_INVALID_DEVICE);
  t_device = dev->id;
  if (g_default_device < 0) g_default_device = dev->id;
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_set_default_device(const IcicleDevice* dev)
{
  ICICLE_TRY(icicle_set_device(dev));
  g_default_device = dev->id;
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_get_active_device(IcicleDevice* dev)
{
  if (!dev) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  memset(dev->type, 0, sizeof dev->type);
  strcpy(dev->type, "HIP");
  dev->id = t_device;
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_get_registered_devices(char* output, size_t output_size)
{
  if (!output || output_size < 4) return ICICLE_INVALID_ARGUMENT;
  strcpy(output, "HIP");
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_get_device_count(int* n)
{
  if (!n) return ICICLE_INVALID_POINTER;
  HIP_TRY(hipGetDeviceCount(n), ICICLE_INVALID_DEVICE);
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_is_host_memory(const void* ptr)
{
  return identify(ptr) < 0 ? ICICLE_SUCCESS : ICICLE_INVALID_POINTER;
}

ISNARK_API eIcicleError icicle_is_active_device_memory(const void* ptr)
{
  int d = identify(ptr);
  if (d < 0) return ICICLE_INVALID_POINTER;
  return d == active_device() ? ICICLE_SUCCESS : ICICLE_INVALID_POINTER;
}

// ---- device allocations ---------------------------------------------------------------------------------------------
// The reference's host allocates and drops its big work buffers in EVERY prove (DeviceVec::device_malloc_async / Drop →
// icicle_free: n_coef·32 B, 3n·32 B, n_vars·32 B — 350 MB at 1.6 M constraints, src/proof_helper.rs:41-42,189).  hipMalloc /
// hipFree of such blocks cost milliseconds each and hipFree drains the device, so freed blocks of ≥ 1 MB are kept in a
// size-keyed cache (per device, ≤ ALLOC_CACHE_MAX bytes) and handed out again.  icicle_free keeps hipFree's implicit
// "everything submitted so far has finished" by synchronising the device before the block becomes reusable.
constexpr size_t ALLOC_CACHE_MIN = 1u << 20;
static size_t alloc_cache_max()
{
  static const size_t v = getenv("ICICLE_SNARK_ALLOC_CACHE_MB") ? (size_t)atoll(getenv("ICICLE_SNARK_ALLOC_CACHE_MB")) << 20 : (size_t)16 << 30;
  return v;
}
struct AllocCache {
  std::multimap<size_t, void*> blocks; // capacity → block
  size_t bytes = 0;
};
static std::map<int, AllocCache> g_alloc_cache;            // device → cache
static std::map<uintptr_t, size_t> g_capacity;             // live block → its real (hipMalloc) size

static void alloc_cache_flush(int dev) // caller holds g_ac_mu
{
  AllocCache& c = g_alloc_cache[dev];
  for (auto& kv : c.blocks) (void)hipFree(kv.second);
  c.blocks.clear();
  c.bytes = 0;
}
namespace isnark {
void release_cached_device_memory()
{
  int d = 0;
  (void)hipGetDevice(&d);
  {
    std::lock_guard<std::mutex> lk(g_ac_mu);
    alloc_cache_flush(d);
  }
  ws_release_idle(d);
}
} // namespace isnark
static hipError_t cached_malloc(void** ptr, size_t size)
{
  const int dev = t_device;
  if (size >= ALLOC_CACHE_MIN) {
    std::lock_guard<std::mutex> lk(g_ac_mu);
    AllocCache& c = g_alloc_cache[dev];
    auto it = c.blocks.lower_bound(size);
    if (it != c.blocks.end() && it->first <= size + size / 8) {
      *ptr = it->second;
      g_capacity[(uintptr_t)*ptr] = it->first;
      c.bytes -= it->first;
      c.blocks.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(ptr, size);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_ac_mu);
    alloc_cache_flush(dev); // give the cached blocks back to the driver and retry once
    e = hipMalloc(ptr, size);
  }
  if (e == hipSuccess && size >= ALLOC_CACHE_MIN) {
    std::lock_guard<std::mutex> lk(g_ac_mu);
    g_capacity[(uintptr_t)*ptr] = size;
  }
  return e;
}
// the block leaves the caller's hands: cache it (after draining the device) or return it to the driver
static hipError_t cached_free(void* ptr, int owner)
{
  size_t cap = 0;
  {
    std::lock_guard<std::mutex> lk(g_ac_mu);
    auto it = g_capacity.find((uintptr_t)ptr);
    if (it != g_capacity.end()) {
      cap = it->second;
      g_capacity.erase(it);
    }
  }
  if (cap >= ALLOC_CACHE_MIN && cap <= alloc_cache_max() / 2) {
    hipError_t e = hipDeviceSynchronize(); // what hipFree would have implied
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_ac_mu);
    AllocCache& c = g_alloc_cache[owner];
    while (!c.blocks.empty() && c.bytes + cap > alloc_cache_max()) { // evict the smallest blocks first
      auto it = c.blocks.begin();
      (void)hipFree(it->second);
      c.bytes -= it->first;
      c.blocks.erase(it);
    }
    c.blocks.emplace(cap, ptr);
    c.bytes += cap;
    return hipSuccess;
  }
  return hipFree(ptr);
}

ISNARK_API eIcicleError icicle_malloc(void** ptr, size_t size)
{
  if (!ptr) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  HIP_TRY(cached_malloc(ptr, size), ICICLE_ALLOCATION_FAILED);
  track(*ptr, size);
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_malloc_async(void** ptr, size_t size, icicleStreamHandle stream)
{
  if (!ptr) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  // The returned block is usable at once on any stream, which satisfies the stream-ordered contract; the
  // stream-ordered pool of the HIP runtime is avoided (see common.h, workspace arena).
  (void)stream;
  HIP_TRY(cached_malloc(ptr, size), ICICLE_ALLOCATION_FAILED);
  track(*ptr, size);
  return ICICLE_SUCCESS;
}

// the whole tracked block that contains p is about to go away / be rewritten
static void note_block_write(const void* p)
{
  size_t size = 0;
  uintptr_t base = 0;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return;
    --it;
    if ((uintptr_t)p >= it->first + it->second.first) return;
    base = it->first;
    size = it->second.first;
  }
  note_device_write((const void*)base, size);
}

ISNARK_API eIcicleError icicle_free(void* ptr)
{
  if (!ptr) return ICICLE_SUCCESS;
  note_block_write(ptr);
  // memory of a non-active device: switch, release, switch back (runtime.cpp:66-93)
  int owner = identify(ptr);
  if (owner < 0) return ICICLE_INVALID_POINTER;
  int cur = active_device();
  if (owner != cur) (void)hipSetDevice(owner);
  untrack(ptr);
  hipError_t e = cached_free(ptr, owner);
  if (owner != cur && cur >= 0) (void)hipSetDevice(cur);
  if (e != hipSuccess) {
    set_last_error("hipFree failed: %s", hipGetErrorString(e));
    return ICICLE_DEALLOCATION_FAILED;
  }
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_free_async(void* ptr, icicleStreamHandle stream)
{
  if (!ptr) return ICICLE_SUCCESS;
  note_block_write(ptr);
  const int owner = identify(ptr);
  if (!untrack(ptr)) return ICICLE_INVALID_POINTER;
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream), ICICLE_SYNCHRONIZATION_FAILED);
  HIP_TRY(cached_free(ptr, owner < 0 ? t_device : owner), ICICLE_DEALLOCATION_FAILED);
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_get_available_memory(size_t* total, size_t* free_)
{
  if (!total || !free_) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  HIP_TRY(hipMemGetInfo(free_, total), ICICLE_UNKNOWN_ERROR);
  {
    std::lock_guard<std::mutex> lk(g_ac_mu); // cached blocks are available to the caller (the next allocation re-uses or releases them)
    *free_ += g_alloc_cache[t_device].bytes;
  }
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_memset(void* ptr, int value, size_t size)
{
  if (icicle_is_active_device_memory(ptr) != ICICLE_SUCCESS) return ICICLE_INVALID_POINTER;
  note_device_write(ptr, size);
  HIP_TRY(hipMemset(ptr, value, size), ICICLE_UNKNOWN_ERROR);
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_memset_async(void* ptr, int value, size_t size, icicleStreamHandle stream)
{
  if (icicle_is_active_device_memory(ptr) != ICICLE_SUCCESS) return ICICLE_INVALID_POINTER;
  note_device_write(ptr, size);
  HIP_TRY(hipMemsetAsync(ptr, value, size, (hipStream_t)stream), ICICLE_UNKNOWN_ERROR);
  return ICICLE_SUCCESS;
}

static hipMemcpyKind direction(void* dst, const void* src)
{
  bool d = identify(dst) >= 0, s = identify(src) >= 0;
  if (d && s) return hipMemcpyDeviceToDevice;
  if (d) return hipMemcpyHostToDevice;
  if (s) return hipMemcpyDeviceToHost;
  return hipMemcpyHostToHost;
}

ISNARK_API eIcicleError icicle_copy(void* dst, const void* src, size_t size)
{
  ICICLE_TRY(require_device());
  if (identify(dst) >= 0) note_device_write(dst, size);
  HIP_TRY(hipMemcpy(dst, src, size, direction(dst, src)), ICICLE_COPY_FAILED);
  return ICICLE_SUCCESS;
}
ISNARK_API eIcicleError icicle_copy_async(void* dst, const void* src, size_t size, icicleStreamHandle stream)
{
  ICICLE_TRY(require_device());
  if (identify(dst) >= 0) note_device_write(dst, size);
  HIP_TRY(hipMemcpyAsync(dst, src, size, direction(dst, src), (hipStream_t)stream), ICICLE_COPY_FAILED);
  return ICICLE_SUCCESS;
}
// Large copies from/to PAGEABLE host memory (what the reference's Rust host passes: Vec<…> and mmap'ed files,
// src/proof_helper.rs:73,96-101, src/cache.rs:199-207) go through the parallel pinned-staging engine: the stream is
// drained first (the copy is ordered behind everything enqueued before it), the bytes are in place when the call
// returns — a stricter behaviour than "asynchronous", which a pageable copy never is in CUDA/HIP either.
static eIcicleError big_pageable_copy(void* dst, const void* src, size_t size, bool to_device, hipStream_t stream, bool* done)
{
  *done = false;
  if (size < STAGED_MIN_BYTES || is_pinned_host(to_device ? src : dst)) return ICICLE_SUCCESS;
  HIP_TRY(hipStreamSynchronize(stream), ICICLE_SYNCHRONIZATION_FAILED);
  const CopyJob j = {dst, src, size};
  HIP_TRY(staged_copy(active_device(), &j, 1, to_device), ICICLE_COPY_FAILED);
  *done = true;
  return ICICLE_SUCCESS;
}
ISNARK_API eIcicleError icicle_copy_to_host(void* dst, const void* src, size_t size)
{
  ICICLE_TRY(require_device());
  bool done;
  ICICLE_TRY(big_pageable_copy(dst, src, size, false, nullptr, &done));
  if (!done) HIP_TRY(hipMemcpy(dst, src, size, hipMemcpyDeviceToHost), ICICLE_COPY_FAILED);
  return ICICLE_SUCCESS;
}
ISNARK_API eIcicleError icicle_copy_to_host_async(void* dst, const void* src, size_t size, icicleStreamHandle stream)
{
  ICICLE_TRY(require_device());
  bool done;
  ICICLE_TRY(big_pageable_copy(dst, src, size, false, (hipStream_t)stream, &done));
  if (!done) HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, (hipStream_t)stream), ICICLE_COPY_FAILED);
  return ICICLE_SUCCESS;
}
ISNARK_API eIcicleError icicle_copy_to_device(void* dst, const void* src, size_t size)
{
  ICICLE_TRY(require_device());
  note_device_write(dst, size);
  bool done;
  ICICLE_TRY(big_pageable_copy(dst, src, size, true, nullptr, &done));
  if (!done) HIP_TRY(hipMemcpy(dst, src, size, hipMemcpyHostToDevice), ICICLE_COPY_FAILED);
  return ICICLE_SUCCESS;
}
ISNARK_API eIcicleError icicle_copy_to_device_async(void* dst, const void* src, size_t size, icicleStreamHandle stream)
{
  ICICLE_TRY(require_device());
  note_device_write(dst, size);
  bool done;
  ICICLE_TRY(big_pageable_copy(dst, src, size, true, (hipStream_t)stream, &done));
  if (!done) HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, (hipStream_t)stream), ICICLE_COPY_FAILED);
  return ICICLE_SUCCESS;
}

// Streams are pooled: hipStreamCreate costs ≈4 ms on this stack and the reference's host creates and destroys three to
// five streams in every prove (src/proof_helper.rs:32,186-187, src/conversions.rs:14).  A destroyed stream is drained and
// parked (≤ STREAM_POOL_MAX per device, with the workspace blocks cached for it); the next create takes it back.
constexpr size_t STREAM_POOL_MAX = 10; // a key's six + the two upload lanes of a cold pipeline (prover/cache.cpp) + slack
ISNARK_API eIcicleError icicle_create_stream(icicleStreamHandle* stream)
{
  if (!stream) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  {
    std::lock_guard<std::mutex> lk(g_sp_mu);
    std::vector<hipStream_t>& v = g_stream_pool[t_device];
    if (!v.empty()) {
      *stream = v.back();
      v.pop_back();
      return ICICLE_SUCCESS;
    }
  }
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), ICICLE_STREAM_CREATION_FAILED);
  *stream = s;
  return ICICLE_SUCCESS;
}
ISNARK_API eIcicleError icicle_destroy_stream(icicleStreamHandle stream)
{
  if (!stream) return ICICLE_SUCCESS;
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream), ICICLE_STREAM_DESTRUCTION_FAILED);
  {
    int d = 0;
    hipDevice_t sd = 0;
    if (hipStreamGetDevice((hipStream_t)stream, &sd) == hipSuccess) d = (int)sd; // the stream's own device, whatever is active now
    else (void)hipGetDevice(&d);
    std::lock_guard<std::mutex> lk(g_sp_mu);
    std::vector<hipStream_t>& v = g_stream_pool[d];
    if (v.size() < STREAM_POOL_MAX) {
      v.push_back((hipStream_t)stream);
      return ICICLE_SUCCESS;
    }
  }
  // the workspace blocks cached for this stream move to the device's orphan list
  ws_release_stream((hipStream_t)stream);
  stream_pipes_forget((hipStream_t)stream);
  HIP_TRY(hipStreamDestroy((hipStream_t)stream), ICICLE_STREAM_DESTRUCTION_FAILED);
  return ICICLE_SUCCESS;
}
namespace isnark {
// What the first cache load of a process would otherwise pay inside its first prove (measured on MI355X: 48 ms for six
// streams and their DMA queues, ≈ 10 ms for the pinned staging pool): create it ahead of time on `dev` — the prover host
// calls this from a helper thread when its CacheManager is created (prover.cpp).  Streams go to the pool icicle_create_stream
// draws from; each has carried one host→device copy, which is what sets up its DMA queue.
void prewarm_device(int dev, int n_streams)
{
  if (dev < 0 || dev >= STAGED_DEVICES || hipSetDevice(dev) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  {
    StagedPool& P = g_staged[dev];
    std::lock_guard<std::mutex> lk(P.mu);
    if (staged_pool_init(P) != hipSuccess) (void)hipGetLastError();
  }
  size_t have = 0;
  {
    std::lock_guard<std::mutex> lk(g_sp_mu);
    have = g_stream_pool[dev].size();
  }
  // one staging chunk per stream, host to device: a copy of a few KB goes through a blit kernel and leaves the stream's DMA queue
  // to be set up by the first real chunk of a cold upload (10–40 ms for the first 141 MB of a process instead of 4)
  const size_t warm_bytes = g_staged[dev].pinned ? STAGED_CHUNK : 4096;
  void* d = nullptr;
  if (hipMalloc(&d, warm_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  std::vector<hipStream_t> fresh;
  for (size_t k = have; k < (size_t)n_streams && k < STREAM_POOL_MAX; k++) {
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
    (void)hipMemcpyAsync(d, g_staged[dev].pinned ? (void*)g_staged[dev].pinned : d, warm_bytes, g_staged[dev].pinned ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st);
    fresh.push_back(st);
  }
  for (hipStream_t st : fresh) (void)hipStreamSynchronize(st);
  (void)hipFree(d);
  (void)hipGetLastError();
  // which of them share a hardware pipe (microbench.hip): the keys built later deal their streams' roles by it (prover/cache.cpp)
  static const bool pipe_roles = !(getenv("ICICLE_SNARK_PIPE_ROLES") && atoi(getenv("ICICLE_SNARK_PIPE_ROLES")) == 0);
  if (pipe_roles && fresh.size() >= 2) stream_pipes_measure(fresh.data(), (int)fresh.size());
  std::lock_guard<std::mutex> lk(g_sp_mu);
  std::vector<hipStream_t>& v = g_stream_pool[dev];
  for (hipStream_t st : fresh) {
    if (v.size() < STREAM_POOL_MAX) v.push_back(st);
    else (void)hipStreamDestroy(st);
  }
}
void prewarm_modules(int dev)
{
  if (dev < 0 || hipSetDevice(dev) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  module_warm_csr(s);
  module_warm_qap(s);
  module_warm_sort(s);
  module_warm_g1(s);
  module_warm_g2(s);
  module_warm_g2acc(s);
  module_warm_ntt(s);
  module_warm_vec(s);
  (void)hipStreamSynchronize(s);
  (void)hipStreamDestroy(s);
  (void)hipGetLastError();
}
int default_device_or_none() { return active_device(); }
} // namespace isnark

ISNARK_API eIcicleError icicle_stream_synchronize(icicleStreamHandle stream)
{
  ICICLE_TRY(require_device());
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream), ICICLE_SYNCHRONIZATION_FAILED);
  return ICICLE_SUCCESS;
}
ISNARK_API eIcicleError icicle_device_synchronize(void)
{
  ICICLE_TRY(require_device());
  HIP_TRY(hipDeviceSynchronize(), ICICLE_SYNCHRONIZATION_FAILED);
  return ICICLE_SUCCESS;
}

ISNARK_API eIcicleError icicle_get_device_properties(IcicleDeviceProperties* p)
{
  if (!p) return ICICLE_INVALID_POINTER;
  ICICLE_TRY(require_device());
  p->using_host_memory = false;
  p->num_memory_regions = 0;
  p->supports_pinned_memory = true;
  return ICICLE_SUCCESS;
}

// ---- config extension: string-keyed int/bool bag (icicle/include/icicle/config_extension.h) ----
struct ConfigExtension {
  std::map<std::string, int> ints;
  std::map<std::string, bool> bools;
};
ISNARK_API ConfigExtension* create_config_extension(void) { return new ConfigExtension(); }
ISNARK_API void destroy_config_extension(ConfigExtension* ext) { delete ext; }
ISNARK_API void config_extension_set_int(ConfigExtension* ext, const char* key, int value)
{
  if (ext && key) ext->ints[key] = value;
}
ISNARK_API void config_extension_set_bool(ConfigExtension* ext, const char* key, bool value)
{
  if (ext && key) ext->bools[key] = value;
}
// The reference throws on a null extension / missing key (config_extension.cpp:21-30); a C ABI must
// not unwind, so these return 0 / false and record the error text.
ISNARK_API int config_extension_get_int(const ConfigExtension* ext, const char* key)
{
  if (!ext || !key) { set_last_error("ConfigExtension is null"); return 0; }
  auto it = ext->ints.find(key);
  if (it == ext->ints.end()) { set_last_error("ConfigExtension: no int key '%s'", key); return 0; }
  return it->second;
}
ISNARK_API bool config_extension_get_bool(const ConfigExtension* ext, const char* key)
{
  if (!ext || !key) { set_last_error("ConfigExtension is null"); return false; }
  auto it = ext->bools.find(key);
  if (it == ext->bools.end()) { set_last_error("ConfigExtension: no bool key '%s'", key); return false; }
  return it->second;
}
ISNARK_API ConfigExtension* clone_config_extension(const ConfigExtension* ext)
{
  if (!ext) { set_last_error("ConfigExtension is null"); return nullptr; }
  return new ConfigExtension(*ext);
}

namespace isnark {
bool ext_get_int(const ConfigExtension* ext, const char* key, int* out)
{
  if (!ext) return false;
  auto it = ext->ints.find(key);
  if (it == ext->ints.end()) return false;
  *out = it->second;
  return true;
}
bool ext_get_bool(const ConfigExtension* ext, const char* key, bool* out)
{
  if (!ext) return false;
  auto it = ext->bools.find(key);
  if (it == ext->bools.end()) return false;
  *out = it->second;
  return true;
}
} // namespace isnark
