// common.h — shared plumbing of the C-ABI implementation (error handling, staging of host operands).
#pragma once
#include <atomic>
#include <condition_variable>
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/icicle_snark_hip.h"

namespace isnark {

void set_last_error(const char* fmt, ...);

#define ISNARK_API extern "C" __attribute__((visibility("default")))

// Return `code` (an eIcicleError) from the enclosing function when a HIP call fails.
#define HIP_TRY(call, code)                                                                                   \
  do {                                                                                                        \
    hipError_t e__ = (call);                                                                                  \
    if (e__ != hipSuccess) {                                                                                  \
      ::isnark::set_last_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);   \
      return (code);                                                                                          \
    }                                                                                                         \
  } while (0)

#define ICICLE_TRY(call)                                                                                      \
  do {                                                                                                        \
    eIcicleError e__ = (call);                                                                                \
    if (e__ != ICICLE_SUCCESS) return e__;                                                                    \
  } while (0)

// True once icicle_set_device() selected the HIP device on this thread (or a default exists).
eIcicleError require_device();
// `p` lies inside a block handed out by icicle_malloc{,_async} (the map behind icicle_is_active_device_memory)
bool is_tracked_device_ptr(const void* p);
// create on `dev`, ahead of the first cache load, what that load would pay for: the pinned staging pool and n_streams pooled
// streams with their DMA queues set up (runtime.cpp); the device the calling thread would use, or −1 when none is chosen yet
void prewarm_device(int dev, int n_streams);
// pipe class (0 … 3) of each of n ≤ 16 streams, measured (microbench.hip); false (and −1 everywhere) when the measurement was disturbed
bool probe_stream_pipes(hipStream_t* st, int n, int* cls);
// the same through a per-process cache (a stream keeps its queue): measured once — by the prewarm thread for the pool's streams
// (stream_pipes_measure), else by the first key that asks — and looked up afterwards; stream_pipes_forget when a stream is destroyed
void stream_pipes_measure(hipStream_t* st, int n);
bool stream_pipe_classes(hipStream_t* st, int n, int* cls, bool allow_measure = true); // (false: only what has been measured already)
void stream_pipes_forget(hipStream_t st);
// code objects of the prove path loaded onto `dev` ahead of their first use (one empty launch per translation unit)
void prewarm_modules(int dev);
void module_warm_csr(hipStream_t s);
void module_warm_qap(hipStream_t s);
void module_warm_sort(hipStream_t s);
void module_warm_g1(hipStream_t s);
void module_warm_g2(hipStream_t s);
void module_warm_g2acc(hipStream_t s);
void module_warm_ntt(hipStream_t s);
void module_warm_vec(hipStream_t s);
int default_device_or_none();
// hand the blocks cached by icicle_free back to the driver (call before giving up on an allocation)
void release_cached_device_memory();

// ---- pageable-memory copies ------------------------------------------------------------------------------------------
// A hipMemcpy from/to pageable host memory is a single-threaded staging copy (≈5–10 GB/s measured here).  staged_copy
// runs up to STAGED_LANES workers, each moving 2 MB chunks through its own pair of pinned buffers and its own stream, so
// page faults / memcpy of one chunk overlap the DMA of the others (54 GB/s host→device on the MI355X box).  It returns
// when every byte has arrived.  `lanes`: streams to enqueue the DMAs on (idle streams of the caller); nullptr = the
// engine's own persistent lane streams of the active device.  Callers order it against earlier stream work themselves.
struct CopyJob {
  void* dst;
  const void* src;
  size_t n;
};
constexpr int STAGED_LANES = 8;
constexpr size_t STAGED_CHUNK_BYTES = 2u << 20; // bytes per staging chunk (and per DMA)
// Progress of a host→device staged copy that runs on a thread of its own while the caller goes on enqueueing work: the HEAD —
// the first head_bytes of job 0 — has arrived once every lane's `ev` has completed.  Each lane records its event behind its last
// chunk of the head and then counts itself in `lanes_reported`; the caller waits (on the host, briefly: the records follow the
// staging copies, not the DMAs) until lanes_reported == lanes_total, and only then makes its streams wait for ev[0 … lanes_total):
// a hipStreamWaitEvent on an event that has not been recorded yet would not wait.
struct StagedProgress {
  size_t head_bytes = 0;
  hipEvent_t ev[STAGED_LANES] = {};
  std::atomic<int> lanes_total{-1}; // set before the workers start
  std::atomic<int> lanes_reported{0};
  std::atomic<bool> done{false};    // set by whoever ran staged_copy, after it returned (error paths included)
  // the caller BLOCKS until the head's records are in (a yield() spin until round 4: it held a CPU of the container's quota that
  // the staging lanes and the runtime's own threads compete for)
  std::mutex m;
  std::condition_variable cv;
  void notify()
  {
    std::lock_guard<std::mutex> lk(m);
    cv.notify_all();
  }
  void wait_head()
  {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] {
      const int tot = lanes_total.load(std::memory_order_acquire);
      return (tot >= 0 && lanes_reported.load(std::memory_order_acquire) >= tot) || done.load(std::memory_order_acquire);
    });
  }
};
hipError_t staged_copy(int device_id, const CopyJob* jobs, size_t njobs, bool to_device, const hipStream_t* lanes = nullptr, int n_lanes = 0, bool own_temp_streams = false,
                       StagedProgress* progress = nullptr);
// the calling thread's next staged copies read sources inside [base, base + len) from file descriptor fd (fd < 0: off)
void staged_copy_file_hint(const void* base, size_t len, int fd);
void staged_copy_file_hint_get(const void** base, size_t* len, int* fd); // the calling thread's hint (to hand it to worker threads)
// true when hipMemcpyAsync can DMA straight from/to `host_ptr` (pinned / registered memory)
bool is_pinned_host(const void* host_ptr, int device_id = -1);
// copies of at least this many bytes from/to pageable memory take the staged path
constexpr size_t STAGED_MIN_BYTES = 4u << 20;

// Every entry point that WRITES device memory on behalf of the caller reports the range here (copies, memsets, frees, device
// results of vec ops / conversions / transforms / MSMs): fixed-base tables built for a base array that overlaps it are dropped
// (msm_tables in runtime.cpp).  Cheap: a lock and a scan of the handful of live tables.
void note_device_write(const void* p, size_t bytes);

// Stages a host-resident operand on the device for the lifetime of the object (the reference's
// wrappers do the same per VecOpsConfig / NTTConfig / MSMConfig flags, e.g.
// icicle/backend/cuda/src/field/cuda_vec_ops.cu:17-54).  Output operands are copied back by finish().
//
// This is the slow, convenience path (the prover passes device pointers everywhere).  It is kept
// deliberately conservative: plain hipMalloc/hipFree and BLOCKING copies — H2D completes before the
// kernel is enqueued, D2H runs after an explicit stream synchronisation.  (Stream-ordered pool
// memory + asynchronous copies from/to pageable host memory were observed to race on this stack.)
class Staged
{
public:
  Staged() {}
  ~Staged() { release(); }
  // Residency is declared by the caller's flags — with one leniency the reference's own host needs: the Rust wrapper of
  // ScalarField::from_mont passes a DEVICE buffer as the output while leaving is_result_on_device at its default, false
  // (wrappers/rust/icicle-core/src/field.rs:379-398 → src/proof_helper.rs:79, src/cache.rs:228; the CUDA backend then
  // "copies back" into that device pointer, cuda_mont.cuh:33-50).  A pointer that lies inside a block this runtime
  // allocated is therefore treated as device memory whatever the flag says.
  eIcicleError in(const void* p, size_t bytes, bool on_device, hipStream_t s)
  {
    stream_ = s;
    if (!on_device && bytes && is_tracked_device_ptr(p)) on_device = true;
    if (on_device || bytes == 0) {
      dev_ = const_cast<void*>(p);
      return ICICLE_SUCCESS;
    }
    HIP_TRY(hipMalloc(&dev_, bytes), ICICLE_ALLOCATION_FAILED);
    owned_ = true;
    if (bytes >= STAGED_MIN_BYTES && !is_pinned_host(p)) {
      int d = 0;
      (void)hipGetDevice(&d);
      const CopyJob j = {dev_, p, bytes};
      HIP_TRY(staged_copy(d, &j, 1, true), ICICLE_COPY_FAILED);
    } else HIP_TRY(hipMemcpy(dev_, p, bytes, hipMemcpyHostToDevice), ICICLE_COPY_FAILED);
    return ICICLE_SUCCESS;
  }
  eIcicleError out(void* p, size_t bytes, bool on_device, hipStream_t s)
  {
    stream_ = s;
    host_out_ = nullptr;
    if (!on_device && bytes && is_tracked_device_ptr(p)) on_device = true;
    if (on_device || bytes == 0) {
      dev_ = p;
      if (bytes) note_device_write(p, bytes);
      return ICICLE_SUCCESS;
    }
    HIP_TRY(hipMalloc(&dev_, bytes), ICICLE_ALLOCATION_FAILED);
    owned_ = true;
    host_out_ = p;
    bytes_ = bytes;
    return ICICLE_SUCCESS;
  }
  // a host-resident result is complete when the call returns, also for is_async calls
  eIcicleError finish()
  {
    if (host_out_) {
      HIP_TRY(hipStreamSynchronize(stream_), ICICLE_SYNCHRONIZATION_FAILED);
      if (bytes_ >= STAGED_MIN_BYTES && !is_pinned_host(host_out_)) {
        int d = 0;
        (void)hipGetDevice(&d);
        const CopyJob j = {host_out_, dev_, bytes_};
        HIP_TRY(staged_copy(d, &j, 1, false), ICICLE_COPY_FAILED);
      } else HIP_TRY(hipMemcpy(host_out_, dev_, bytes_, hipMemcpyDeviceToHost), ICICLE_COPY_FAILED);
      host_out_ = nullptr;
    }
    return ICICLE_SUCCESS;
  }
  void release()
  {
    if (owned_ && dev_) {
      (void)hipStreamSynchronize(stream_); // kernels reading a staged input may still be in flight
      (void)hipFree(dev_);
    }
    owned_ = false;
    dev_ = nullptr;
  }
  template <class T>
  T* ptr() const
  {
    return reinterpret_cast<T*>(dev_);
  }

private:
  void* dev_ = nullptr;
  void* host_out_ = nullptr;
  size_t bytes_ = 0;
  bool owned_ = false;
  hipStream_t stream_ = nullptr;
};

// Workspace allocation for kernels' temporaries: a per-stream cache of device blocks.
// ws_free() only marks the block reusable by LATER work on the SAME stream (in-order execution makes
// that safe without any synchronisation); blocks are returned to the driver when their stream is
// destroyed or the library unloads.  hipMallocAsync's stream-ordered pool is deliberately not used:
// on this ROCm stack recycled pool blocks were observed to be handed out while earlier kernels on
// the stream were still using them (nondeterministic MSM results; see DESIGN.md §1 "workspace arena").
hipError_t ws_alloc(void** p, size_t bytes, hipStream_t s);
hipError_t ws_free(void* p, hipStream_t s);
void ws_release_stream(hipStream_t s); // caller has synchronised the stream

// Scoped workspace block: returned to its stream's arena when the scope ends, on every path (error returns too).
template <class T>
struct WsScoped {
  T* p = nullptr;
  hipStream_t s = nullptr;
  WsScoped() = default;
  WsScoped(const WsScoped&) = delete;
  WsScoped& operator=(const WsScoped&) = delete;
  ~WsScoped() { release(); }
  hipError_t alloc(size_t count, hipStream_t stream)
  {
    release();
    s = stream;
    return ws_alloc((void**)&p, (count ? count : 1) * sizeof(T), stream);
  }
  void release()
  {
    if (p) (void)ws_free(p, s);
    p = nullptr;
  }
  operator T*() const { return p; }
};

// First statement of the short kernels on a prove's critical chains (QAP front end, digit sorts): raise the wave's issue
// priority.  When the witness MSMs of a small circuit run beside them (prover.cpp: early start), an accumulation wave and a
// transform wave share a SIMD; without the priority they split its issue slots and the chain H waits for stretches 3×
// (100 k constraints: last transform pass 0.50 → 0.2 ms).  Neutral for the large circuits, whose QAP runs alone.
#if defined(__HIP_DEVICE_COMPILE__)
#define ISNARK_CRITICAL_CHAIN_KERNEL() __builtin_amdgcn_s_setprio(3)
#else
#define ISNARK_CRITICAL_CHAIN_KERNEL() (void)0
#endif

// finish an API call: synchronise unless the caller asked for async execution
inline eIcicleError end_call(hipStream_t s, bool is_async)
{
  if (!is_async) HIP_TRY(hipStreamSynchronize(s), ICICLE_SYNCHRONIZATION_FAILED);
  return ICICLE_SUCCESS;
}

inline eIcicleError check_launch(const char* what)
{
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error("kernel launch %s failed: %s", what, hipGetErrorString(e));
    return ICICLE_UNKNOWN_ERROR;
  }
  return ICICLE_SUCCESS;
}

} // namespace isnark
