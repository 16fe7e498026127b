// common.h — shared plumbing of the C-ABI implementation (error handling, staging of host operands).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

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

// Stages a host-resident operand on the device for the lifetime of the object (the reference's
// wrappers do the same per VecOpsConfig / NTTConfig / MSMConfig flags, e.g.
// icicle/backend/cuda/src/field/cuda_vec_ops.cu:17-54).  Output operands are copied back by finish().
class Staged
{
public:
  Staged() {}
  ~Staged() { release(); }
  // input operand
  eIcicleError in(const void* p, size_t bytes, bool on_device, hipStream_t s)
  {
    stream_ = s;
    if (on_device || bytes == 0) {
      dev_ = const_cast<void*>(p);
      return ICICLE_SUCCESS;
    }
    HIP_TRY(hipMallocAsync(&dev_, bytes, s), ICICLE_ALLOCATION_FAILED);
    owned_ = true;
    HIP_TRY(hipMemcpyAsync(dev_, p, bytes, hipMemcpyHostToDevice, s), ICICLE_COPY_FAILED);
    return ICICLE_SUCCESS;
  }
  // output operand (alias_of: if the output host pointer equals an input host pointer reuse its staging)
  eIcicleError out(void* p, size_t bytes, bool on_device, hipStream_t s)
  {
    stream_ = s;
    host_out_ = nullptr;
    if (on_device || bytes == 0) {
      dev_ = p;
      return ICICLE_SUCCESS;
    }
    HIP_TRY(hipMallocAsync(&dev_, bytes, s), ICICLE_ALLOCATION_FAILED);
    owned_ = true;
    host_out_ = p;
    bytes_ = bytes;
    return ICICLE_SUCCESS;
  }
  eIcicleError finish()
  {
    if (host_out_) {
      HIP_TRY(hipMemcpyAsync(host_out_, dev_, bytes_, hipMemcpyDeviceToHost, stream_), ICICLE_COPY_FAILED);
      host_out_ = nullptr;
    }
    return ICICLE_SUCCESS;
  }
  void release()
  {
    if (owned_ && dev_) (void)hipFreeAsync(dev_, stream_);
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
