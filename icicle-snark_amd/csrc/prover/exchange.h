// exchange.h — pull kernels of the in-process multi-GPU exchanges (exchange.hip), internal API.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace isnark {

constexpr uint32_t XCHG_MAX_PEERS = 16;
struct PeerPtrs {
  const void* p[XCHG_MAX_PEERS]; // one buffer per rank (device memory of that rank's GPU, mapped for peer access)
};

// in-place all-gather: bufs.p[k] = rank k's buffer of G slices; rank `me` fills its slices p ≠ me from slice p of bufs.p[p]
hipError_t xchg_allgather_pull(const PeerPtrs& bufs, uint32_t G, uint32_t me, size_t slice_bytes, hipStream_t s);
// all-to-all of [row][peer][chunk] buffers: recv[q][p] = sends.p[p][q][me]
hipError_t xchg_alltoall_pull(const PeerPtrs& sends, void* recv, uint32_t G, uint32_t me, uint32_t rows, size_t row_bytes, size_t chunk_bytes, hipStream_t s);

// a single-lane kernel that occupies stream `s` for about `ms` milliseconds (exchange self-test: makes the producers late)
hipError_t xchg_delay(double ms, hipStream_t s);

} // namespace isnark
