// rccl_comm.cpp — the one exchange step of the multi-GPU prover: every rank holds the
// five partial commitments of its point-range shard (576 B); an RCCL all-gather over xGMI gives every
// rank all blocks, which are then summed with the curve's group law (RCCL has no EC-add reduction op, so
// this is an all-gather of raw bytes + local adds, not an all-reduce).  Built as a separate small DSO
// (libicicle_snark_rccl.so) so that the single-GPU library carries no RCCL dependency.
//
// Two process models:
//  * one process per GPU (torch.distributed / torchrun provides rank, world size and the rendezvous used to broadcast the
//    ncclUniqueId): icicle_snark_rccl_init + the host-synchronous calls below, driven from parallel.py;
//  * ONE process that drives a group of GPUs (groth16_prove with the device string "HIP:0-7", prover/multi.cpp):
//    icicle_snark_rccl_init_all (ncclCommInitAll) + the *_on calls, which only ENQUEUE the collective on the caller's
//    stream — the prover's own — so that it is ordered with the kernels before and after it without any host wait.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define API extern "C" __attribute__((visibility("default")))

namespace {
thread_local char g_err[256] = "";
struct Comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  uint8_t* d_in = nullptr;
  uint8_t* d_out = nullptr;
  uint8_t* h_pin = nullptr;
  size_t cap = 0;
  int world = 1, device = 0;
};
int fail(const char* what, int code)
{
  snprintf(g_err, sizeof g_err, "%s failed (%d)", what, code);
  return code ? code : -1;
}
} // namespace

API const char* icicle_snark_rccl_last_error(void) { return g_err; }
API int icicle_snark_rccl_destroy(void* comm);

// rank 0 creates the id (128 bytes) and shares it with the other ranks out of band
API int icicle_snark_rccl_unique_id(uint8_t out[NCCL_UNIQUE_ID_BYTES])
{
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", (int)r);
  memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

API int icicle_snark_rccl_init(const uint8_t id_bytes[NCCL_UNIQUE_ID_BYTES], int rank, int world, int device_id, size_t max_bytes_per_rank, void** out)
{
  if (hipSetDevice(device_id) != hipSuccess) return fail("hipSetDevice", -1);
  Comm* c = new Comm();
  c->world = world;
  c->device = device_id;
  c->cap = max_bytes_per_rank;
  ncclUniqueId id;
  memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return fail("ncclCommInitRank", (int)r); }
  // any failure from here on releases what was created so far (communicator, stream, buffers)
  const char* what = nullptr;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) what = "hipStreamCreate";
  else if (hipMalloc((void**)&c->d_in, max_bytes_per_rank) != hipSuccess) what = "hipMalloc(d_in)";
  else if (hipMalloc((void**)&c->d_out, max_bytes_per_rank * world) != hipSuccess) what = "hipMalloc(d_out)";
  else if (hipHostMalloc((void**)&c->h_pin, max_bytes_per_rank * (world + 1)) != hipSuccess) what = "hipHostMalloc";
  if (what) {
    (void)icicle_snark_rccl_destroy(c);
    return fail(what, -1);
  }
  *out = c;
  return 0;
}

// all-gather `bytes` bytes per rank: host block in → host blocks out (rank order)
API int icicle_snark_rccl_allgather(void* comm, const void* in, size_t bytes, void* out)
{
  Comm* c = (Comm*)comm;
  if (!c || bytes > c->cap) return fail("allgather: bad arguments", -1);
  if (hipSetDevice(c->device) != hipSuccess) return fail("hipSetDevice", -1); // the active device is per thread
  memcpy(c->h_pin, in, bytes);
  if (hipMemcpyAsync(c->d_in, c->h_pin, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail("H2D", -1);
  ncclResult_t r = ncclAllGather(c->d_in, c->d_out, bytes, ncclUint8, c->comm, c->stream);
  if (r != ncclSuccess) return fail("ncclAllGather", (int)r);
  if (hipMemcpyAsync(c->h_pin + c->cap, c->d_out, bytes * c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return fail("D2H", -1);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return fail("sync", -1);
  memcpy(out, c->h_pin + c->cap, bytes * c->world);
  return 0;
}

// all-to-all of DEVICE buffers for the distributed QAP front end (include/groth16_prover.h: groth16_dist_stage1/2): for every
// row q < rows and every peer p the chunk at q·row_bytes + p·chunk_bytes of `d_send` goes to rank p, and what rank p sent for
// this rank lands at the same offset of `d_recv` (both buffers are [row][peer][chunk]).  One grouped ncclSend / ncclRecv
// batch — point-to-point over the xGMI links, which is what an all-to-all is on this fabric; 3·(n/G)·32·(G−1)/G bytes per
// rank and exchange (22 MB at n = 2^21, G = 8).
API int icicle_snark_rccl_alltoall_rows_on(void* comm, const void* d_send, void* d_recv, int rows, size_t row_bytes, size_t chunk_bytes, void* stream)
{
  Comm* c = (Comm*)comm;
  if (!c || !d_send || !d_recv || rows < 1) return fail("alltoall: bad arguments", -1);
  if (hipSetDevice(c->device) != hipSuccess) return fail("hipSetDevice", -1);
  hipStream_t st = (hipStream_t)stream;
  int me = 0;
  ncclResult_t r = ncclCommUserRank(c->comm, &me);
  if (r != ncclSuccess) return fail("ncclCommUserRank", (int)r);
  r = ncclGroupStart();
  if (r != ncclSuccess) return fail("ncclGroupStart", (int)r);
  for (int q = 0; q < rows && r == ncclSuccess; q++)
    for (int p = 0; p < c->world && r == ncclSuccess; p++) {
      const size_t off = (size_t)q * row_bytes + (size_t)p * chunk_bytes;
      if (p == me) {
        if (hipMemcpyAsync((uint8_t*)d_recv + off, (const uint8_t*)d_send + off, chunk_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) r = ncclSystemError;
        continue;
      }
      r = ncclSend((const uint8_t*)d_send + off, chunk_bytes, ncclUint8, p, c->comm, st);
      if (r == ncclSuccess) r = ncclRecv((uint8_t*)d_recv + off, chunk_bytes, ncclUint8, p, c->comm, st);
    }
  const ncclResult_t r2 = ncclGroupEnd();
  if (r != ncclSuccess) return fail("ncclSend/ncclRecv", (int)r);
  if (r2 != ncclSuccess) return fail("ncclGroupEnd", (int)r2);
  return 0;
}
API int icicle_snark_rccl_alltoall_rows(void* comm, const void* d_send, void* d_recv, int rows, size_t row_bytes, size_t chunk_bytes)
{
  Comm* c = (Comm*)comm;
  if (!c) return fail("alltoall: bad arguments", -1);
  if (int rc = icicle_snark_rccl_alltoall_rows_on(comm, d_send, d_recv, rows, row_bytes, chunk_bytes, c->stream)) return rc;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return fail("sync", -1);
  return 0;
}

// in-place all-gather of a DEVICE buffer of world × slice_bytes bytes: this rank's slice sits at d_buf + rank·slice_bytes
// (the witness of a sharded prove: every rank uploads 1/world of it over PCIe and the rest arrives over xGMI —
// include/groth16_prover.h: groth16_upload_witness_slice)
API int icicle_snark_rccl_allgather_device_on(void* comm, void* d_buf, size_t slice_bytes, void* stream)
{
  Comm* c = (Comm*)comm;
  if (!c || !d_buf || slice_bytes == 0) return fail("allgather_device: bad arguments", -1);
  if (hipSetDevice(c->device) != hipSuccess) return fail("hipSetDevice", -1);
  int me = 0;
  ncclResult_t r = ncclCommUserRank(c->comm, &me);
  if (r != ncclSuccess) return fail("ncclCommUserRank", (int)r);
  r = ncclAllGather((const uint8_t*)d_buf + (size_t)me * slice_bytes, d_buf, slice_bytes, ncclUint8, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail("ncclAllGather (device, in place)", (int)r);
  return 0;
}
API int icicle_snark_rccl_allgather_device(void* comm, void* d_buf, size_t slice_bytes)
{
  Comm* c = (Comm*)comm;
  if (!c) return fail("allgather_device: bad arguments", -1);
  if (int rc = icicle_snark_rccl_allgather_device_on(comm, d_buf, slice_bytes, c->stream)) return rc;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return fail("sync", -1);
  return 0;
}

// One process, n devices: ncclCommInitAll.  comms_out[k] is rank k's communicator (on device devs[k]) for the *_on calls; it
// owns no stream and no staging buffers.  The devices must be distinct (RCCL refuses duplicates).
API int icicle_snark_rccl_init_all(int n, const int* devs, void** comms_out)
{
  if (n < 1 || n > 64 || !devs || !comms_out) return fail("init_all: bad arguments", -1);
  ncclComm_t cs[64];
  ncclResult_t r = ncclCommInitAll(cs, n, devs);
  if (r != ncclSuccess) return fail("ncclCommInitAll", (int)r);
  for (int k = 0; k < n; k++) {
    Comm* c = new Comm();
    c->comm = cs[k];
    c->world = n;
    c->device = devs[k];
    comms_out[k] = c;
  }
  return 0;
}

// max over ranks of one double (used for the benchmark's max-over-ranks timing)
API int icicle_snark_rccl_allreduce_max(void* comm, double* value)
{
  Comm* c = (Comm*)comm;
  if (!c) return fail("allreduce: null comm", -1);
  if (hipSetDevice(c->device) != hipSuccess) return fail("hipSetDevice", -1); // the active device is per thread
  memcpy(c->h_pin, value, 8);
  if (hipMemcpyAsync(c->d_in, c->h_pin, 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail("H2D", -1);
  ncclResult_t r = ncclAllReduce(c->d_in, c->d_out, 1, ncclDouble, ncclMax, c->comm, c->stream);
  if (r != ncclSuccess) return fail("ncclAllReduce", (int)r);
  if (hipMemcpyAsync(c->h_pin, c->d_out, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return fail("D2H", -1);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return fail("sync", -1);
  memcpy(value, c->h_pin, 8);
  return 0;
}

API int icicle_snark_rccl_destroy(void* comm)
{
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->d_in) (void)hipFree(c->d_in);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}
