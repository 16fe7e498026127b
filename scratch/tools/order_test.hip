// Do two streams of one device overlap, so that a consumer without an event wait reads a late producer's buffer too early?  (what the
// exchange self-test of csrc/prover/multi.cpp relies on across devices; on ONE device the prover streams of aliased shards share hardware
// queues and serialise, so the negative case cannot be shown inside the library on a 1-GPU box)
// build: hipcc -O2 -std=c++17 --offload-arch=gfx950 -o scratch/tools/order_test scratch/tools/order_test.hip
#include "../../icicle-snark_amd/csrc/prover/exchange.hip"
#include <chrono>
#include <cstdio>
#include <vector>
int main()
{
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  const int NS = 14;
  std::vector<hipStream_t> st(NS);
  for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  uint8_t *buf, *out; hipMalloc((void**)&buf, 4096); hipMalloc((void**)&out, 4096);
  for (int other : {1, 6, 7, 13}) {
    hipMemset(buf, 0, 4096); hipMemset(out, 0xff, 4096); hipDeviceSynchronize();
    hipStream_t s0 = st[0], s1 = st[other];
    isnark::xchg_delay(4.0, s1); hipMemsetAsync(buf, 1, 4096, s1);
    isnark::xchg_delay(2.0, s0);
    isnark::PeerPtrs pp = {}; pp.p[0] = out; pp.p[1] = buf;
    // "rank 0" pulls slice 1 (2048 bytes) of rank 1's buffer into its own
    isnark::xchg_allgather_pull(pp, 2, 0, 2048, s0);
    auto t0 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s0);
    double t_s0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    uint8_t h[4096]; hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
    hipDeviceSynchronize();
    printf("streams 0 and %d: s0 done after %.2f ms, pulled byte = %d (0 = read before the producer: streams overlap; 1 = serialised)\n", other, t_s0, h[2048]);
  }
  return 0;
}
