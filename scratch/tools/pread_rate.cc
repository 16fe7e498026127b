// How fast can k threads pread() a page-cache-resident 51 MB file into (a) malloc'd, (b) hipHostMalloc'd buffers, in 2 MB chunks,
// and how fast does the DMA of the same chunks go?  (the witness upload of a file-to-file prove: scratch/README.md)
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
  const size_t N = 51200000, CH = 2u << 20;
  const char* path = "/tmp/pread_rate.bin";
  {
    std::vector<char> v(N, 1);
    for (size_t i = 0; i < N; i += 4096) v[i] = (char)rand();
    FILE* f = fopen(path, "wb"); fwrite(v.data(), 1, N, f); fclose(f);
  }
  int fd = open(path, O_RDONLY);
  char* pinned; char* plain = (char*)malloc(16 * 2 * CH);
  if (hipHostMalloc((void**)&pinned, 16 * 2 * CH, hipHostMallocPortable) != hipSuccess) return 1;
  memset(pinned, 0, 16 * 2 * CH); memset(plain, 0, 16 * 2 * CH);
  char* dev; hipMalloc((void**)&dev, N);
  const size_t nch = (N + CH - 1) / CH;
  for (int mode = 0; mode < 3; mode++)
    for (int k : {1, 2, 3, 4, 6, 8}) {
      double best = 1e9;
      for (int rep = 0; rep < 7; rep++) {
        std::atomic<size_t> next{0};
        std::vector<hipStream_t> st(k);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const double t0 = now();
        auto w = [&](int t) {
          char* b = (mode == 0 ? plain : pinned) + (size_t)t * 2 * CH;
          hipEvent_t ev[2]; bool used[2] = {false, false};
          if (mode == 2) { hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); hipEventCreateWithFlags(&ev[1], hipEventDisableTiming); }
          for (int kk = 0;; kk ^= 1) {
            size_t i = next.fetch_add(1);
            if (i >= nch) break;
            const size_t n = N - i * CH < CH ? N - i * CH : CH;
            if (mode == 2 && used[kk]) hipEventSynchronize(ev[kk]);
            size_t got = 0;
            while (got < n) got += pread(fd, b + kk * CH + got, n - got, i * CH + got);
            if (mode == 2) { hipMemcpyAsync(dev + i * CH, b + kk * CH, n, hipMemcpyHostToDevice, st[t]); hipEventRecord(ev[kk], st[t]); used[kk] = true; }
          }
          if (mode == 2) hipStreamSynchronize(st[t]);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < k; t++) th.emplace_back(w, t);
        w(0);
        for (auto& x : th) x.join();
        const double dt = now() - t0;
        if (dt < best) best = dt;
        for (auto& s : st) hipStreamDestroy(s);
      }
      printf("%s threads %d: %.3f ms  %.1f GB/s\n", mode == 0 ? "pread->malloc " : mode == 1 ? "pread->pinned " : "pread+DMA     ", k, best, N / best / 1e6);
    }
  // one DMA from a fully pinned copy
  char* all; hipHostMalloc((void**)&all, N, hipHostMallocPortable); memset(all, 1, N);
  hipStream_t s; hipStreamCreate(&s);
  for (int rep = 0; rep < 3; rep++) { const double t0 = now(); hipMemcpyAsync(dev, all, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("single DMA of 51 MB from pinned: %.3f ms\n", now() - t0); }
  return 0;
}
