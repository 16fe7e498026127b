// Which streams share a hardware pipe?  A kernel with far more workgroups than the GPU holds keeps the pipe of its queue busy
// dispatching for as long as it runs; a one-workgroup kernel on another stream finishes at once unless its queue sits on the same pipe.
// Prints, for n streams created back to back, the delay (µs) of the tiny kernel on stream j while the big one runs on stream i.
// build: hipcc --offload-arch=gfx950 -O2 -o build/pipe_probe scratch/pipe_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
__global__ void big(unsigned long long* out, int iters)
{
  unsigned long long x = threadIdx.x + blockIdx.x;
  for (int i = 0; i < iters; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
  if (x == 42) out[0] = x;
}
__global__ void tiny(unsigned long long* out) { if (threadIdx.x == 999) out[1] = 1; }
int main(int argc, char** argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 10;
  std::vector<hipStream_t> st(n);
  unsigned long long* d;
  hipMalloc(&d, 64);
  for (int i = 0; i < n; i++) {
    hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[i], d); // first use: the stream gets its queue now
    hipStreamSynchronize(st[i]);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("GPU_MAX_HW_QUEUES=%s; rows: stream of the big kernel, columns: stream of the tiny one; delay of the tiny kernel in us\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
  for (int i = 0; i < n; i++) {
    printf("big on %2d:", i);
    for (int j = 0; j < n; j++) {
      if (i == j) { printf("      -"); continue; }
      hipLaunchKernelGGL(big, dim3(40000), dim3(256), 0, st[i], d, 3000);
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[j], d);
      hipStreamSynchronize(st[j]);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      hipStreamSynchronize(st[i]);
      printf(" %6.0f", us);
    }
    printf("\n");
  }
  return 0;
}
