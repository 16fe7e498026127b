// msm_g2_acc.hip — the G2 bucket-accumulation kernel in its own translation unit, compiled with the Fq2
// multiply/square INLINED (ISNARK_FQ2_INLINE): this is the one G2 kernel whose speed matters (≈35 % of the MSM
// work of a prove), and out-of-line Fq2 calls cost it register shuffles and ~0.8 KB/lane of scratch traffic.
#define ISNARK_FQ2_INLINE 1
#include "msm_impl.h"

namespace isnark {
void msm_g2_accumulate_launch(const SortPlan* pl, const void* d_points, int points_mont, uint32_t skip_below, uint32_t stride, hipStream_t s, void* buckets, int into, bool resident)
{
  AccumulateLauncher<G2>::launch(pl, (const G2::A*)d_points, points_mont, skip_below, stride, s, (G2::X*)buckets, into, resident);
}
} // namespace isnark

// first launch of a translation unit's code object loads it onto the device (milliseconds): prewarm_modules (runtime.cpp) does that ahead
// of the first prove of a process
namespace isnark {
__global__ void module_warm_g2acc_kernel() {}
void module_warm_g2acc(hipStream_t s) { hipLaunchKernelGGL(module_warm_g2acc_kernel, dim3(1), dim3(1), 0, s); }
} // namespace isnark
