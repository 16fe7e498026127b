// the FAST PATH of a G1 mixed addition on the lazy radix-2^29 field, alone in a kernel, for an instruction histogram per opcode class
// (scratch/isa_hist.py): load one packed affine base (internal encoding), conditional negation, the ten field products of x_madd
// (ec29.h) without its rare exact branches (identity accumulator, doubling, cancellation).  Not run; compiled with -S only.
#include "ec29.h"
using namespace bn254;
__global__ void madd_fast_kernel(const G1::A* __restrict__ bases, const uint32_t* __restrict__ idx, uint32_t n, G1::X* out)
{
  typedef Fq29 F;
  typedef G1L CL;
  CL::X acc = CL::x_load_internal(out[threadIdx.x]);
#pragma unroll 1
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t e = idx[k * 64 + threadIdx.x];
    const CL::A b = CL::load_affine(bases[e & 0x7fffffffu], 2, (e >> 31) != 0);
    const fe9 U2 = F::mul(b.x, acc.zz);
    const fe9 S2 = F::mul(b.y, acc.zzz);
    const fe9 Pn = F::norm(F::subx(U2, acc.x));
    const fe9 Rn = F::norm(F::sub3(S2, acc.y));
    const fe9 PP = F::sqr_n(Pn);
    const fe9 PPP = F::mul(Pn, PP);
    const fe9 Q = F::mul(acc.x, PP);
    const fe9 RR = F::sqr_n(Rn);
    const fe9 X3 = F::x3(RR, PPP, Q);
    const fe9 D = F::subx(Q, X3);
    acc.y = F::y3(Rn, D, acc.y, PPP);
    acc.x = X3;
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
  }
  out[threadIdx.x] = CL::x_store_internal(acc);
}
