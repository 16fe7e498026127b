// ntt_fuse.h — internal entry points of ntt.hip for the prover (prover/prover.cpp): the two batched transforms of
// construct_r1cs (src/proof_helper.rs:116,145) with the pointwise work around them folded into their last passes.
#pragma once
#include "common.h"
#include "ff.h"

namespace isnark {

struct NttFuse {
  // inverse transform: out[i] *= scale_tab[i·scale_stride] instead of n⁻¹ — with tab[i] = n⁻¹·g^i the three coset
  // multiplications of src/proof_helper.rs:121-141 cost nothing (ntt_build_scaled_keys)
  const bn254::fe* scale_tab = nullptr;
  uint32_t scale_stride = 1;
  // forward transform of the rows [B | A | C'] (batch 3): write A·B − C' (src/proof_helper.rs:154-167) here (n elements)
  // instead of storing the three transformed rows
  bn254::fe* fused_out = nullptr;
};

// in-place transform of `batch` rows of n elements at d_inout (device, natural order), asynchronous on s
eIcicleError ntt_fused(bn254::fe* d_inout, uint32_t n, int batch, bool inverse, hipStream_t s, const NttFuse& fuse);
// every pass of a size-n transform works on full 2048-element tiles (precondition of NttFuse::fused_out)
bool ntt_fusable(uint32_t n);
// d_tab[i] = n⁻¹·ω_2n^i (Montgomery), i < n, from the current domain (which must hold ≥ 2n roots)
eIcicleError ntt_build_scaled_keys(uint32_t n, bn254::fe* d_tab, hipStream_t s);

} // namespace isnark
