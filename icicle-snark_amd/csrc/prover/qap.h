// qap.h — device kernels of the QAP front end (construct_r1cs, src/proof_helper.rs:31-170), internal API.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../ff.h"

namespace isnark {

// d_vec = [B | A | A∘B] evaluations on the size-n domain from the witness:
//   row r of the CSR (rowptr/cols/vals) is target index c + m·n of src/proof_helper.rs:81-92
//   (rows [0,n): A (m = 0), rows [n,2n): B (m = 1)); vals are Montgomery-form coefficients.
// Replaces the host gather (:55-61), the pointwise multiply (:75) and the serial host scatter-add (:81-92).
hipError_t qap_spmv(const bn254::fe* witness, const uint32_t* rowptr, const uint32_t* cols, const bn254::fe* vals, uint32_t n, bn254::fe* d_vec, hipStream_t s);
// d_vec[k·n + i] *= keys[i·key_stride]  for k = 0,1,2 (keys Montgomery form) — src/proof_helper.rs:121-141
hipError_t qap_coset_mul3(bn254::fe* d_vec, const bn254::fe* keys, uint32_t key_stride, uint32_t n, hipStream_t s);
// slot1 = slot0∘slot1 − slot2 — src/proof_helper.rs:154-167 (slot0 is left untouched)
hipError_t qap_final(bn254::fe* d_vec, uint32_t n, hipStream_t s);

// cold path (csr.hip): CSR of zkey section 4 built on the device from the raw 44-byte records
// {m:u32 c:u32 s:u32 value[32 B]} (src/cache.rs:126-166); vals come out as Montgomery-form coefficients (:214).
// rowptr has 2n+1 entries.  *first_bad = index of the first out-of-range record, 0xffffffff if none.  Synchronises `s`.
hipError_t qap_build_csr(const uint32_t* d_records, uint32_t n_coef, uint32_t n, uint32_t n_vars, uint32_t* d_rowptr, uint32_t* d_cols, bn254::fe* d_vals,
                         uint32_t* first_bad, hipStream_t s);

} // namespace isnark
