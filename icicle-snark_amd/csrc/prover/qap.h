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

// Multi-GPU shards (power-of-two count G, rank r): the rank needs the coset evaluations only at k ≡ r (mod G).  With
// m = n/G and j = j' + t·m:  X[r + G·k'] = Σ_{j'<m} ω_m^{j'k'} · z[j'],  z[j'] = Σ_{t<G} x[j' + t·m] · ω_2n^e,
// e = (j' + t·m) + 2m·(t·r mod G) + 2·(j'·r mod n)   (coset key g^j, ω_G^{tr} and ω_n^{j'r} in one table look-up).
// Writes the three folded rows [3][m] to `out`; a size-m forward NTT (batch 3) of `out` then gives the rank's slice at
// the cost of one multiplication per coefficient instead of a full size-n transform.  tw = ω_N^i table, N = 2n·tw_scale.
hipError_t qap_coset_fold3(const bn254::fe* d_vec, const bn254::fe* tw, uint32_t tw_scale, uint32_t n, uint32_t G, uint32_t r, bn254::fe* out, hipStream_t s);
// dst[k] = src[first + k·stride] for elements of `elem_fe` field elements (strided point-range shard of the H bases)
hipError_t qap_gather_strided(const bn254::fe* src, bn254::fe* dst, uint32_t elem_fe, uint32_t count, uint32_t stride, uint32_t first, hipStream_t s);

// Distributed front end for G ∈ {2, 4, 8} GPUs (tests/dist_qap_model.py has the algebra and its CPU restatement):
//   stage 1  qap_spmv_strided (rows c ≡ r mod G → [B | A | A∘B] over m = n/G elements) + size-m inverse transform whose
//            per-element scale is the table of qap_dist_tw1 (n⁻¹·ω_n^{−r·k2}); exchange 1 = all-to-all of blocks of m/G;
//   stage 2  qap_dist_mid: size-G inverse DFT over the sources, coset key g^k, size-G forward DFT, twist ω_n^{k2·i1};
//            exchange 2 = all-to-all of blocks of m/G;  stage 3 = size-m forward transform with the A·B − C epilogue.
// tw = ω_N^i table of the NTT domain, N ≥ 2n.  Buffers of stage 2 are [row][peer][m/G] (= [row][m]).
hipError_t qap_spmv_strided(const bn254::fe* witness, const uint32_t* rowptr, const uint32_t* cols, const bn254::fe* vals, uint32_t n, uint32_t G, uint32_t r, bn254::fe* out, hipStream_t s);
hipError_t qap_dist_tw1(const bn254::fe* tw, uint32_t N, uint32_t n, uint32_t G, uint32_t r, bn254::fe* tab, hipStream_t s);
hipError_t qap_dist_mid(const bn254::fe* recv, bn254::fe* send, const bn254::fe* tw, uint32_t N, uint32_t n, uint32_t G, uint32_t b, hipStream_t s);

// cold path (csr.hip): CSR of zkey section 4 built on the device from the raw 44-byte records
// {m:u32 c:u32 s:u32 value[32 B]} (src/cache.rs:126-166); vals come out as Montgomery-form coefficients (:214).
// rowptr has 2n+1 entries.  *first_bad = index of the first out-of-range record, 0xffffffff if none.  Synchronises `s`.
hipError_t qap_build_csr(const uint32_t* d_records, uint32_t n_coef, uint32_t n, uint32_t n_vars, uint32_t* d_rowptr, uint32_t* d_cols, bn254::fe* d_vals,
                         uint32_t* first_bad, hipStream_t s);

} // namespace isnark
