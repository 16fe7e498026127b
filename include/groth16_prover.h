/*
 * groth16_prover.h — C API of the prover host that ships inside libicicle_snark_hip.so.
 *
 * The reference's host is Rust (src/lib.rs, src/proof_helper.rs, src/cache.rs); no Rust toolchain
 * exists in this build environment, so the same host logic is provided in C++ behind this C API,
 * with the reference's names and argument meaning:
 *
 *   groth16_prove(witness, zkey, proof, public, device, &mut CacheManager)   — src/lib.rs:33-61
 *   CacheManager::{compute, get_cache, insert_cache, contains}                — src/cache.rs:110-262
 *
 * File formats are snarkjs `.zkey` / `.wtns` in, `proof.json` / `public.json` out
 * (src/file_wrapper.rs:45-113, src/zkey.rs:47-85, src/conversions.rs:30-56).
 * All functions return 0 on success, an eIcicleError code (> 0) for device errors and a negative
 * value for I/O / format errors; groth16_last_error() gives the text.  Nothing unwinds across the ABI.
 */
#ifndef GROTH16_PROVER_H
#define GROTH16_PROVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Groth16CacheManager Groth16CacheManager;

/* CacheManager::default() / drop — src/cache.rs:110-115 */
Groth16CacheManager* groth16_cache_manager_new(void);
void groth16_cache_manager_free(Groth16CacheManager* cm);
/* Optional: create on `device_id`, on a helper thread, what the first cache load of a process would otherwise pay for inside
 * its first prove (six streams with their DMA queues: 48 ms measured, the pinned staging pool).  groth16_cache_manager_new
 * does this by itself when a device has been made current before it (icicle_set_device); the REPL calls it at start-up.
 * ICICLE_SNARK_PREWARM=0 switches it off. */
void groth16_cache_manager_prewarm(Groth16CacheManager* cm, int device_id);

/* groth16_prove — src/lib.rs:33-61.  `device` is the reference's free-form device string (the reference passes the type
 * and always uses id 0, src/lib.rs:25-31); this library registers "HIP" (and the alias "CUDA"); anything else, including
 * "CPU", is an error — there is no CPU fallback.  The string may name the devices to prove on:
 *     "HIP"  (device 0, or the list in ICICLE_SNARK_DEVICES)   "HIP:2"   "HIP:0-7"   "HIP:0,2,4,6"
 * More than one device = ONE prove sharded over a device group inside this process: point-range shards of
 * the A, B1, B2, C bases, residue-class shards of H, the QAP front end distributed, device-side exchanges over xGMI.  A
 * device may be named several times ("HIP:0,0,0,0": four shards on GPU 0 — how a 1-GPU machine exercises the path).
 * Blinding factors r, s are drawn at random (default build of the reference). */
int groth16_prove(const char* witness_path, const char* zkey_path, const char* proof_path, const char* public_path,
                  const char* device, Groth16CacheManager* cm);

/* ---- finer-grained entry points used by bench.py / tests (same pipeline, memory in / memory out) ---- */

/* Build (or find) the device-resident cache for a zkey image held in memory (CacheManager::compute,
 * src/cache.rs:117-241).  `key` plays the role of "{zkey_path}_{device}" (src/lib.rs:44).
 * shard_rank / shard_count: this process keeps only the points [rank·L/count, (rank+1)·L/count) of each of
 * the five MSM bases (multi-GPU point-range sharding); 0 / 1 for a single GPU. */
int groth16_cache_load(Groth16CacheManager* cm, const char* key, const void* zkey, size_t zkey_len, int device_id,
                       int shard_rank, int shard_count);
int groth16_cache_load_file(Groth16CacheManager* cm, const char* key, const char* zkey_path, int device_id,
                            int shard_rank, int shard_count);
/* A single-device key is usable as soon as its sections are on the device: the first proofs run the classic bucket layout
 * while a worker thread builds the key's fixed-base tables (13 instead of 16 digits per scalar; 0.26 s of GPU work at 1.6 M
 * constraints) on a low-priority stream, and the first prove that finds them complete adopts them.  Proofs are identical
 * either way.  Returns 1 when the key proves in its final layout, 0 while the build is under way, negative on error;
 * wait != 0 blocks until the build has ended.  ICICLE_SNARK_DEFER_TABLES=0 builds the tables inside groth16_cache_load. */
int groth16_cache_tables_ready(Groth16CacheManager* cm, const char* key, int wait);
/* The same key over a GROUP of devices (what groth16_prove builds for "HIP:a-b"): shard k of n_devices lives on
 * device_ids[k].  Every entry point that takes a key (groth16_commitments, groth16_prove_mem, groth16_prove_resident,
 * groth16_cache_info, groth16_last_timings, groth16_cache_evict) then works on the group: groth16_commitments returns the
 * SUM of the shards' commitments. */
int groth16_cache_load_devices(Groth16CacheManager* cm, const char* key, const void* zkey, size_t zkey_len, const int* device_ids,
                               int n_devices);
/* device string → ids ("HIP:0-7" → 0 … 7): returns how many devices it names (`ids` receives the first `cap`), or a
 * negative error code.  Needs no GPU. */
int groth16_parse_device(const char* device, int* ids, int cap);
/* Device-memory budget for the cached keys of ONE device (bytes; 0 = unlimited, the reference's behaviour — its
 * CacheManager never evicts, src/cache.rs:110-114).  Before a new key is built, the least recently used single-device
 * keys of that device are evicted until the new entry fits.  ICICLE_SNARK_CACHE_BUDGET_MB sets the initial value. */
void groth16_cache_set_budget(Groth16CacheManager* cm, uint64_t bytes_per_device);
int groth16_cache_contains(const Groth16CacheManager* cm, const char* key);
void groth16_cache_evict(Groth16CacheManager* cm, const char* key);

/* The five commitments of groth16_commitments (src/proof_helper.rs:172-241) for this process's shard,
 * as standard-form projective points in the order A (G1, 96 B), B1 (G1, 96 B), B2 (G2, 192 B),
 * C (G1, 96 B), H (G1, 96 B)  — 576 bytes.  Includes construct_r1cs (src/proof_helper.rs:31-170).
 * wtns == NULL re-uses the witness uploaded by the previous call for this key (inputs already resident in
 * HBM — what bench.py times); otherwise the witness is staged through pinned memory and uploaded first. */
#define GROTH16_COMMITMENTS_BYTES 576
typedef struct {
  double h2d_ms;       /* witness upload */
  double qap_ms;       /* construct_r1cs on the device (sparse mat-vec, 2 batched NTTs, pointwise) */
  double msm_ms;       /* the five MSMs (two streams) */
  double total_ms;     /* wall clock of the call */
} Groth16Timings;
int groth16_commitments(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len,
                        uint8_t out_points[GROTH16_COMMITMENTS_BYTES], Groth16Timings* timings /* may be NULL */);

/* Multi-GPU witness distribution.  Every rank of a sharded key needs the whole witness on its device (its MSM range, and
 * the rows of the QAP front end it evaluates read arbitrary wires).  Instead of shard_count full uploads over PCIe, rank r
 * uploads elements [r·slice, min(n_vars, (r+1)·slice)), slice = ⌈n_vars / shard_count⌉, to their place in its device
 * witness buffer; the caller completes the buffer with an IN-PLACE all-gather of `slice_bytes` bytes per rank over
 * `*d_witness` (world × slice_bytes bytes; RCCL over xGMI: icicle_snark_rccl_allgather_device) and then calls
 * groth16_witness_ready, after which groth16_commitments / groth16_dist_stage1 take wtns = NULL. */
int groth16_upload_witness_slice(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, void** d_witness, uint64_t* slice_bytes);
int groth16_witness_ready(Groth16CacheManager* cm, const char* key);

/* Distributed QAP front end for 2, 4 or 8 shards (H sharded by residue class; HISTORY.md §5, tests/dist_qap_model.py):
 * instead of replicating the spmv and the inverse transform on every rank, each rank transforms 1/count of the rows and
 * two all-to-alls move the blocks.  Sequence per prove, on every rank:
 *     groth16_dist_stage1(wtns)  →  all-to-all(send, recv)  →  groth16_dist_stage2()  →  all-to-all(send, recv)
 *     →  groth16_dist_exchange_done()  →  groth16_commitments(wtns = NULL)
 *                                                  (finishes with the size-n/count forward transform and the five MSMs)
 * The buffers are device memory owned by the cache entry: 3 rows of row_bytes; the chunk a rank exchanges with `peer` for
 * row `q` sits at q·row_bytes + peer·chunk_bytes in BOTH the send and the receive buffer.  groth16_dist_supported tells
 * whether the entry can take this path (otherwise groth16_commitments alone does everything, replicated). */
int groth16_dist_supported(Groth16CacheManager* cm, const char* key);
int groth16_dist_stage1(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, void** d_send,
                        void** d_recv, uint32_t* rows, uint64_t* row_bytes, uint64_t* chunk_bytes);
int groth16_dist_stage2(Groth16CacheManager* cm, const char* key, void** d_send, void** d_recv);
/* The caller confirms that the second all-to-all has DELIVERED into *d_recv of groth16_dist_stage2.  Only after this call
 * does groth16_commitments(wtns = NULL) finish from those rows; without it (an exchange that failed, a retry) the
 * commitments call recomputes the whole front end itself, replicated.  A new witness clears the confirmation. */
int groth16_dist_exchange_done(Groth16CacheManager* cm, const char* key);

/* Element-wise group sum of `count` commitment blocks (gathered from the shards): out = Σ_k blocks[k]. */
int groth16_sum_commitments(const uint8_t* blocks, int count, uint8_t out_points[GROTH16_COMMITMENTS_BYTES]);

/* Tail of groth16_prove_helper (src/proof_helper.rs:274-316): blinding with (r, s) — 32-byte little-endian
 * standard-form scalars; NULL draws them at random; r = s = 1 reproduces the `no-randomness` feature —
 * affine conversion and JSON rendering.  Outputs are NUL-terminated pretty-printed JSON texts identical
 * in layout to serde_json::to_writer_pretty.  Returns the needed size (incl. NUL) if a buffer is too small. */
int groth16_assemble_proof(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len,
                           const uint8_t points[GROTH16_COMMITMENTS_BYTES], const uint8_t* r, const uint8_t* s,
                           char* proof_json, size_t proof_cap, char* public_json, size_t public_cap);

/* One-GPU convenience: commitments + assemble. */
int groth16_prove_mem(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, const uint8_t* r,
                      const uint8_t* s, char* proof_json, size_t proof_cap, char* public_json, size_t public_cap,
                      Groth16Timings* timings);

/* Same, but with wtns_resident != 0 the witness uploaded by an earlier call is re-used on the device (the bytes are
 * still needed for the public signals): "inputs already resident in HBM", what bench.py times. */
int groth16_prove_resident(Groth16CacheManager* cm, const char* key, const void* wtns, size_t wtns_len, int wtns_resident,
                           const uint8_t* r, const uint8_t* s, char* proof_json, size_t proof_cap, char* public_json,
                           size_t public_cap, Groth16Timings* timings);

/* sizes of the cached circuit */
typedef struct {
  uint32_t n_vars, n_public, domain_size, n_coef;
  uint64_t device_bytes;
  uint32_t b_bases;  /* bases the two B MSMs run over (= the wires of this shard; kept for layout compatibility) */
  uint32_t shards;   /* device group: number of shards (device_bytes and b_bases are sums over them); else 0 */
} Groth16CircuitInfo;
int groth16_cache_info(const Groth16CacheManager* cm, const char* key, Groth16CircuitInfo* info);
/* the same for a caller that passes sizeof(its Groth16CircuitInfo): fields beyond info_size are not written, so a binary
 * built against an older, shorter struct keeps working when the struct grows */
int groth16_cache_info_sized(const Groth16CacheManager* cm, const char* key, void* info, size_t info_size);

/* What the key runs on, as one line of JSON — for a device group {"shards": G, "devices": [...], "distinct_devices": k,
 * "transport": "pull" | "memcpy" | "rccl", "peer_access": bool, "rccl_ranks": n (0 unless the rccl transport moves the
 * exchanges), "distributed_front_end": bool, "transport_forced_by_env": bool}; a single-device key answers "shards": 0.
 * Returns 0, or the size needed (incl. NUL) when `cap` is too small.  bench.py prints it with every multi-GPU line. */
int groth16_group_describe(const Groth16CacheManager* cm, const char* key, char* out, size_t cap);

/* phase timings (HIP events) of the most recent prove of `key` through ANY entry point — groth16_prove returns none,
 * like the reference's; bench.py reads them here.  ICICLE_SNARK_QUIET=1 suppresses groth16_prove's "proof took: …" line
 * (src/lib.rs:58) for callers whose stdout is machine-read. */
int groth16_last_timings(Groth16CacheManager* cm, const char* key, Groth16Timings* timings);

const char* groth16_last_error(void);

/* groth16_verify — src/lib.rs:63-82 with groth16_verify_helper (src/proof_helper.rs:319-372) and the snarkjs
 * verification_key.json reader (src/cache.rs:74-108).  Checks e(−A,B)·e(IC₀+Σ pubᵢ·ICᵢ₊₁, γ₂)·e(C,δ₂)·e(α₁,β₂) = 1
 * with four host pairings.  groth16_verify (paths): 0 = accepted, 1 = "Verification failed" (the reference
 * asserts), negative = I/O or format error.  groth16_verify_json (texts): 1 = accepted, 0 = rejected, negative =
 * format error.  No device is needed. */
int groth16_verify(const char* proof_path, const char* public_path, const char* vk_path);
int groth16_verify_json(const char* proof_json, const char* public_json, const char* vk_json);
const char* groth16_verify_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
