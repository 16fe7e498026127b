/*
 * icicle_snark_hip.h — C ABI of libicicle_snark_hip.so (MI355X / gfx950).
 *
 * The entry points below are exactly the `extern "C"` symbols the reference's Rust host reaches
 * through its FFI wrappers for the Groth16 hot path.  Names, argument order,
 * struct layouts and error codes are the reference's, so `wrappers/rust/icicle-runtime`,
 * `icicle-core` and `icicle-bn254` link against this library unchanged.
 * Every declaration cites the reference interface it replaces.
 *
 * Conventions (all from the reference):
 *  - field elements: 8×u32 little-endian limbs, STANDARD form unless a config flag says Montgomery
 *    (wrappers/rust/icicle-core/src/field.rs:10-15);
 *  - G1 affine {x,y} 64 B, projective {x,y,z} 96 B; G2 affine 128 B ({x.c0,x.c1,y.c0,y.c1}),
 *    projective 192 B; affine identity (0,0), projective identity (0,1,0)
 *    (wrappers/rust/icicle-core/src/curve.rs:45-59,104-111);
 *  - every function returns eIcicleError (int32, 0 = success) and never unwinds across the ABI;
 *  - host/device residency of each buffer is declared by the config flags, never probed;
 *  - with is_async the call only enqueues on `stream`; the caller synchronises.
 *  - the active device is thread-local (icicle/src/device_api.cpp:87-116); the only device type
 *    registered by this library is "HIP" ("CUDA" is accepted as an alias so that existing
 *    `--device CUDA` command lines keep working).  "CPU" is NOT provided: there is no CPU fallback.
 */
#ifndef ICICLE_SNARK_HIP_H
#define ICICLE_SNARK_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* icicle/include/icicle/errors.h:13-29 ; wrappers/rust/icicle-runtime/src/errors.rs:3-20 */
typedef enum {
  ICICLE_SUCCESS = 0,
  ICICLE_INVALID_DEVICE,
  ICICLE_OUT_OF_MEMORY,
  ICICLE_INVALID_POINTER,
  ICICLE_ALLOCATION_FAILED,
  ICICLE_DEALLOCATION_FAILED,
  ICICLE_COPY_FAILED,
  ICICLE_SYNCHRONIZATION_FAILED,
  ICICLE_STREAM_CREATION_FAILED,
  ICICLE_STREAM_DESTRUCTION_FAILED,
  ICICLE_API_NOT_IMPLEMENTED,
  ICICLE_INVALID_ARGUMENT,
  ICICLE_BACKEND_LOAD_FAILED,
  ICICLE_LICENSE_CHECK_ERROR,
  ICICLE_UNKNOWN_ERROR
} eIcicleError;

/* icicle/include/icicle/device.h:14-16 */
typedef struct {
  char type[64];
  int id;
} IcicleDevice;

typedef void* icicleStreamHandle; /* = hipStream_t */
typedef struct ConfigExtension ConfigExtension;

typedef struct { uint32_t limbs[8]; } bn254_scalar_t; /* Fr */
typedef struct { uint32_t limbs[8]; } bn254_fq_t;     /* Fq */
typedef struct { bn254_fq_t x, y; } bn254_affine_t;
typedef struct { bn254_fq_t x, y, z; } bn254_projective_t;
typedef struct { bn254_fq_t c0, c1; } bn254_fq2_t;
typedef struct { bn254_fq2_t x, y; } bn254_g2_affine_t;
typedef struct { bn254_fq2_t x, y, z; } bn254_g2_projective_t;
typedef struct { bn254_fq2_t c[2][3]; } bn254_fq12_t; /* PairingConfig::TargetField: c[i][j] = coefficient of v^j w^i */

/* icicle/include/icicle/msm.h:21-53 == wrappers/rust/icicle-core/src/msm/mod.rs:13-49 */
typedef struct {
  icicleStreamHandle stream;
  int precompute_factor;
  int c;
  int bitsize;
  int batch_size;
  bool are_points_shared_in_batch;
  bool are_scalars_on_device;
  bool are_scalars_montgomery_form;
  bool are_points_on_device;
  bool are_points_montgomery_form;
  bool are_results_on_device;
  bool is_async;
  ConfigExtension* ext;
} MSMConfig;

/* icicle/include/icicle/ntt.h:27-50 */
typedef enum { kForward = 0, kInverse = 1 } NTTDir;
typedef enum { kNN = 0, kNR, kRN, kRR, kNM, kMN } Ordering;

/* icicle/include/icicle/ntt.h:52-64 == wrappers/rust/icicle-core/src/ntt/mod.rs:73-91 */
typedef struct {
  icicleStreamHandle stream;
  bn254_scalar_t coset_gen;
  int batch_size;
  bool columns_batch;
  Ordering ordering;
  bool are_inputs_on_device;
  bool are_outputs_on_device;
  bool is_async;
  ConfigExtension* ext;
} NTTConfig;

/* icicle/include/icicle/ntt.h:92-96 */
typedef struct {
  icicleStreamHandle stream;
  bool is_async;
  ConfigExtension* ext;
} NTTInitDomainConfig;

/* icicle/include/icicle/vec_ops.h:18-36 == wrappers/rust/icicle-core/src/vec_ops/mod.rs:6-17 */
typedef struct {
  icicleStreamHandle stream;
  bool is_a_on_device;
  bool is_b_on_device;
  bool is_result_on_device;
  bool is_async;
  int batch_size;
  bool columns_batch;
  ConfigExtension* ext;
} VecOpsConfig;

/* ---- runtime: icicle/src/runtime.cpp ; wrappers/rust/icicle-runtime/src/runtime.rs:10-54 ---- */
eIcicleError icicle_load_backend(const char* path, bool is_recursive);          /* runtime.cpp:288 (no-op: backend is built in) */
eIcicleError icicle_load_backend_from_env_or_default(void);                     /* runtime.cpp:355 */
eIcicleError icicle_set_device(const IcicleDevice* device);                     /* runtime.cpp:15 */
eIcicleError icicle_set_default_device(const IcicleDevice* device);             /* runtime.cpp:17 */
eIcicleError icicle_get_active_device(IcicleDevice* device);                    /* runtime.cpp:22 */
eIcicleError icicle_is_host_memory(const void* ptr);                            /* runtime.cpp:29 */
eIcicleError icicle_is_active_device_memory(const void* ptr);                   /* runtime.cpp:35 */
eIcicleError icicle_get_device_count(int* device_count);                        /* runtime.cpp:43 */
eIcicleError icicle_is_device_available(const IcicleDevice* device);            /* runtime.cpp:259 */
eIcicleError icicle_get_registered_devices(char* output, size_t output_size);   /* runtime.cpp:264 */
eIcicleError icicle_malloc(void** ptr, size_t size);                            /* runtime.cpp:48 */
eIcicleError icicle_malloc_async(void** ptr, size_t size, icicleStreamHandle stream); /* runtime.cpp:57 */
eIcicleError icicle_free(void* ptr);                                            /* runtime.cpp:66 */
eIcicleError icicle_free_async(void* ptr, icicleStreamHandle stream);           /* runtime.cpp:95 */
eIcicleError icicle_get_available_memory(size_t* total, size_t* free);          /* runtime.cpp:120 */
eIcicleError icicle_memset(void* ptr, int value, size_t size);                  /* runtime.cpp:125 */
eIcicleError icicle_memset_async(void* ptr, int value, size_t size, icicleStreamHandle stream); /* runtime.cpp:134 */
eIcicleError icicle_copy(void* dst, const void* src, size_t size);              /* runtime.cpp:187 */
eIcicleError icicle_copy_async(void* dst, const void* src, size_t size, icicleStreamHandle stream); /* runtime.cpp:207 */
eIcicleError icicle_copy_to_host(void* dst, const void* src, size_t size);      /* runtime.cpp:224 */
eIcicleError icicle_copy_to_host_async(void* dst, const void* src, size_t size, icicleStreamHandle stream);   /* runtime.cpp:229 */
eIcicleError icicle_copy_to_device(void* dst, const void* src, size_t size);    /* runtime.cpp:234 */
eIcicleError icicle_copy_to_device_async(void* dst, const void* src, size_t size, icicleStreamHandle stream); /* runtime.cpp:239 */
eIcicleError icicle_create_stream(icicleStreamHandle* stream);                  /* runtime.cpp:269 */
eIcicleError icicle_destroy_stream(icicleStreamHandle stream);                  /* runtime.cpp:274 */
eIcicleError icicle_stream_synchronize(icicleStreamHandle stream);              /* runtime.cpp:244 */
eIcicleError icicle_device_synchronize(void);                                   /* runtime.cpp:249 */
/* icicle/include/icicle/device.h:53-58 == wrappers/rust/icicle-runtime/src/device.rs:14-20 */
typedef struct {
  bool using_host_memory;
  int num_memory_regions;
  bool supports_pinned_memory;
} IcicleDeviceProperties;
eIcicleError icicle_get_device_properties(IcicleDeviceProperties* properties);  /* runtime.cpp:254 */

/* ---- config extension: icicle/src/config_extension.cpp:7-37 ---- */
ConfigExtension* create_config_extension(void);
void destroy_config_extension(ConfigExtension* ext);
void config_extension_set_int(ConfigExtension* ext, const char* key, int value);
void config_extension_set_bool(ConfigExtension* ext, const char* key, bool value);
int config_extension_get_int(const ConfigExtension* ext, const char* key);
bool config_extension_get_bool(const ConfigExtension* ext, const char* key);
ConfigExtension* clone_config_extension(const ConfigExtension* ext);

/* ---- host-side scalar-field FFI: icicle/src/fields/ffi_extern.cpp:6-44 (synchronous, host) ---- */
void bn254_generate_scalars(bn254_scalar_t* scalars, int size);
void bn254_add(const bn254_scalar_t* a, const bn254_scalar_t* b, bn254_scalar_t* result);
void bn254_sub(const bn254_scalar_t* a, const bn254_scalar_t* b, bn254_scalar_t* result);
void bn254_mul(const bn254_scalar_t* a, const bn254_scalar_t* b, bn254_scalar_t* result);
void bn254_inv(const bn254_scalar_t* a, bn254_scalar_t* result);
void bn254_pow(const bn254_scalar_t* base, int exp, bn254_scalar_t* result);
void bn254_from_u32(uint32_t val, bn254_scalar_t* result);

/* ---- host-side curve FFI: icicle/src/curves/ffi_extern.cpp:9-133 ---- */
bool bn254_eq(const bn254_projective_t* a, const bn254_projective_t* b);
void bn254_ecadd(const bn254_projective_t* a, const bn254_projective_t* b, bn254_projective_t* result);
void bn254_ecsub(const bn254_projective_t* a, const bn254_projective_t* b, bn254_projective_t* result);
void bn254_mul_scalar(const bn254_projective_t* p, const bn254_scalar_t* s, bn254_projective_t* result);
void bn254_to_affine(const bn254_projective_t* p, bn254_affine_t* out);
void bn254_from_affine(const bn254_affine_t* p, bn254_projective_t* out);
void bn254_generator(bn254_projective_t* out);
bool bn254_is_on_curve(const bn254_projective_t* p);
void bn254_base_field_from_u32(uint32_t val, bn254_fq_t* result);
bool bn254_g2_eq(const bn254_g2_projective_t* a, const bn254_g2_projective_t* b);
void bn254_g2_ecadd(const bn254_g2_projective_t* a, const bn254_g2_projective_t* b, bn254_g2_projective_t* result);
void bn254_g2_ecsub(const bn254_g2_projective_t* a, const bn254_g2_projective_t* b, bn254_g2_projective_t* result);
void bn254_g2_mul_scalar(const bn254_g2_projective_t* p, const bn254_scalar_t* s, bn254_g2_projective_t* result);
void bn254_g2_to_affine(const bn254_g2_projective_t* p, bn254_g2_affine_t* out);
void bn254_g2_from_affine(const bn254_g2_affine_t* p, bn254_g2_projective_t* out);
void bn254_g2_generator(bn254_g2_projective_t* out);
bool bn254_g2_is_on_curve(const bn254_g2_projective_t* p);
void bn254_g2_base_field_from_u32(uint32_t val, bn254_fq2_t* result);

/* ---- pairing (host): icicle/src/pairing.cpp:11-26, models/bn.h; Rust: icicle-core/src/pairing/mod.rs:38-43 ----
 * The reference's C++ symbol returns void (pairing.cpp:22-26) while its Rust declaration reads an eIcicleError from it
 * (pairing/mod.rs:38-43); this one returns the code the Rust side expects (always ICICLE_SUCCESS for non-null arguments). */
eIcicleError bn254_pairing(const bn254_affine_t* p, const bn254_g2_affine_t* q, bn254_fq12_t* out);
/* TargetField host FFI: icicle/src/fields/ffi_extern_pairing_extension.cpp:6-52 (Rust: icicle-bn254/src/pairing/mod.rs:16) */
void bn254_pairing_target_field_add(const bn254_fq12_t* a, const bn254_fq12_t* b, bn254_fq12_t* result);
void bn254_pairing_target_field_sub(const bn254_fq12_t* a, const bn254_fq12_t* b, bn254_fq12_t* result);
void bn254_pairing_target_field_mul(const bn254_fq12_t* a, const bn254_fq12_t* b, bn254_fq12_t* result);
void bn254_pairing_target_field_inv(const bn254_fq12_t* a, bn254_fq12_t* result);
void bn254_pairing_target_field_pow(const bn254_fq12_t* base, int exp, bn254_fq12_t* result);
void bn254_pairing_target_field_from_u32(uint32_t val, bn254_fq12_t* result);
void bn254_pairing_target_field_generate_scalars(bn254_fq12_t* out, int size);

/* ---- device vector ops: icicle/src/vec_ops.cpp:52-97,165-171 (CUDA: cuda_vec_ops.cu, cuda_mont.cuh) ---- */
eIcicleError bn254_vector_add(const bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
eIcicleError bn254_vector_sub(const bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
eIcicleError bn254_vector_mul(const bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
/* remaining vec ops of the wrapper crate: icicle/src/vec_ops.cpp:9-34 (Σ, Π per batch vector), :55-65 (a += b),
 * :102-113 (a·b⁻¹, inverse(0) = 0), :118-161 (scalar[b] ∘ vector(b, ·)); batch layout per VecOpsConfig.columns_batch */
eIcicleError bn254_vector_div(const bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
eIcicleError bn254_vector_accumulate(bn254_scalar_t* a, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg);
eIcicleError bn254_vector_sum(const bn254_scalar_t* a, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
eIcicleError bn254_vector_product(const bn254_scalar_t* a, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
eIcicleError bn254_scalar_add_vec(const bn254_scalar_t* scalar, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
eIcicleError bn254_scalar_sub_vec(const bn254_scalar_t* scalar, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
eIcicleError bn254_scalar_mul_vec(const bn254_scalar_t* scalar, const bn254_scalar_t* b, uint64_t n, const VecOpsConfig* cfg, bn254_scalar_t* out);
eIcicleError bn254_scalar_convert_montgomery(const bn254_scalar_t* in, uint64_t n, bool is_to_montgomery, const VecOpsConfig* cfg, bn254_scalar_t* out);

/* ---- Montgomery conversion of points: icicle/src/curves/montgomery_conversion.cpp:13-33 ---- */
eIcicleError bn254_affine_convert_montgomery(const bn254_affine_t* in, uint64_t n, bool is_into, const VecOpsConfig* cfg, bn254_affine_t* out);
eIcicleError bn254_g2_affine_convert_montgomery(const bn254_g2_affine_t* in, size_t n, bool is_into, const VecOpsConfig* cfg, bn254_g2_affine_t* out);
/* icicle/src/curves/montgomery_conversion.cpp:47-74 */
eIcicleError bn254_projective_convert_montgomery(const bn254_projective_t* in, size_t n, bool is_into, const VecOpsConfig* cfg, bn254_projective_t* out);
eIcicleError bn254_g2_projective_convert_montgomery(const bn254_g2_projective_t* in, size_t n, bool is_into, const VecOpsConfig* cfg, bn254_g2_projective_t* out);

/* ---- NTT: icicle/src/ntt.cpp:10-63 (CUDA: ntt.cuh:441-758, mixed_radix_ntt.cu) ---- */
eIcicleError bn254_ntt(const bn254_scalar_t* input, int size, NTTDir dir, const NTTConfig* cfg, bn254_scalar_t* output);
eIcicleError bn254_ntt_init_domain(const bn254_scalar_t* primitive_root, const NTTInitDomainConfig* cfg);
eIcicleError bn254_ntt_release_domain(void);
eIcicleError bn254_get_root_of_unity(uint64_t max_size, bn254_scalar_t* rou);
eIcicleError bn254_get_root_of_unity_from_domain(uint64_t logn, bn254_scalar_t* rou);

/* ---- MSM: icicle/src/msm.cpp:12-32 (CUDA: cuda_msm.cuh:960-1443) ---- */
eIcicleError bn254_msm(const bn254_scalar_t* scalars, const bn254_affine_t* bases, int msm_size, const MSMConfig* cfg, bn254_projective_t* results);
eIcicleError bn254_g2_msm(const bn254_scalar_t* scalars, const bn254_g2_affine_t* bases, int msm_size, const MSMConfig* cfg, bn254_g2_projective_t* results);
/* icicle/src/msm.cpp:45-72: output[f·i + j] = 2^(j·shift)·P_i for f = cfg->precompute_factor; pass the result as `bases`
 * of an MSM with the same precompute_factor (and c).  Output residency: cfg->are_results_on_device. */
eIcicleError bn254_msm_precompute_bases(const bn254_affine_t* bases, int nof_bases, const MSMConfig* cfg, bn254_affine_t* output_bases);
eIcicleError bn254_g2_msm_precompute_bases(const bn254_g2_affine_t* bases, int nof_bases, const MSMConfig* cfg, bn254_g2_affine_t* output_bases);

/* ------------------------------------------------------------------------------------------------
 * Extensions (not in the reference; prefixed icicle_snark_).  Used by this repository's own prover
 * host, tests and bench; a reference-side caller never needs them.
 * ---------------------------------------------------------------------------------------------- */
/* last error text of the calling thread (the reference logs to stderr instead) */
const char* icicle_snark_last_error(void);
/* out[i] = s[i]·G (affine, standard form) for the G1 / G2 generator; device pointers.  Batch
 * fixed-base multiplication used by the zkey synthesiser. */
eIcicleError icicle_snark_g1_generator_mul(const bn254_scalar_t* s, uint64_t n, icicleStreamHandle stream, bn254_affine_t* out);
eIcicleError icicle_snark_g2_generator_mul(const bn254_scalar_t* s, uint64_t n, icicleStreamHandle stream, bn254_g2_affine_t* out);
/* per-phase device timings of the most recent MSM on this thread, milliseconds (HIP events):
 * [0] recode+sort, [1] bucket accumulation, [2] bucket reduction, [3] total.  Valid only when the
 * environment variable ICICLE_SNARK_PROFILE=1 is set (adds stream synchronisation). */
eIcicleError icicle_snark_last_msm_timings(float out_ms[4]);
/* HIP-event profile of the `back`-th most recent MSM of this process (0 = latest); the caller must have
 * synchronised that MSM's stream.  out_ms = {recode+sort, bucket-accumulation kernel, large buckets +
 * reduction + tail, total, the digit sort alone (0 if this MSM re-used another one's sort)};
 * geom = {L, nbuckets, c, W, is_g2}. */
eIcicleError icicle_snark_msm_profile(int back, float out_ms[5], uint32_t geom[5]);
/* PMC calibration: five kernels (probe_gather_kernel ×2, probe_stream_kernel, probe_store_kernel ×2) that move KNOWN byte counts
 * with the access patterns of this library — 64-byte and 128-byte random gathers, coalesced streaming reads, scattered and
 * coalesced 4-byte stores; out = those byte counts.  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (scratch/pmc_calibrate.sh). */
eIcicleError icicle_snark_pmc_probes(double out_bytes[5]);
/* measured machine constants for bench.py: out[0] = device-to-device copy GB/s (read + write bytes),
 * out[1] = v_mad_u64_u32 lane-operations per second / 10^12 */
eIcicleError icicle_snark_microbench(double out[2]);
/* the five access patterns of icicle_snark_pmc_probes TIMED (HIP events): out = GB/s of {64-byte gathers, 128-byte gathers, coalesced
 * reads, scattered 4-byte stores, coalesced 4-byte stores} over a 2 GiB buffer — what kind of box a bench line comes from */
eIcicleError icicle_snark_access_probes(double out_gbps[5]);

#ifdef __cplusplus
}
#endif
#endif /* ICICLE_SNARK_HIP_H */
