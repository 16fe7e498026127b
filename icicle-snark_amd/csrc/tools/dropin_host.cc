// dropin_host.cc — the reference's Rust host, call for call, over the C ABI of this library.
//
// The drop-in claim of INTEGRATION.md is that the UNCHANGED Rust host of the reference (src/lib.rs, src/cache.rs,
// src/proof_helper.rs, src/icicle_helper.rs, src/conversions.rs) links against these DSOs and works.  There is no Rust
// toolchain in this image, so this program issues exactly the FFI sequence that host issues — the same exports with the
// same arguments, flags, streams, interior-pointer slices and host-side loops — and nothing else from this library:
// only `icicle_*`, `bn254_*` and `*_config_extension` symbols of include/icicle_snark_hip.h are used (no groth16_* entry
// point, no fused kernels, no cached tables).  tests/test_dropin_sequence.py compares its proofs with the CPU oracle's;
// bench.py reports its time as config.prove_ms_dropin_sequence.
//
//   dropin_host <zkey> <wtns> <proof.json> <public.json> [--iters K] [--rs <r decimal> <s decimal>] [--keys-dir DIR]
//
// Mirrors (reference file:line → function here):
//   src/lib.rs:25-61                       main / prove_once
//   src/cache.rs:117-241, 264-289          compute_cache, pre_compute_keys (the CWD key file is kept: same name, same bytes)
//   src/cache.rs:242-256                   get_cache (domain from points_a.len())
//   src/proof_helper.rs:31-170             construct_r1cs
//   src/proof_helper.rs:172-241            groth16_commitments  + src/icicle_helper.rs:13-47 (ntt_helper, msm_helper)
//   src/proof_helper.rs:243-317            prove_helper (blinding through the host EC FFI, to_affine, JSON)
//   src/conversions.rs:13-56               from_affine_mont, serialize_g1/g2
//   wrappers/rust/icicle-core/src/{field.rs:379-398, curve.rs:339-351, vec_ops/mod.rs:166-190, msm/mod.rs:106-154,
//   ntt/mod.rs:202-216}                    the flag set-up each wrapper does before the FFI call
#include <chrono>
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "icicle_snark_hip.h"

typedef bn254_scalar_t F;
typedef bn254_affine_t G1A;
typedef bn254_g2_affine_t G2A;
typedef bn254_projective_t G1P;
typedef bn254_g2_projective_t G2P;

#define CHK(call)                                                                                    \
  do {                                                                                               \
    eIcicleError e__ = (call);                                                                       \
    if (e__ != ICICLE_SUCCESS) {                                                                     \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, (int)e__, icicle_snark_last_error());           \
      exit(2);                                                                                       \
    }                                                                                                \
  } while (0)

static bool g_trace = false;
static double g_t_last = 0;
static double now_ms();
static void lap(const char* what)
{
  if (!g_trace) return;
  const double t = now_ms();
  fprintf(stderr, "[dropin] %-40s %8.3f ms\n", what, t - g_t_last);
  g_t_last = t;
}
static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- DeviceVec / stream helpers as the Rust runtime crate issues them (icicle-runtime/src/memory.rs, stream.rs) -------
static void* device_malloc_async(size_t bytes, icicleStreamHandle s)
{
  void* p = nullptr;
  CHK(icicle_malloc_async(&p, bytes, s));
  return p;
}
static void check_active(const void* p) // every DeviceSlice argument is checked (memory.rs:120-125)
{
  if (icicle_is_active_device_memory(p) != ICICLE_SUCCESS) {
    fprintf(stderr, "pointer %p is not on the active device\n", p);
    exit(2);
  }
}
static icicleStreamHandle stream_create()
{
  icicleStreamHandle s = nullptr;
  CHK(icicle_create_stream(&s));
  return s;
}

// ---- snarkjs containers (src/file_wrapper.rs:45-103) ------------------------------------------------------------------
struct Section {
  const uint8_t* p = nullptr;
  uint64_t size = 0;
};
struct BinFile {
  const uint8_t* data = nullptr;
  size_t len = 0;
  std::vector<Section> sec;
  bool open(const char* path, const char* type)
  {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st)) return false;
    len = (size_t)st.st_size;
    data = (const uint8_t*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (data == MAP_FAILED || len < 12 || memcmp(data, type, 4)) return false;
    uint32_t nsec;
    memcpy(&nsec, data + 8, 4);
    sec.assign(16, Section());
    size_t pos = 12;
    for (uint32_t i = 0; i < nsec; i++) {
      if (len - pos < 12) return false;
      uint32_t t;
      uint64_t l;
      memcpy(&t, data + pos, 4);
      memcpy(&l, data + pos + 4, 8);
      pos += 12;
      if (l > len - pos) return false;
      if (t < sec.size()) sec[t] = {data + pos, l};
      pos += l;
    }
    return true;
  }
};

// ---- VecOpsConfig as the wrappers build it --------------------------------------------------------------------------
static VecOpsConfig vec_cfg_default()
{
  VecOpsConfig c;
  memset(&c, 0, sizeof c);
  c.batch_size = 1;
  c.ext = create_config_extension(); // VecOpsConfig::default() allocates one (vec_ops/mod.rs)
  return c;
}
static void vec_cfg_drop(VecOpsConfig& c)
{
  if (c.ext) destroy_config_extension(c.ext);
  c.ext = nullptr;
}

// ScalarField::from_mont(&mut DeviceSlice, &stream) — field.rs:379-398: is_a_on_device = true, is_result_on_device stays at
// its default (false) although the output is the same DEVICE buffer, is_async = !stream.is_null()
static void scalar_from_mont(F* d_values, size_t n, icicleStreamHandle s)
{
  check_active(d_values);
  VecOpsConfig c = vec_cfg_default();
  c.is_a_on_device = true;
  c.is_async = s != nullptr;
  c.stream = s;
  CHK(bn254_scalar_convert_montgomery(d_values, (uint64_t)n, false, &c, d_values));
  vec_cfg_drop(c);
}
// Affine::from_mont — curve.rs:148-153 → convert_affine_montgomery :339-351 (a and result on device, is_async = false)
static void g1_from_mont(G1A* d, size_t n, icicleStreamHandle s)
{
  check_active(d);
  VecOpsConfig c = vec_cfg_default();
  c.is_a_on_device = c.is_result_on_device = true;
  c.is_async = false;
  c.stream = s;
  CHK(bn254_affine_convert_montgomery(d, n, false, &c, d));
  vec_cfg_drop(c);
}
static void g2_from_mont(G2A* d, size_t n, icicleStreamHandle s)
{
  check_active(d);
  VecOpsConfig c = vec_cfg_default();
  c.is_a_on_device = c.is_result_on_device = true;
  c.is_async = false;
  c.stream = s;
  CHK(bn254_g2_affine_convert_montgomery(d, n, false, &c, d));
  vec_cfg_drop(c);
}
// mul_scalars / sub_scalars — vec_ops/mod.rs:233-245 → setup_config :166-190 (clone of the caller's config + residency)
static void vec_binop(bool mul, const F* a, bool a_dev, const F* b, bool b_dev, F* r, bool r_dev, size_t n, const VecOpsConfig& cfg)
{
  if (a_dev) check_active(a);
  if (b_dev) check_active(b);
  if (r_dev) check_active(r);
  VecOpsConfig c = cfg;
  c.ext = clone_config_extension(cfg.ext);
  c.batch_size = 1;
  c.is_a_on_device = a_dev;
  c.is_b_on_device = b_dev;
  c.is_result_on_device = r_dev;
  if (mul) CHK(bn254_vector_mul(a, b, (uint64_t)n, &c, r));
  else CHK(bn254_vector_sub(a, b, (uint64_t)n, &c, r));
  vec_cfg_drop(c);
}

// ---- the cache (src/cache.rs:58-72) ---------------------------------------------------------------------------------
struct ZKeyCache {
  size_t n_vars = 0, n_public = 0, domain_size = 0, power = 0;
  F r;
  G1P vk_alpha_1, vk_beta_1, vk_delta_1;
  G2P vk_beta_2, vk_gamma_2, vk_delta_2;
  std::vector<size_t> s_values, c_values, m_values;
  F* first_slice = nullptr; // device, n_coef
  F* keys = nullptr;        // device, domain_size
  G1A *points_a = nullptr, *points_b1 = nullptr, *points_c = nullptr, *points_h = nullptr;
  G2A* points_b = nullptr;
  size_t len_a = 0, len_b1 = 0, len_b = 0, len_c = 0, len_h = 0;
};

// src/conversions.rs:13-28 — device round trip for a handful of header points
template <class A>
static void from_affine_mont(A* pts, size_t n, bool g2)
{
  icicleStreamHandle s = stream_create();
  A* d = (A*)device_malloc_async(n * sizeof(A), s);
  CHK(icicle_copy_to_device_async(d, pts, n * sizeof(A), s));
  if (g2) g2_from_mont((G2A*)d, n, s);
  else g1_from_mont((G1A*)d, n, s);
  CHK(icicle_copy_to_host_async(pts, d, n * sizeof(A), s));
  CHK(icicle_stream_synchronize(s));
  CHK(icicle_destroy_stream(s));
  CHK(icicle_free(d)); // DeviceVec::drop
}
static G1P to_projective(const G1A& a) // curve.rs:78-88
{
  static const G1A zero = {};
  G1P p;
  memset(&p, 0, sizeof p);
  if (!memcmp(&a, &zero, sizeof a)) {
    p.y.limbs[0] = 1;
    return p;
  }
  p.x = a.x;
  p.y = a.y;
  p.z.limbs[0] = 1;
  return p;
}
static G2P to_projective(const G2A& a)
{
  static const G2A zero = {};
  G2P p;
  memset(&p, 0, sizeof p);
  if (!memcmp(&a, &zero, sizeof a)) {
    p.y.c0.limbs[0] = 1;
    return p;
  }
  p.x = a.x;
  p.y = a.y;
  p.z.c0.limbs[0] = 1;
  return p;
}

// W[power + 1] of src/cache.rs:25-56 equals bn254_get_root_of_unity(2^(power+1)) (checked in the survey)
static F coset_inc(size_t power)
{
  F w;
  CHK(bn254_get_root_of_unity(1ull << (power + 1), &w));
  return w;
}

// pre_compute_keys — src/cache.rs:264-289: serial host loop key ← key·inc through the field FFI, cached in a CWD file
static std::vector<F> pre_compute_keys(const F& inc, size_t size, const std::string& dir)
{
  char name[160];
  snprintf(name, sizeof name, "precomputed_%zu_0x", size);
  std::string path = dir + "/" + name;
  for (int i = 7; i >= 0; i--) {
    char h[16];
    snprintf(h, sizeof h, "%08x", inc.limbs[i]);
    path += h;
  }
  path += ".bin";
  std::vector<F> keys(size);
  FILE* f = fopen(path.c_str(), "rb");
  if (f) {
    const size_t got = fread(keys.data(), sizeof(F), size, f);
    fclose(f);
    if (got == size) return keys;
  }
  F key;
  bn254_from_u32(1, &key);
  for (size_t i = 0; i < size; i++) {
    keys[i] = key;
    bn254_mul(&key, &inc, &key);
  }
  f = fopen(path.c_str(), "wb");
  if (f) {
    fwrite(keys.data(), sizeof(F), size, f);
    fclose(f);
  }
  return keys;
}

// CacheManager::compute — src/cache.rs:117-241
static ZKeyCache compute_cache(const char* zkey_path, const std::string& keys_dir)
{
  ZKeyCache z;
  icicleStreamHandle stream = stream_create();
  BinFile f;
  if (!f.open(zkey_path, "zkey")) {
    fprintf(stderr, "cannot read zkey %s\n", zkey_path);
    exit(2);
  }
  // read_header_groth16 — src/zkey.rs:47-85
  const uint8_t* h = f.sec[2].p;
  uint32_t u;
  memcpy(&u, h + 72, 4); z.n_vars = u;
  memcpy(&u, h + 76, 4); z.n_public = u;
  memcpy(&u, h + 80, 4); z.domain_size = u;
  memcpy(&z.r, h + 40, 32);
  z.power = (size_t)log2f((float)z.domain_size);
  G1A g1s[3];
  G2A g2s[3];
  memcpy(&g1s[0], h + 84, 64);        // alpha1
  memcpy(&g1s[1], h + 84 + 64, 64);   // beta1
  memcpy(&g2s[0], h + 84 + 128, 128); // beta2
  memcpy(&g2s[1], h + 84 + 256, 128); // gamma2
  memcpy(&g1s[2], h + 84 + 384, 64);  // delta1
  memcpy(&g2s[2], h + 84 + 448, 128); // delta2
  from_affine_mont(g1s, 3, false);
  from_affine_mont(g2s, 3, true);
  z.vk_alpha_1 = to_projective(g1s[0]);
  z.vk_beta_1 = to_projective(g1s[1]);
  z.vk_delta_1 = to_projective(g1s[2]);
  z.vk_beta_2 = to_projective(g2s[0]);
  z.vk_gamma_2 = to_projective(g2s[1]);
  z.vk_delta_2 = to_projective(g2s[2]);

  const uint8_t* co = f.sec[4].p;
  const size_t s_coef = 4 * 3 + 32, n_coef = (f.sec[4].size - 4) / s_coef;
  std::vector<F> first_slice(n_coef);
  z.s_values.resize(n_coef);
  z.c_values.resize(n_coef);
  z.m_values.resize(n_coef);
  {
    const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
      th.emplace_back([&, t] {
        for (size_t i = t; i < n_coef; i += nt) {
          const uint8_t* b = co + 4 + i * s_coef;
          uint32_t s, c;
          memcpy(&c, b + 4, 4);
          memcpy(&s, b + 8, 4);
          z.s_values[i] = s;
          z.c_values[i] = c;
          z.m_values[i] = b[0];
          memcpy(&first_slice[i], b + 12, 32);
        }
      });
    for (auto& x : th) x.join();
  }
  const F inc = coset_inc(z.power);
  z.len_a = f.sec[5].size / 64; z.len_b1 = f.sec[6].size / 64; z.len_b = f.sec[7].size / 128; z.len_c = f.sec[8].size / 64; z.len_h = f.sec[9].size / 64;
  z.points_a = (G1A*)device_malloc_async(f.sec[5].size, stream);
  z.points_b1 = (G1A*)device_malloc_async(f.sec[6].size, stream);
  z.points_b = (G2A*)device_malloc_async(f.sec[7].size, stream);
  z.points_c = (G1A*)device_malloc_async(f.sec[8].size, stream);
  z.points_h = (G1A*)device_malloc_async(f.sec[9].size, stream);
  z.first_slice = (F*)device_malloc_async(n_coef * sizeof(F), stream);
  CHK(icicle_copy_to_device_async(z.points_a, f.sec[5].p, f.sec[5].size, stream));
  CHK(icicle_copy_to_device_async(z.points_b1, f.sec[6].p, f.sec[6].size, stream));
  CHK(icicle_copy_to_device_async(z.points_b, f.sec[7].p, f.sec[7].size, stream));
  CHK(icicle_copy_to_device_async(z.points_c, f.sec[8].p, f.sec[8].size, stream));
  CHK(icicle_copy_to_device_async(z.points_h, f.sec[9].p, f.sec[9].size, stream));
  CHK(icicle_copy_to_device_async(z.first_slice, first_slice.data(), n_coef * sizeof(F), stream));
  g1_from_mont(z.points_a, z.len_a, stream);
  g1_from_mont(z.points_b1, z.len_b1, stream);
  g2_from_mont(z.points_b, z.len_b, stream);
  g1_from_mont(z.points_c, z.len_c, stream);
  g1_from_mont(z.points_h, z.len_h, stream);
  scalar_from_mont(z.first_slice, n_coef, stream);
  CHK(icicle_stream_synchronize(stream));
  CHK(icicle_destroy_stream(stream));
  {
    // keys: allocated and copied on the (already destroyed) stream handle in the reference (cache.rs:208-213); a null
    // stream here
    std::vector<F> keys = pre_compute_keys(inc, z.domain_size, keys_dir);
    z.keys = (F*)device_malloc_async(z.domain_size * sizeof(F), nullptr);
    CHK(icicle_copy_to_device_async(z.keys, keys.data(), z.domain_size * sizeof(F), nullptr));
    CHK(icicle_stream_synchronize(nullptr));
  }
  return z;
}

// ntt_helper — src/icicle_helper.rs:13-32 + ntt_inplace (ntt/mod.rs:202-216)
static void ntt_helper(F* d_vec, size_t total, bool inverse, icicleStreamHandle s)
{
  check_active(d_vec);
  NTTConfig c;
  memset(&c, 0, sizeof c);
  c.coset_gen.limbs[0] = 1; // F::one()
  c.batch_size = 3;
  c.ordering = kNN;
  c.is_async = true;
  c.stream = s;
  c.are_inputs_on_device = c.are_outputs_on_device = true;
  c.ext = create_config_extension();
  CHK(bn254_ntt(d_vec, (int)(total / 3), inverse ? kInverse : kForward, &c, d_vec));
  destroy_config_extension(c.ext);
}

// construct_r1cs — src/proof_helper.rs:31-170
static F* construct_r1cs(const F* witness, const ZKeyCache& z, double* t_host_ms)
{
  icicleStreamHandle stream = stream_create();
  VecOpsConfig cfg = vec_cfg_default();
  cfg.is_async = true;
  cfg.stream = stream;
  const size_t n_coef = z.c_values.size(), nof_coef = z.domain_size;
  F* d_second_slice = (F*)device_malloc_async(n_coef * sizeof(F), stream);
  F* d_vec = (F*)device_malloc_async(nof_coef * 3 * sizeof(F), stream);
  // vec![ScalarField::zero(); nof_coef * 2] is a zeroed allocation (calloc); Vec::with_capacity + set_len (:49-52, :66-69)
  // leaves the memory uninitialised — std::vector<F>(n) would write 340 MB of zeros here
  struct HostBuf {
    F* p;
    explicit HostBuf(F* q) : p(q) {}
    ~HostBuf() { free(p); }
    F* data() const { return p; }
    F& operator[](size_t i) const { return p[i]; }
  };
  HostBuf out_buff_b_a((F*)calloc(nof_coef * 2, sizeof(F))), second_slice((F*)malloc(n_coef * sizeof(F))), res((F*)malloc(n_coef * sizeof(F)));
  if (!out_buff_b_a.p || !second_slice.p || !res.p) {
    fprintf(stderr, "out of host memory\n");
    exit(2);
  }
  lap("r1cs: malloc + host vectors");
  const double t0 = now_ms();
  {
    // second_slice.par_iter_mut(): rayon → plain threads
    const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
      th.emplace_back([&, t] {
        const size_t lo = n_coef * t / nt, hi = n_coef * (t + 1) / nt;
        for (size_t i = lo; i < hi; i++) second_slice[i] = witness[z.s_values[i]];
      });
    for (auto& x : th) x.join();
  }
  double host_ms = now_ms() - t0;
  lap("r1cs: host gather");
  CHK(icicle_copy_to_device_async(d_second_slice, second_slice.data(), n_coef * sizeof(F), stream));
  lap("r1cs: H2D second_slice");
  scalar_from_mont(d_second_slice, n_coef, stream);
  vec_binop(true, z.first_slice, true, d_second_slice, true, res.data(), false, n_coef, cfg); // result on the HOST
  CHK(icicle_stream_synchronize(stream));
  lap("r1cs: from_mont + mul -> host + sync");
  const double t1 = now_ms();
  {
    F zero;
    memset(&zero, 0, sizeof zero);
    for (size_t i = 0; i < n_coef; i++) { // the serial host scatter-add of :83-95
      F* value = &out_buff_b_a[z.c_values[i] + z.m_values[i] * nof_coef];
      if (!memcmp(value, &zero, sizeof zero)) *value = res[i];
      else if (memcmp(&res[i], &zero, sizeof zero)) bn254_add(value, &res[i], value);
    }
  }
  host_ms += now_ms() - t1;
  lap("r1cs: host scatter-add");
  CHK(icicle_copy_to_device_async(d_vec, out_buff_b_a.data() + nof_coef, nof_coef * sizeof(F), stream));
  CHK(icicle_copy_to_device_async(d_vec + nof_coef, out_buff_b_a.data(), nof_coef * sizeof(F), stream));
  lap("r1cs: H2D b, a");
  vec_binop(true, d_vec, true, d_vec + nof_coef, true, d_vec + 2 * nof_coef, true, nof_coef, cfg);
  ntt_helper(d_vec, 3 * nof_coef, true, stream);
  for (int k = 0; k < 3; k++) vec_binop(true, d_vec + k * nof_coef, true, z.keys, true, d_vec + k * nof_coef, true, nof_coef, cfg);
  ntt_helper(d_vec, 3 * nof_coef, false, stream);
  CHK(icicle_stream_synchronize(stream));
  lap("r1cs: mul, intt, 3 key muls, ntt + sync");
  CHK(icicle_destroy_stream(stream));
  // L·R − O with a fresh default config (synchronous, null stream) — :152-167
  VecOpsConfig c2 = vec_cfg_default();
  vec_binop(true, d_vec, true, d_vec + nof_coef, true, d_vec, true, nof_coef, c2);
  vec_binop(false, d_vec, true, d_vec + 2 * nof_coef, true, d_vec + nof_coef, true, nof_coef, c2);
  vec_cfg_drop(c2);
  vec_cfg_drop(cfg);
  lap("r1cs: final mul/sub (sync)");
  CHK(icicle_free(d_second_slice)); // DeviceVec::drop at the end of the function
  lap("r1cs: free");
  if (t_host_ms) *t_host_ms = host_ms;
  return d_vec;
}

// msm_helper — src/icicle_helper.rs:34-47 + msm() (msm/mod.rs:106-154)
template <class A, class P>
static P* msm_helper(const F* d_scalars, size_t n_scalars, const A* d_points, size_t n_points, icicleStreamHandle s, bool g2)
{
  P* d_result = (P*)device_malloc_async(sizeof(P), s);
  MSMConfig c;
  memset(&c, 0, sizeof c);
  c.precompute_factor = 1;
  c.batch_size = 1;
  c.are_points_shared_in_batch = true;
  c.ext = create_config_extension();
  c.stream = s;
  c.is_async = true;
  if (n_points == 0 || n_scalars % n_points) {
    fprintf(stderr, "Number of bases %zu does not divide the number of scalars %zu\n", n_points, n_scalars);
    exit(2);
  }
  check_active(d_scalars);
  check_active(d_points);
  check_active(d_result);
  MSMConfig l = c;
  l.ext = clone_config_extension(c.ext);
  l.are_points_shared_in_batch = n_points < n_scalars;
  l.batch_size = 1;
  l.are_scalars_on_device = l.are_points_on_device = l.are_results_on_device = true;
  if (g2) CHK(bn254_g2_msm(d_scalars, (const G2A*)d_points, (int)n_scalars, &l, (G2P*)d_result));
  else CHK(bn254_msm(d_scalars, (const G1A*)d_points, (int)n_scalars, &l, (G1P*)d_result));
  destroy_config_extension(l.ext);
  destroy_config_extension(c.ext);
  return d_result;
}

struct Commitments {
  G1P a, b1, c, h;
  G2P b;
};
// groth16_commitments — src/proof_helper.rs:172-241
static Commitments groth16_commitments(F* d_vec, const F* scalars, const ZKeyCache& z)
{
  const size_t nof_coef = z.domain_size;
  lap("prove: between construct_r1cs and commitments");
  icicleStreamHandle g1 = stream_create(), g2 = stream_create();
  lap("commitments: 2 streams created");
  F* d_scalars = (F*)device_malloc_async(z.n_vars * sizeof(F), g1);
  lap("commitments: malloc scalars");
  CHK(icicle_copy_to_device_async(d_scalars, scalars, z.n_vars * sizeof(F), g1));
  lap("commitments: H2D scalars");
  G1P* ca = msm_helper<G1A, G1P>(d_scalars, z.n_vars, z.points_a, z.len_a, g1, false);
  G1P* cb1 = msm_helper<G1A, G1P>(d_scalars, z.n_vars, z.points_b1, z.len_b1, g1, false);
  G1P* cc = msm_helper<G1A, G1P>(d_scalars + z.n_public + 1, z.n_vars - z.n_public - 1, z.points_c, z.len_c, g1, false);
  G1P* ch = msm_helper<G1A, G1P>(d_vec + nof_coef, nof_coef, z.points_h, z.len_h, g1, false);
  G2P* cb = msm_helper<G2A, G2P>(d_scalars, z.n_vars, z.points_b, z.len_b, g2, true);
  Commitments out;
  CHK(icicle_copy_to_host_async(&out.a, ca, sizeof(G1P), g1));
  CHK(icicle_copy_to_host_async(&out.b1, cb1, sizeof(G1P), g1));
  CHK(icicle_copy_to_host_async(&out.b, cb, sizeof(G2P), g2));
  CHK(icicle_copy_to_host_async(&out.c, cc, sizeof(G1P), g1));
  CHK(icicle_copy_to_host_async(&out.h, ch, sizeof(G1P), g1));
  lap("commitments: 5 msm enqueued");
  CHK(icicle_stream_synchronize(g1));
  CHK(icicle_stream_synchronize(g2));
  lap("commitments: sync");
  CHK(icicle_destroy_stream(g1));
  CHK(icicle_destroy_stream(g2));
  for (void* p : {(void*)ca, (void*)cb1, (void*)cc, (void*)ch, (void*)cb, (void*)d_scalars, (void*)d_vec}) CHK(icicle_free(p)); // drops
  return out;
}

// BigUint::to_str_radix(10)
static std::string to_decimal(const uint32_t limbs[8])
{
  uint32_t w[8];
  memcpy(w, limbs, 32);
  std::string out;
  bool nz = true;
  while (nz) {
    uint64_t rem = 0;
    nz = false;
    for (int i = 7; i >= 0; i--) {
      const uint64_t cur = (rem << 32) | w[i];
      w[i] = (uint32_t)(cur / 1000000000u);
      rem = cur % 1000000000u;
      if (w[i]) nz = true;
    }
    char buf[16];
    snprintf(buf, sizeof buf, nz ? "%09u" : "%u", (unsigned)rem);
    out.insert(0, buf);
  }
  return out;
}
static bool from_decimal(const char* s, F* out)
{
  uint32_t w[8] = {0};
  for (; *s; s++) {
    if (*s < '0' || *s > '9') return false;
    uint64_t carry = (uint64_t)(*s - '0');
    for (int i = 0; i < 8; i++) {
      const uint64_t cur = (uint64_t)w[i] * 10 + carry;
      w[i] = (uint32_t)cur;
      carry = cur >> 32;
    }
  }
  memcpy(out->limbs, w, 32);
  return true;
}

struct Timing {
  double total_ms, r1cs_ms, r1cs_host_ms, msm_ms;
};
// groth16_prove_helper — src/proof_helper.rs:243-317 + save_json_file
static void prove_once(const char* wtns_path, const ZKeyCache& z, const char* proof_path, const char* public_path, const F* rs_fixed, Timing* tm)
{
  const double t0 = now_ms();
  g_t_last = t0;
  BinFile w;
  if (!w.open(wtns_path, "wtns")) {
    fprintf(stderr, "cannot read wtns %s\n", wtns_path);
    exit(2);
  }
  uint32_t n_witness;
  memcpy(&n_witness, w.sec[1].p + 36, 4);
  if (memcmp(w.sec[1].p + 4, &z.r, 32)) {
    fprintf(stderr, "Curve of the witness does not match the curve of the proving key\n");
    exit(2);
  }
  if (n_witness != z.n_vars) {
    fprintf(stderr, "Invalid witness length. Circuit: %zu, witness: %u\n", z.n_vars, n_witness);
    exit(2);
  }
  const F* scalars = (const F*)w.sec[2].p;
  double host_ms = 0;
  F* d_vec = construct_r1cs(scalars, z, &host_ms);
  const double t1 = now_ms();
  Commitments cm = groth16_commitments(d_vec, scalars, z);
  const double t2 = now_ms();
  F rs[2];
  if (rs_fixed) memcpy(rs, rs_fixed, sizeof rs);
  else bn254_generate_scalars(rs, 2); // ScalarCfg::generate_random(2)
  const F &r = rs[0], &s = rs[1];
  G1P pi_a, pi_b1, pi_c, t;
  G2P pi_b, t2p;
  // pi_a = pi_a + vk_alpha_1 + vk_delta_1 * r
  bn254_ecadd(&cm.a, &z.vk_alpha_1, &pi_a);
  bn254_mul_scalar(&z.vk_delta_1, &r, &t);
  bn254_ecadd(&pi_a, &t, &pi_a);
  // pi_b = pi_b + vk_beta_2 + vk_delta_2 * s
  bn254_g2_ecadd(&cm.b, &z.vk_beta_2, &pi_b);
  bn254_g2_mul_scalar(&z.vk_delta_2, &s, &t2p);
  bn254_g2_ecadd(&pi_b, &t2p, &pi_b);
  // pi_b1 = pi_b1 + vk_beta_1 + vk_delta_1 * s
  bn254_ecadd(&cm.b1, &z.vk_beta_1, &pi_b1);
  bn254_mul_scalar(&z.vk_delta_1, &s, &t);
  bn254_ecadd(&pi_b1, &t, &pi_b1);
  // pi_c = pi_c + pi_h + pi_a * s + pi_b1 * r - vk_delta_1 * r * s      (left to right)
  bn254_ecadd(&cm.c, &cm.h, &pi_c);
  bn254_mul_scalar(&pi_a, &s, &t);
  bn254_ecadd(&pi_c, &t, &pi_c);
  bn254_mul_scalar(&pi_b1, &r, &t);
  bn254_ecadd(&pi_c, &t, &pi_c);
  bn254_mul_scalar(&z.vk_delta_1, &r, &t);
  bn254_mul_scalar(&t, &s, &t);
  bn254_ecsub(&pi_c, &t, &pi_c);
  G1A a_aff, c_aff;
  G2A b_aff;
  bn254_to_affine(&pi_a, &a_aff);
  bn254_g2_to_affine(&pi_b, &b_aff);
  bn254_to_affine(&pi_c, &c_aff);
  // serde_json::to_writer_pretty of json!(proof): keys sorted, 2-space indent
  std::string pj = "{\n  \"curve\": \"bn128\",\n";
  pj += "  \"pi_a\": [\n    \"" + to_decimal(a_aff.x.limbs) + "\",\n    \"" + to_decimal(a_aff.y.limbs) + "\",\n    \"1\"\n  ],\n";
  pj += "  \"pi_b\": [\n    [\n      \"" + to_decimal(b_aff.x.c0.limbs) + "\",\n      \"" + to_decimal(b_aff.x.c1.limbs) + "\"\n    ],\n    [\n      \"" + to_decimal(b_aff.y.c0.limbs) +
        "\",\n      \"" + to_decimal(b_aff.y.c1.limbs) + "\"\n    ],\n    [\n      \"1\",\n      \"0\"\n    ]\n  ],\n";
  pj += "  \"pi_c\": [\n    \"" + to_decimal(c_aff.x.limbs) + "\",\n    \"" + to_decimal(c_aff.y.limbs) + "\",\n    \"1\"\n  ],\n";
  pj += "  \"protocol\": \"groth16\"\n}";
  std::string qj = z.n_public ? "[\n" : "[]";
  for (size_t i = 1; i <= z.n_public; i++) {
    qj += "  \"" + to_decimal(scalars[i].limbs) + "\"";
    qj += i == z.n_public ? "\n]" : ",\n";
  }
  for (int k = 0; k < 2; k++) {
    FILE* f = fopen(k ? public_path : proof_path, "wb");
    if (!f) {
      fprintf(stderr, "cannot write %s\n", k ? public_path : proof_path);
      exit(2);
    }
    fputs(k ? qj.c_str() : pj.c_str(), f);
    fclose(f);
  }
  munmap((void*)w.data, w.len);
  if (tm) *tm = {now_ms() - t0, t1 - t0, host_ms, t2 - t1};
}

int main(int argc, char** argv)
{
  if (argc < 5) {
    fprintf(stderr, "usage: %s <zkey> <wtns> <proof.json> <public.json> [--iters K] [--rs <r> <s>] [--keys-dir DIR]\n", argv[0]);
    return 1;
  }
  int iters = 1;
  F rs[2];
  bool have_rs = false;
  std::string keys_dir = ".";
  for (int i = 5; i < argc; i++) {
    if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rs") && i + 2 < argc) {
      have_rs = from_decimal(argv[i + 1], &rs[0]) && from_decimal(argv[i + 2], &rs[1]);
      i += 2;
    } else if (!strcmp(argv[i], "--keys-dir") && i + 1 < argc) keys_dir = argv[++i];
  }
  g_trace = getenv("DROPIN_TRACE") != nullptr;
  // try_load_and_set_backend_device — src/lib.rs:25-31
  CHK(icicle_load_backend_from_env_or_default());
  IcicleDevice dev;
  memset(&dev, 0, sizeof dev);
  strcpy(dev.type, "HIP");
  dev.id = 0;
  CHK(icicle_set_device(&dev));
  const double tc = now_ms();
  ZKeyCache z = compute_cache(argv[1], keys_dir);
  // get_cache — src/cache.rs:242-256: domain from points_a.len()
  F root;
  CHK(bn254_get_root_of_unity((uint64_t)z.len_a, &root));
  NTTInitDomainConfig ic;
  memset(&ic, 0, sizeof ic);
  ic.ext = create_config_extension();
  CHK(bn254_ntt_init_domain(&root, &ic));
  destroy_config_extension(ic.ext);
  printf("cache took: %.3fms\n", now_ms() - tc);
  for (int it = 0; it < iters; it++) {
    Timing tm;
    prove_once(argv[2], z, argv[3], argv[4], have_rs ? rs : nullptr, &tm);
    printf("proof took: %.3fms (construct_r1cs %.3f of which host gather/scatter %.3f, commitments %.3f)\n", tm.total_ms, tm.r1cs_ms, tm.r1cs_host_ms, tm.msm_ms);
    fflush(stdout);
  }
  return 0;
}
