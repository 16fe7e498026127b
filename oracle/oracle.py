"""ctypes front end of the CPU oracle (oracle/bn254_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py — never by the product path (icicle-snark_amd/).

Besides thin wrappers over the C functions this module restates, in plain Python,
the host-side orchestration of the reference prover so that an oracle proof can be
produced from (zkey bytes, wtns bytes, r, s):

* snarkjs container parsing      — src/file_wrapper.rs:45-103, src/zkey.rs:47-85
* ZKeyCache construction         — src/cache.rs:117-241
* groth16_prove_helper           — src/proof_helper.rs:243-317
* proof / public JSON values     — src/conversions.rs:30-56

Data conventions: a field element is 32 bytes little-endian (8×u32 limbs, the
reference layout, icicle-core field.rs:10-15); numpy arrays of elements have dtype
uint64 and shape (..., 4).
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbn254_oracle.so")

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, a few seconds). Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("bn254_oracle.c", "ec_tmpl.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libbn254_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = C.CDLL(_LIB_PATH)
        except OSError:
            build(force=True)
            _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_init()
    return _lib


# --------------------------------------------------------------------------- helpers
def to_le(x: int) -> bytes:
    return int(x).to_bytes(32, "little")


def from_le(b: bytes) -> int:
    return int.from_bytes(bytes(b), "little")


def ints_to_arr(xs) -> np.ndarray:
    """list of ints → (n,4) uint64"""
    buf = b"".join(to_le(x) for x in xs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def arr_to_ints(a: np.ndarray):
    raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1, 32)
    return [int.from_bytes(r.tobytes(), "little") for r in raw]


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _new(*shape):
    return np.zeros(shape, dtype=np.uint64)


# --------------------------------------------------------------------------- field ops
def _bin(name, a, b):
    out = _new(4)
    getattr(lib(), name)(_p(ints_to_arr([a])), _p(ints_to_arr([b])), _p(out))
    return arr_to_ints(out)[0]


def fr_mul(a, b): return _bin("oracle_fr_mul", a, b)
def fr_add(a, b): return _bin("oracle_fr_add", a, b)
def fr_sub(a, b): return _bin("oracle_fr_sub", a, b)
def fq_mul(a, b): return _bin("oracle_fq_mul", a, b)
def fq_add(a, b): return _bin("oracle_fq_add", a, b)
def fq_sub(a, b): return _bin("oracle_fq_sub", a, b)


def fr_inv(a):
    out = _new(4)
    lib().oracle_fr_inv(_p(ints_to_arr([a])), _p(out))
    return arr_to_ints(out)[0]


def fq_inv(a):
    out = _new(4)
    lib().oracle_fq_inv(_p(ints_to_arr([a])), _p(out))
    return arr_to_ints(out)[0]


def fr_omega(logn: int) -> int:
    out = _new(4)
    rc = lib().oracle_fr_omega(C.c_uint32(logn), _p(out))
    if rc:
        raise ValueError("no root of unity of that order")
    return arr_to_ints(out)[0]


def get_root_of_unity(max_size: int) -> int:
    """bn254_get_root_of_unity — icicle/src/ntt.cpp:52-61 (ceil(log2(max_size)))"""
    logn = max(0, (int(max_size) - 1).bit_length())
    return fr_omega(logn)


def fr_convert_montgomery(a: np.ndarray, to_mont: bool) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    lib().oracle_fr_convert_montgomery(_p(a), C.c_uint64(a.shape[0]), C.c_int(int(to_mont)), _p(out))
    return out


def fq_convert_montgomery(a: np.ndarray, to_mont: bool) -> np.ndarray:
    """a: (..., 4) uint64 array of Fq coordinates (any leading shape)."""
    shape = a.shape
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    lib().oracle_fq_convert_montgomery(_p(a), C.c_uint64(a.shape[0]), C.c_int(int(to_mont)), _p(out))
    return out.reshape(shape)


def _vec(name, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    assert a.shape == b.shape
    out = np.empty_like(a)
    getattr(lib(), name)(_p(a), _p(b), C.c_uint64(a.shape[0]), _p(out))
    return out


def fr_vector_mul(a, b): return _vec("oracle_fr_vector_mul", a, b)
def fr_vector_sub(a, b): return _vec("oracle_fr_vector_sub", a, b)
def fr_vector_add(a, b): return _vec("oracle_fr_vector_add", a, b)


def fr_ntt(a: np.ndarray, inverse: bool, batch: int = 1, domain_log: int | None = None) -> np.ndarray:
    """batch contiguous rows, natural order in/out (NTTConfig defaults + batch_size)."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    n = a.shape[0] // batch
    logn = n.bit_length() - 1
    if domain_log is None:
        domain_log = logn
    root = ints_to_arr([fr_omega(domain_log)])
    out = np.empty_like(a)
    rc = lib().oracle_fr_ntt(_p(a), C.c_int(n), C.c_int(int(inverse)), C.c_int(batch), _p(root), C.c_int(domain_log), _p(out))
    if rc:
        raise ValueError(f"oracle_fr_ntt rc={rc}")
    return out


def fr_dft_naive(a: np.ndarray, w: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    lib().oracle_fr_dft_naive(_p(a), C.c_int(a.shape[0]), _p(ints_to_arr([w])), _p(out))
    return out


# --------------------------------------------------------------------------- curve ops
# G1: affine (2,4) u64 / projective (3,4);  G2: affine (4,4) [x.c0,x.c1,y.c0,y.c1] / projective (6,4)
_DIMS = {"g1": (2, 3), "g2": (4, 6)}


def _g(group, name):
    return getattr(lib(), f"oracle_{group}_{name}")


def ec_add(group, a, b):
    out = _new(_DIMS[group][1], 4)
    _g(group, "add")(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
    return out


def ec_sub(group, a, b):
    out = _new(_DIMS[group][1], 4)
    _g(group, "sub")(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
    return out


def ec_add_mixed(group, a, b_aff):
    out = _new(_DIMS[group][1], 4)
    _g(group, "add_mixed")(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b_aff)), _p(out))
    return out


def ec_dbl(group, a):
    out = _new(_DIMS[group][1], 4)
    _g(group, "dbl")(_p(np.ascontiguousarray(a)), _p(out))
    return out


def ec_mul_scalar(group, a, s: int):
    out = _new(_DIMS[group][1], 4)
    _g(group, "mul_scalar")(_p(np.ascontiguousarray(a)), _p(ints_to_arr([s])), _p(out))
    return out


def ec_to_affine(group, a):
    out = _new(_DIMS[group][0], 4)
    _g(group, "to_affine")(_p(np.ascontiguousarray(a)), _p(out))
    return out


def ec_from_affine(group, a):
    out = _new(_DIMS[group][1], 4)
    _g(group, "from_affine")(_p(np.ascontiguousarray(a)), _p(out))
    return out


def ec_eq(group, a, b) -> bool:
    return bool(_g(group, "eq")(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b))))


def ec_is_on_curve(group, a) -> bool:
    return bool(_g(group, "is_on_curve")(_p(np.ascontiguousarray(a))))


def ec_generator(group):
    out = _new(_DIMS[group][1], 4)
    _g(group, "generator")(_p(out))
    return out


def ec_zero(group):
    out = _new(_DIMS[group][1], 4)
    out[_DIMS[group][1] // 3, 0] = 1  # y = 1  (G2: y.c0 = 1)
    return out


def msm(group, scalars: np.ndarray, bases: np.ndarray, c: int = 0, naive: bool = False):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(scalars.shape[0], _DIMS[group][0], 4)
    out = _new(_DIMS[group][1], 4)
    if naive:
        _g(group, "msm_naive")(_p(scalars), _p(bases), C.c_int(scalars.shape[0]), _p(out))
    else:
        _g(group, "msm")(_p(scalars), _p(bases), C.c_int(scalars.shape[0]), C.c_int(c), _p(out))
    return out


def fixed_base_mul(group, base_aff: np.ndarray, scalars: np.ndarray) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = _new(scalars.shape[0], _DIMS[group][0], 4)
    _g(group, "fixed_base_mul")(_p(np.ascontiguousarray(base_aff)), _p(scalars), C.c_int64(scalars.shape[0]), _p(out))
    return out


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def set_num_threads(n: int):
    lib().oracle_set_num_threads(C.c_int(int(n)))


def cpu_quota() -> int | None:
    """CPUs this process may actually run on at once: the cgroup CPU quota (cpu.max = "<quota> <period>", v1: cfs_quota_us /
    cfs_period_us) when one is set — on the MI355X boxes the container sees 256 logical CPUs and a quota of 16."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, -(-int(q) // int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, -(-q // p))
    except (OSError, ValueError):
        pass
    return None


def calibrate_threads(candidates=(8, 16, 32, 64, 128, 256)) -> int:
    """OpenMP thread count for the CPU baseline.  With a cgroup CPU quota that is the answer (more threads are only
    throttled: thread sweep on the MI355X box, quota 16 of 256 logical CPUs — G1 MSM of 2^20: 0.38 s on 16 threads, 0.51 s on 32,
    0.86 s on 64, 1.34 s on 256; profiles/r02_cpu_threads.txt).  Without one: the candidate that runs a G1 MSM of 2^17 fastest."""
    import time
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota()
    if q is not None:
        best = max(1, min(q, ncpu))
        set_num_threads(best)
        return best
    rng = np.random.default_rng(1)
    n = 1 << 17
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    set_num_threads(min(ncpu, 64))
    pts = fixed_base_mul("g1", ec_to_affine("g1", ec_generator("g1")), sc[::-1].copy())
    best, best_t = None, None
    for t in candidates:
        if t > ncpu:
            break
        set_num_threads(t)
        msm("g1", sc, pts)
        t0 = time.perf_counter()
        msm("g1", sc, pts)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = t, dt
    set_num_threads(best or min(ncpu, 8))
    return best or min(ncpu, 8)


# --------------------------------------------------------------------------- snarkjs containers
def read_sections(data: bytes, expected_type: bytes, max_version: int = 2):
    """FileWrapper::read_bin_file — src/file_wrapper.rs:45-103. Returns {id: (offset, size)}."""
    if data[:4] != expected_type:
        raise ValueError("Invalid File format")
    version, nsec = struct.unpack_from("<II", data, 4)
    if version > max_version:
        raise ValueError("Version not supported")
    pos = 12
    sections = {}
    for _ in range(nsec):
        ht, hl = struct.unpack_from("<IQ", data, pos)
        pos += 12
        sections.setdefault(ht, []).append((pos, hl))
        pos += hl
    return sections


def _section(data, sections, sid):
    (off, size), = sections[sid]
    return memoryview(data)[off:off + size]


def parse_wtns(data: bytes):
    """read_wtns_header — src/file_wrapper.rs:169-177 ; section 2 = n_witness × 32 B standard form."""
    sec = read_sections(data, b"wtns")
    h = _section(data, sec, 1)
    n8 = struct.unpack_from("<I", h, 0)[0]
    q = from_le(h[4:4 + n8])
    n_witness = struct.unpack_from("<I", h, 4 + n8)[0]
    w = np.frombuffer(_section(data, sec, 2), dtype=np.uint64).reshape(-1, 4)
    return {"n8": n8, "q": q, "n_witness": n_witness, "witness": w}


def parse_zkey(data: bytes):
    """read_zkey_header + the section views of CacheManager::compute —
    src/zkey.rs:47-85, src/cache.rs:126-181.  Everything is returned exactly as stored
    (Montgomery form); conversions happen in build_cache()."""
    sec = read_sections(data, b"zkey")
    if struct.unpack_from("<I", _section(data, sec, 1), 0)[0] != 1:
        raise ValueError("Protocol not supported")
    h = _section(data, sec, 2)
    pos = 0
    n8q = struct.unpack_from("<I", h, pos)[0]; pos += 4
    q = from_le(h[pos:pos + n8q]); pos += n8q
    n8r = struct.unpack_from("<I", h, pos)[0]; pos += 4
    r = from_le(h[pos:pos + n8r]); pos += n8r
    n_vars, n_public, domain_size = struct.unpack_from("<III", h, pos); pos += 12

    def g1():
        nonlocal pos
        a = np.frombuffer(h[pos:pos + 64], dtype=np.uint64).reshape(2, 4).copy(); pos += 64
        return a

    def g2():
        nonlocal pos
        a = np.frombuffer(h[pos:pos + 128], dtype=np.uint64).reshape(4, 4).copy(); pos += 128
        return a

    z = dict(n8q=n8q, q=q, n8r=n8r, r=r, n_vars=n_vars, n_public=n_public, domain_size=domain_size)
    z["vk_alpha_1"] = g1(); z["vk_beta_1"] = g1(); z["vk_beta_2"] = g2()
    z["vk_gamma_2"] = g2(); z["vk_delta_1"] = g1(); z["vk_delta_2"] = g2()
    coeffs = _section(data, sec, 4)
    n_coef = (len(coeffs) - 4) // (12 + n8r)
    rec = np.frombuffer(coeffs[4:4 + n_coef * 44], dtype=np.uint8).reshape(n_coef, 44)
    z["m"] = rec[:, 0].astype(np.uint32)  # only byte 0 is read — src/cache.rs:159
    z["c"] = rec[:, 4:8].copy().view(np.uint32).reshape(-1)
    z["s"] = rec[:, 8:12].copy().view(np.uint32).reshape(-1)
    z["coef"] = rec[:, 12:44].copy().view(np.uint64).reshape(-1, 4)
    z["A"] = np.frombuffer(_section(data, sec, 5), dtype=np.uint64).reshape(-1, 2, 4)
    z["B1"] = np.frombuffer(_section(data, sec, 6), dtype=np.uint64).reshape(-1, 2, 4)
    z["B2"] = np.frombuffer(_section(data, sec, 7), dtype=np.uint64).reshape(-1, 4, 4)
    z["C"] = np.frombuffer(_section(data, sec, 8), dtype=np.uint64).reshape(-1, 2, 4)
    z["H"] = np.frombuffer(_section(data, sec, 9), dtype=np.uint64).reshape(-1, 2, 4)
    return z


def build_cache(z: dict) -> dict:
    """CacheManager::compute — src/cache.rs:117-241: from_mont on every point array and on the
    coefficient values (once: file holds value·R², the cache value·R), coset keys inc^i."""
    cache = dict(z)
    for k in ("A", "B1", "B2", "C", "H", "vk_alpha_1", "vk_beta_1", "vk_beta_2", "vk_gamma_2", "vk_delta_1", "vk_delta_2"):
        cache[k] = fq_convert_montgomery(z[k], to_mont=False)
    cache["coef_R"] = fr_convert_montgomery(z["coef"], to_mont=False)
    n = z["domain_size"]
    power = int(np.log2(np.float32(n)))  # zkey.rs:62
    inc = fr_omega(power + 1)             # cache.rs:183-184, W[power+1]
    keys, k = [], 1
    for _ in range(n):                    # cache.rs:281-284
        keys.append(k)
        k = k * inc % R_MOD
    cache["inc"] = inc
    cache["keys"] = ints_to_arr(keys)
    # get_cache: domain sized from points_a.len() (= n_vars) — src/cache.rs:249
    cache["domain_log"] = max(0, (int(z["n_vars"]) - 1).bit_length())
    return cache


def construct_r1cs(witness: np.ndarray, cache: dict) -> np.ndarray:
    """construct_r1cs — src/proof_helper.rs:31-170 (C restatement oracle_construct_r1cs)."""
    n = cache["domain_size"]
    d_vec = _new(3 * n, 4)
    root = ints_to_arr([fr_omega(cache["domain_log"])])
    w = np.ascontiguousarray(witness, dtype=np.uint64)
    rc = lib().oracle_construct_r1cs(
        _p(w), _p(np.ascontiguousarray(cache["coef_R"])), _p(np.ascontiguousarray(cache["s"])),
        _p(np.ascontiguousarray(cache["c"])), _p(np.ascontiguousarray(cache["m"])),
        C.c_uint64(len(cache["s"])), C.c_uint64(n), _p(cache["keys"]), _p(root), C.c_int(cache["domain_log"]), _p(d_vec))
    if rc:
        raise ValueError(f"construct_r1cs rc={rc}")
    return d_vec


def _aff_to_strs_g1(a):
    x, y = arr_to_ints(a)
    return [str(x), str(y), "1"]


def _aff_to_strs_g2(a):
    x0, x1, y0, y1 = arr_to_ints(a)
    return [[str(x0), str(x1)], [str(y0), str(y1)], ["1", "0"]]


def groth16_commitments_of(cache: dict, w: np.ndarray, timings: dict | None = None) -> dict:
    """construct_r1cs + groth16_commitments — src/proof_helper.rs:31-241: the five MSM results (projective, this
    restatement's representatives) for one witness.  They do not depend on (r, s): a test that proves one witness under
    several blinding pairs computes them once and calls groth16_assemble per pair."""
    import time
    t0 = time.perf_counter()
    d_vec = construct_r1cs(w, cache)
    t1 = time.perf_counter()
    n, npub = cache["domain_size"], cache["n_public"]
    cm = dict(a=msm("g1", w, cache["A"]), b1=msm("g1", w, cache["B1"]), b=msm("g2", w, cache["B2"]),
              c=msm("g1", w[npub + 1:], cache["C"]), h=msm("g1", d_vec[n:2 * n], cache["H"]))
    if timings is not None:
        timings.update(qap_s=t1 - t0, msm_s=time.perf_counter() - t1)
    return cm


def groth16_assemble(cache: dict, w: np.ndarray, cm: dict, r: int = 1, s: int = 1):
    """blinding, to_affine, JSON values — src/proof_helper.rs:274-316"""
    npub = cache["n_public"]
    P = lambda g, k: ec_from_affine(g, cache[k])
    pi_a = ec_add("g1", ec_add("g1", cm["a"], P("g1", "vk_alpha_1")), ec_mul_scalar("g1", P("g1", "vk_delta_1"), r))
    pi_b = ec_add("g2", ec_add("g2", cm["b"], P("g2", "vk_beta_2")), ec_mul_scalar("g2", P("g2", "vk_delta_2"), s))
    pi_b1 = ec_add("g1", ec_add("g1", cm["b1"], P("g1", "vk_beta_1")), ec_mul_scalar("g1", P("g1", "vk_delta_1"), s))
    acc = ec_add("g1", cm["c"], cm["h"])
    acc = ec_add("g1", acc, ec_mul_scalar("g1", pi_a, s))
    acc = ec_add("g1", acc, ec_mul_scalar("g1", pi_b1, r))
    rs = ec_mul_scalar("g1", ec_mul_scalar("g1", P("g1", "vk_delta_1"), r), s)
    pi_c = ec_sub("g1", acc, rs)
    proof = {
        "pi_a": _aff_to_strs_g1(ec_to_affine("g1", pi_a)),
        "pi_b": _aff_to_strs_g2(ec_to_affine("g2", pi_b)),
        "pi_c": _aff_to_strs_g1(ec_to_affine("g1", pi_c)),
        "protocol": "groth16",
        "curve": "bn128",
    }
    public = [str(x) for x in arr_to_ints(w[1:npub + 1])]
    return proof, public


def groth16_prove(zkey_bytes: bytes, wtns_bytes: bytes, r: int = 1, s: int = 1, cache: dict | None = None, timings: dict | None = None):
    """groth16_prove_helper — src/proof_helper.rs:243-317.  (r, s) = (1, 1) reproduces the
    `no-randomness` feature (:287-295, algebraically identical to the general formula)."""
    import time
    if cache is None:
        cache = build_cache(parse_zkey(zkey_bytes))
    wt = parse_wtns(wtns_bytes)
    if wt["q"] != cache["r"]:
        raise ValueError("Curve of the witness does not match the curve of the proving key")
    if wt["n_witness"] != cache["n_vars"]:
        raise ValueError("Invalid witness length")
    w = wt["witness"]
    t0 = time.perf_counter()
    cm = groth16_commitments_of(cache, w, timings)
    out = groth16_assemble(cache, w, cm, r, s)
    if timings is not None:
        timings["total_s"] = time.perf_counter() - t0
    return out
