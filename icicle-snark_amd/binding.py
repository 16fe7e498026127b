"""ctypes mirror of include/icicle_snark_hip.h (the reference's extern "C" surface for the Groth16 path).

Names follow the reference's Rust wrappers so that tests read like theirs:
``msm`` (icicle-core/src/msm/mod.rs:106-154), ``ntt`` / ``initialize_domain`` / ``release_domain`` /
``get_root_of_unity`` (ntt/mod.rs:202-216,290-305), ``mul_scalars`` / ``sub_scalars`` / ``add_scalars``
(vec_ops/mod.rs:233-245), ``from_mont`` / ``to_mont`` (field.rs:379-398, curve.rs:140-154), ``DeviceVec``
(icicle-runtime/src/memory.rs), ``IcicleStream`` (stream.rs).

There is NO fallback of any kind here: if the shared library is missing or no HIP device is present the
calls raise.  Arrays are numpy uint64 with a trailing dimension of 4 (32-byte little-endian elements).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libicicle_snark_hip.so")
# The prover overlaps six streams; the HIP runtime's default of four hardware queues makes two of them serialise
# (csrc/runtime.cpp does the same when the library is loaded; set here too so that it also precedes any other DSO
# of this process that initialises the HIP runtime first, e.g. RCCL).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

SUCCESS = 0
ERRORS = ["SUCCESS", "INVALID_DEVICE", "OUT_OF_MEMORY", "INVALID_POINTER", "ALLOCATION_FAILED", "DEALLOCATION_FAILED",
          "COPY_FAILED", "SYNCHRONIZATION_FAILED", "STREAM_CREATION_FAILED", "STREAM_DESTRUCTION_FAILED",
          "API_NOT_IMPLEMENTED", "INVALID_ARGUMENT", "BACKEND_LOAD_FAILED", "LICENSE_CHECK_ERROR", "UNKNOWN_ERROR"]


class IcicleError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        name = ERRORS[code] if 0 <= code < len(ERRORS) else str(code)
        super().__init__(f"eIcicleError::{name} {what}".strip())


class Device(C.Structure):
    _fields_ = [("type", C.c_char * 64), ("id", C.c_int)]


class MSMConfig(C.Structure):
    _fields_ = [("stream", C.c_void_p), ("precompute_factor", C.c_int), ("c", C.c_int), ("bitsize", C.c_int),
                ("batch_size", C.c_int), ("are_points_shared_in_batch", C.c_bool), ("are_scalars_on_device", C.c_bool),
                ("are_scalars_montgomery_form", C.c_bool), ("are_points_on_device", C.c_bool),
                ("are_points_montgomery_form", C.c_bool), ("are_results_on_device", C.c_bool), ("is_async", C.c_bool),
                ("ext", C.c_void_p)]

    @staticmethod
    def default():
        return MSMConfig(None, 1, 0, 0, 1, True, False, False, False, False, False, False, None)


class NTTConfig(C.Structure):
    _fields_ = [("stream", C.c_void_p), ("coset_gen", C.c_uint32 * 8), ("batch_size", C.c_int), ("columns_batch", C.c_bool),
                ("ordering", C.c_int), ("are_inputs_on_device", C.c_bool), ("are_outputs_on_device", C.c_bool),
                ("is_async", C.c_bool), ("ext", C.c_void_p)]

    @staticmethod
    def default():
        one = (C.c_uint32 * 8)(1, 0, 0, 0, 0, 0, 0, 0)
        return NTTConfig(None, one, 1, False, 0, False, False, False, None)


class NTTInitDomainConfig(C.Structure):
    _fields_ = [("stream", C.c_void_p), ("is_async", C.c_bool), ("ext", C.c_void_p)]


class VecOpsConfig(C.Structure):
    _fields_ = [("stream", C.c_void_p), ("is_a_on_device", C.c_bool), ("is_b_on_device", C.c_bool),
                ("is_result_on_device", C.c_bool), ("is_async", C.c_bool), ("batch_size", C.c_int),
                ("columns_batch", C.c_bool), ("ext", C.c_void_p)]

    @staticmethod
    def default():
        return VecOpsConfig(None, False, False, False, False, 1, False, None)


# every symbol include/icicle_snark_hip.h declares (checked by tests/test_abi.py)
DECLARED_SYMBOLS = """
icicle_load_backend icicle_load_backend_from_env_or_default icicle_set_device icicle_set_default_device
icicle_get_active_device icicle_is_host_memory icicle_is_active_device_memory icicle_get_device_count
icicle_is_device_available icicle_get_registered_devices icicle_malloc icicle_malloc_async icicle_free icicle_free_async
icicle_get_available_memory icicle_memset icicle_memset_async icicle_copy icicle_copy_async icicle_copy_to_host
icicle_copy_to_host_async icicle_copy_to_device icicle_copy_to_device_async icicle_create_stream icicle_destroy_stream
icicle_stream_synchronize icicle_device_synchronize icicle_get_device_properties
create_config_extension destroy_config_extension config_extension_set_int config_extension_set_bool
config_extension_get_int config_extension_get_bool clone_config_extension
bn254_generate_scalars bn254_add bn254_sub bn254_mul bn254_inv bn254_pow bn254_from_u32
bn254_eq bn254_ecadd bn254_ecsub bn254_mul_scalar bn254_to_affine bn254_from_affine bn254_generator bn254_is_on_curve
bn254_base_field_from_u32 bn254_g2_eq bn254_g2_ecadd bn254_g2_ecsub bn254_g2_mul_scalar bn254_g2_to_affine
bn254_g2_from_affine bn254_g2_generator bn254_g2_is_on_curve bn254_g2_base_field_from_u32
bn254_vector_add bn254_vector_sub bn254_vector_mul bn254_scalar_convert_montgomery
bn254_affine_convert_montgomery bn254_g2_affine_convert_montgomery
bn254_ntt bn254_ntt_init_domain bn254_ntt_release_domain bn254_get_root_of_unity bn254_get_root_of_unity_from_domain
bn254_msm bn254_g2_msm bn254_pairing
bn254_vector_div bn254_vector_accumulate bn254_vector_sum bn254_vector_product bn254_scalar_add_vec bn254_scalar_sub_vec
bn254_scalar_mul_vec bn254_projective_convert_montgomery bn254_g2_projective_convert_montgomery
bn254_msm_precompute_bases bn254_g2_msm_precompute_bases
bn254_pairing_target_field_add bn254_pairing_target_field_sub bn254_pairing_target_field_mul bn254_pairing_target_field_inv
bn254_pairing_target_field_pow bn254_pairing_target_field_from_u32 bn254_pairing_target_field_generate_scalars
icicle_snark_last_error icicle_snark_g1_generator_mul icicle_snark_g2_generator_mul icicle_snark_last_msm_timings
icicle_snark_msm_profile icicle_snark_microbench icicle_snark_pmc_probes icicle_snark_access_probes
""".split()

_lib = None


def lib():
    """Load the C-ABI library. Raises if it has not been built — there is no Python/CPU substitute."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make` (hipcc --offload-arch=gfx950); "
                              "this package has no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.icicle_snark_last_error.restype = C.c_char_p
        for n in ("bn254_eq", "bn254_g2_eq", "bn254_is_on_curve", "bn254_g2_is_on_curve"):
            getattr(_lib, n).restype = C.c_bool
    return _lib


def check(rc, what=""):
    if rc != SUCCESS:
        msg = lib().icicle_snark_last_error()
        raise IcicleError(rc, f"{what}: {msg.decode() if msg else ''}")


def set_device(dev_type: str = "HIP", dev_id: int = 0):
    """try_load_and_set_backend_device — src/lib.rs:25-31"""
    if dev_type != "CPU":
        check(lib().icicle_load_backend_from_env_or_default(), "load_backend")
    d = Device(dev_type.encode(), dev_id)
    check(lib().icicle_set_device(C.byref(d)), f"set_device({dev_type},{dev_id})")


def ptr_of(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, DeviceVec):
        return C.c_void_p(a.ptr)
    if isinstance(a, int):
        return C.c_void_p(a)
    raise TypeError(type(a))


def _on_dev(a):
    return isinstance(a, (DeviceVec, int))


class IcicleStream:
    """icicle-runtime/src/stream.rs"""

    def __init__(self, handle=None):
        if handle is None:
            h = C.c_void_p()
            check(lib().icicle_create_stream(C.byref(h)), "create_stream")
            self.handle, self._own = h.value, True
        else:
            self.handle, self._own = handle, False

    def synchronize(self):
        check(lib().icicle_stream_synchronize(C.c_void_p(self.handle)), "stream_synchronize")

    def destroy(self):
        if self._own and self.handle is not None:
            check(lib().icicle_destroy_stream(C.c_void_p(self.handle)), "destroy_stream")
            self.handle = None


class DeviceVec:
    """A device allocation of `nbytes` bytes (icicle-runtime/src/memory.rs:351-417). Slices share the parent."""

    def __init__(self, nbytes: int, stream: IcicleStream | None = None, _ptr=None, _parent=None):
        self.nbytes = nbytes
        self._parent = _parent
        if _ptr is not None:
            self.ptr = _ptr
            return
        p = C.c_void_p()
        if stream is None:
            check(lib().icicle_malloc(C.byref(p), C.c_size_t(max(nbytes, 1))), "malloc")
        else:
            check(lib().icicle_malloc_async(C.byref(p), C.c_size_t(max(nbytes, 1)), C.c_void_p(stream.handle)), "malloc_async")
        self.ptr = p.value

    @staticmethod
    def from_host(a: np.ndarray, stream: IcicleStream | None = None) -> "DeviceVec":
        a = np.ascontiguousarray(a)
        d = DeviceVec(a.nbytes, stream)
        d.copy_from_host(a, stream)
        return d

    def slice(self, byte_off: int, nbytes: int) -> "DeviceVec":
        assert 0 <= byte_off and byte_off + nbytes <= self.nbytes
        return DeviceVec(nbytes, _ptr=self.ptr + byte_off, _parent=self)

    def copy_from_host(self, a: np.ndarray, stream: IcicleStream | None = None):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        if stream is None:
            check(lib().icicle_copy_to_device(C.c_void_p(self.ptr), ptr_of(a), C.c_size_t(a.nbytes)), "copy_to_device")
        else:
            check(lib().icicle_copy_to_device_async(C.c_void_p(self.ptr), ptr_of(a), C.c_size_t(a.nbytes), C.c_void_p(stream.handle)), "copy_to_device_async")

    def to_host(self, shape, dtype=np.uint64, stream: IcicleStream | None = None) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if stream is None:
            check(lib().icicle_copy_to_host(ptr_of(out), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)), "copy_to_host")
        else:
            check(lib().icicle_copy_to_host_async(ptr_of(out), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), C.c_void_p(stream.handle)), "copy_to_host_async")
            stream.synchronize()
        return out

    def free(self):
        if self._parent is None and self.ptr:
            check(lib().icicle_free(C.c_void_p(self.ptr)), "free")
            self.ptr = 0


# --------------------------------------------------------------------------------------------- vec ops
def _vec_op(name, a, b, out, n, stream, is_async):
    cfg = VecOpsConfig.default()
    cfg.is_a_on_device, cfg.is_b_on_device, cfg.is_result_on_device = _on_dev(a), _on_dev(b), _on_dev(out)
    cfg.is_async = is_async
    cfg.stream = stream.handle if stream else None
    check(getattr(lib(), name)(ptr_of(a), ptr_of(b), C.c_uint64(n), C.byref(cfg), ptr_of(out)), name)


def _n_of(a):
    return a.size // 4 if isinstance(a, np.ndarray) else a.nbytes // 32


def mul_scalars(a, b, out=None, stream=None, is_async=False):
    if out is None:
        out = np.empty((_n_of(a), 4), dtype=np.uint64)
    _vec_op("bn254_vector_mul", a, b, out, _n_of(a), stream, is_async)
    return out


def sub_scalars(a, b, out=None, stream=None, is_async=False):
    if out is None:
        out = np.empty((_n_of(a), 4), dtype=np.uint64)
    _vec_op("bn254_vector_sub", a, b, out, _n_of(a), stream, is_async)
    return out


def add_scalars(a, b, out=None, stream=None, is_async=False):
    if out is None:
        out = np.empty((_n_of(a), 4), dtype=np.uint64)
    _vec_op("bn254_vector_add", a, b, out, _n_of(a), stream, is_async)
    return out


def vec_op2(name: str, a, b, n=None, batch_size=1, columns_batch=False, out=None, stream=None, is_async=False):
    """element-wise a ∘ b for name in add/sub/mul/div over n·batch_size elements (icicle-core vec_ops/mod.rs)"""
    total = _n_of(a)
    if out is None:
        out = np.empty((total, 4), dtype=np.uint64)
    cfg = VecOpsConfig.default()
    cfg.is_a_on_device, cfg.is_b_on_device, cfg.is_result_on_device = _on_dev(a), _on_dev(b), _on_dev(out)
    cfg.is_async, cfg.batch_size, cfg.columns_batch = is_async, batch_size, columns_batch
    cfg.stream = stream.handle if stream else None
    n = total // batch_size if n is None else n
    check(getattr(lib(), "bn254_vector_" + name)(ptr_of(a), ptr_of(b), C.c_uint64(n), C.byref(cfg), ptr_of(out)), name)
    return out


def accumulate_scalars(a, b, stream=None, is_async=False):
    """a += b in place — vector_accumulate"""
    cfg = VecOpsConfig.default()
    cfg.is_a_on_device, cfg.is_b_on_device, cfg.is_result_on_device = _on_dev(a), _on_dev(b), _on_dev(a)
    cfg.is_async = is_async
    cfg.stream = stream.handle if stream else None
    check(lib().bn254_vector_accumulate(ptr_of(a), ptr_of(b), C.c_uint64(_n_of(a)), C.byref(cfg)), "vector_accumulate")
    return a


def scalar_vec_op(name: str, scalars, v, batch_size=1, columns_batch=False, out=None, stream=None, is_async=False):
    """out(b, i) = scalars[b] ∘ v(b, i) for name in add/sub/mul — scalar_{add,sub,mul}_vec"""
    total = _n_of(v)
    if out is None:
        out = np.empty((total, 4), dtype=np.uint64)
    cfg = VecOpsConfig.default()
    cfg.is_a_on_device, cfg.is_b_on_device, cfg.is_result_on_device = _on_dev(scalars), _on_dev(v), _on_dev(out)
    cfg.is_async, cfg.batch_size, cfg.columns_batch = is_async, batch_size, columns_batch
    cfg.stream = stream.handle if stream else None
    check(getattr(lib(), f"bn254_scalar_{name}_vec")(ptr_of(scalars), ptr_of(v), C.c_uint64(total // batch_size), C.byref(cfg), ptr_of(out)), name)
    return out


def reduce_scalars(name: str, v, batch_size=1, columns_batch=False, stream=None, is_async=False):
    """Σ / Π of every batch vector, name in sum/product → (batch_size, 4) array"""
    out = np.empty((batch_size, 4), dtype=np.uint64)
    cfg = VecOpsConfig.default()
    cfg.is_a_on_device, cfg.is_result_on_device = _on_dev(v), False
    cfg.is_async, cfg.batch_size, cfg.columns_batch = is_async, batch_size, columns_batch
    cfg.stream = stream.handle if stream else None
    check(getattr(lib(), "bn254_vector_" + name)(ptr_of(v), C.c_uint64(_n_of(v) // batch_size), C.byref(cfg), ptr_of(out)), name)
    return out


def projective_convert_montgomery(group: str, a: np.ndarray, to_mont: bool) -> np.ndarray:
    per = 96 if group == "g1" else 192
    out = np.empty_like(a)
    cfg = VecOpsConfig.default()
    name = "bn254_projective_convert_montgomery" if group == "g1" else "bn254_g2_projective_convert_montgomery"
    check(getattr(lib(), name)(ptr_of(a), C.c_size_t(a.nbytes // per), C.c_bool(to_mont), C.byref(cfg), ptr_of(out)), name)
    return out


def scalar_convert_montgomery(a, to_mont: bool, out=None, stream=None, is_async=False):
    if out is None:
        out = np.empty_like(a) if isinstance(a, np.ndarray) else a
    cfg = VecOpsConfig.default()
    cfg.is_a_on_device, cfg.is_result_on_device = _on_dev(a), _on_dev(out)
    cfg.is_async = is_async
    cfg.stream = stream.handle if stream else None
    check(lib().bn254_scalar_convert_montgomery(ptr_of(a), C.c_uint64(_n_of(a)), C.c_bool(to_mont), C.byref(cfg), ptr_of(out)), "scalar_convert_montgomery")
    return out


def affine_convert_montgomery(group: str, a, to_mont: bool, out=None, stream=None, is_async=False):
    per = 64 if group == "g1" else 128
    n = (a.nbytes if isinstance(a, (np.ndarray, DeviceVec)) else 0) // per
    if out is None:
        out = np.empty_like(a) if isinstance(a, np.ndarray) else a
    cfg = VecOpsConfig.default()
    cfg.is_a_on_device, cfg.is_result_on_device = _on_dev(a), _on_dev(out)
    cfg.is_async = is_async
    cfg.stream = stream.handle if stream else None
    name = "bn254_affine_convert_montgomery" if group == "g1" else "bn254_g2_affine_convert_montgomery"
    check(getattr(lib(), name)(ptr_of(a), C.c_uint64(n), C.c_bool(to_mont), C.byref(cfg), ptr_of(out)), name)
    return out


# --------------------------------------------------------------------------------------------- NTT
def get_root_of_unity(max_size: int) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    check(lib().bn254_get_root_of_unity(C.c_uint64(max_size), ptr_of(out)), "get_root_of_unity")
    return out


def initialize_domain(root: np.ndarray, stream=None):
    cfg = NTTInitDomainConfig(stream.handle if stream else None, False, None)
    check(lib().bn254_ntt_init_domain(ptr_of(np.ascontiguousarray(root)), C.byref(cfg)), "ntt_init_domain")


def release_domain():
    check(lib().bn254_ntt_release_domain(), "ntt_release_domain")


def ntt(inp, inverse: bool, out=None, batch_size=1, size=None, stream=None, is_async=False, coset_gen=None, ordering=0, columns_batch=False):
    if size is None:
        size = _n_of(inp) // batch_size
    if out is None:
        out = np.empty_like(inp) if isinstance(inp, np.ndarray) else inp
    cfg = NTTConfig.default()
    cfg.batch_size = batch_size
    cfg.are_inputs_on_device, cfg.are_outputs_on_device = _on_dev(inp), _on_dev(out)
    cfg.is_async = is_async
    cfg.ordering = ordering
    cfg.columns_batch = columns_batch
    cfg.stream = stream.handle if stream else None
    if coset_gen is not None:
        cg = np.ascontiguousarray(coset_gen, dtype=np.uint64).view(np.uint32)
        for i in range(8):
            cfg.coset_gen[i] = int(cg[i])
    check(lib().bn254_ntt(ptr_of(inp), C.c_int(size), C.c_int(1 if inverse else 0), C.byref(cfg), ptr_of(out)), "ntt")
    return out


# --------------------------------------------------------------------------------------------- MSM
def msm(group: str, scalars, bases, out=None, stream=None, is_async=False, c=0, size=None,
        scalars_mont=False, points_mont=False, ext=None, batch_size=1, shared_points=True, precompute_factor=1, bitsize=0):
    """msm() — icicle-core/src/msm/mod.rs:106-154. Returns the projective result (3,4)/(6,4) u64 when `out` is None."""
    if size is None:
        size = _n_of(scalars) // batch_size
    host_out = out is None
    if host_out:
        out = np.zeros(((3, 4) if group == "g1" else (6, 4)) if batch_size == 1 else ((batch_size, 3, 4) if group == "g1" else (batch_size, 6, 4)), dtype=np.uint64)
    cfg = MSMConfig.default()
    cfg.c, cfg.bitsize = c, bitsize
    cfg.batch_size, cfg.are_points_shared_in_batch, cfg.precompute_factor = batch_size, shared_points, precompute_factor
    cfg.are_scalars_on_device, cfg.are_points_on_device, cfg.are_results_on_device = _on_dev(scalars), _on_dev(bases), _on_dev(out)
    cfg.are_scalars_montgomery_form, cfg.are_points_montgomery_form = scalars_mont, points_mont
    cfg.is_async = is_async
    cfg.stream = stream.handle if stream else None
    cfg.ext = ext
    name = "bn254_msm" if group == "g1" else "bn254_g2_msm"
    check(getattr(lib(), name)(ptr_of(scalars), ptr_of(bases), C.c_int(size), C.byref(cfg), ptr_of(out)), name)
    return out


def msm_precompute_bases(group: str, bases: np.ndarray, precompute_factor: int, c=0, points_mont=False, bitsize=0) -> np.ndarray:
    """msm_precompute_bases — icicle-core/src/msm/mod.rs:156-190 (host arrays in and out)"""
    per = 64 if group == "g1" else 128
    n = bases.nbytes // per
    out = np.empty((n * precompute_factor,) + bases.shape[1:], dtype=np.uint64)
    cfg = MSMConfig.default()
    cfg.c, cfg.precompute_factor, cfg.are_points_montgomery_form, cfg.bitsize = c, precompute_factor, points_mont, bitsize
    name = "bn254_msm_precompute_bases" if group == "g1" else "bn254_g2_msm_precompute_bases"
    check(getattr(lib(), name)(ptr_of(np.ascontiguousarray(bases)), C.c_int(n), C.byref(cfg), ptr_of(out)), name)
    return out


def raw_to_host(ptr: int, nbytes: int) -> bytes:
    """device memory at a raw pointer → bytes (exchange emulation in tests / the gloo fallback)"""
    buf = (C.c_uint8 * nbytes)()
    check(lib().icicle_copy_to_host(buf, C.c_void_p(ptr), C.c_size_t(nbytes)), "copy_to_host")
    return bytes(buf)


def raw_to_device(ptr: int, data: bytes):
    check(lib().icicle_copy_to_device(C.c_void_p(ptr), data, C.c_size_t(len(data))), "copy_to_device")


def last_msm_timings():
    out = (C.c_float * 4)()
    check(lib().icicle_snark_last_msm_timings(out), "last_msm_timings")
    return list(out)


def msm_profile(back: int = 0):
    """(ms[5], dict geom) of the `back`-th most recent MSM; see include/icicle_snark_hip.h."""
    ms = (C.c_float * 5)()
    geom = (C.c_uint32 * 5)()
    check(lib().icicle_snark_msm_profile(back, ms, geom), "msm_profile")
    return list(ms), dict(L=geom[0], nbuckets=geom[1], c=geom[2], W=geom[3], is_g2=bool(geom[4]))


def microbench():
    """(device-to-device copy GB/s counting read + write, v_mad_u64_u32 lane-ops/s in 10^12) measured now"""
    out = (C.c_double * 2)()
    check(lib().icicle_snark_microbench(out), "microbench")
    return float(out[0]), float(out[1])


def access_probes():
    """GB/s of 64-byte gathers, 128-byte gathers, coalesced reads, scattered 4-byte stores, coalesced 4-byte stores (2 GiB buffer), measured now"""
    out = (C.c_double * 5)()
    check(lib().icicle_snark_access_probes(out), "access_probes")
    return dict(zip(("gather64_gbps", "gather128_gbps", "stream_read_gbps", "scattered_store_gbps", "coalesced_store_gbps"), (round(float(x), 1) for x in out)))


def generator_mul(group: str, scalars: np.ndarray) -> np.ndarray:
    """out[i] = s[i]·G, affine standard form (extension; used by the zkey synthesiser)."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = scalars.shape[0]
    dims = 2 if group == "g1" else 4
    d_s = DeviceVec.from_host(scalars)
    d_o = DeviceVec(n * dims * 32)
    name = "icicle_snark_g1_generator_mul" if group == "g1" else "icicle_snark_g2_generator_mul"
    check(getattr(lib(), name)(C.c_void_p(d_s.ptr), C.c_uint64(n), None, C.c_void_p(d_o.ptr)), name)
    check(lib().icicle_device_synchronize(), "sync")
    out = d_o.to_host((n, dims, 4))
    d_s.free()
    d_o.free()
    return out


# --------------------------------------------------------------------------------------------- host FFI
def i2a(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint64).copy()


def a2i(a):
    return int.from_bytes(np.ascontiguousarray(a).tobytes(), "little")


def fr_op(op: str, a: int, b: int) -> int:
    out = np.zeros(4, dtype=np.uint64)
    getattr(lib(), f"bn254_{op}")(ptr_of(i2a(a)), ptr_of(i2a(b)), ptr_of(out))
    return a2i(out)


def fr_inv(a: int) -> int:
    out = np.zeros(4, dtype=np.uint64)
    lib().bn254_inv(ptr_of(i2a(a)), ptr_of(out))
    return a2i(out)


_PRE = {"g1": "bn254_", "g2": "bn254_g2_"}
_DIMS = {"g1": (2, 3), "g2": (4, 6)}


def ec(group: str, op: str, *args):
    """host curve FFI: ecadd/ecsub(p,q), mul_scalar(p, int), to_affine(p), from_affine(a), generator()"""
    f = getattr(lib(), _PRE[group] + op)
    na, npj = _DIMS[group]
    if op in ("ecadd", "ecsub"):
        out = np.zeros((npj, 4), dtype=np.uint64)
        f(ptr_of(np.ascontiguousarray(args[0])), ptr_of(np.ascontiguousarray(args[1])), ptr_of(out))
    elif op == "mul_scalar":
        out = np.zeros((npj, 4), dtype=np.uint64)
        f(ptr_of(np.ascontiguousarray(args[0])), ptr_of(i2a(args[1])), ptr_of(out))
    elif op == "to_affine":
        out = np.zeros((na, 4), dtype=np.uint64)
        f(ptr_of(np.ascontiguousarray(args[0])), ptr_of(out))
    elif op == "from_affine":
        out = np.zeros((npj, 4), dtype=np.uint64)
        f(ptr_of(np.ascontiguousarray(args[0])), ptr_of(out))
    elif op == "generator":
        out = np.zeros((npj, 4), dtype=np.uint64)
        f(ptr_of(out))
    else:
        raise ValueError(op)
    return out


def ec_eq(group, a, b) -> bool:
    return bool(getattr(lib(), _PRE[group] + "eq")(ptr_of(np.ascontiguousarray(a)), ptr_of(np.ascontiguousarray(b))))


def ec_is_on_curve(group, a) -> bool:
    return bool(getattr(lib(), _PRE[group] + "is_on_curve")(ptr_of(np.ascontiguousarray(a))))


# --------------------------------------------------------------------------------------------- prover host
COMMITMENTS_BYTES = 576


class Timings(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("qap_ms", C.c_double), ("msm_ms", C.c_double), ("total_ms", C.c_double)]


class CircuitInfo(C.Structure):
    _fields_ = [("n_vars", C.c_uint32), ("n_public", C.c_uint32), ("domain_size", C.c_uint32), ("n_coef", C.c_uint32),
                ("device_bytes", C.c_uint64), ("b_bases", C.c_uint32), ("shards", C.c_uint32)]


class ProverError(RuntimeError):
    pass


def _pcheck(rc, what):
    if rc != 0:
        lib().groth16_last_error.restype = C.c_char_p
        msg = lib().groth16_last_error()
        raise ProverError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


class CacheManager:
    """CacheManager — src/cache.rs:110-262 (include/groth16_prover.h)."""

    def __init__(self):
        lib().groth16_cache_manager_new.restype = C.c_void_p
        self._h = C.c_void_p(lib().groth16_cache_manager_new())

    def close(self):
        if self._h:
            lib().groth16_cache_manager_free(self._h)
            self._h = None

    def contains(self, key: str) -> bool:
        return bool(lib().groth16_cache_contains(self._h, key.encode()))

    def load(self, key: str, zkey: bytes, device_id: int = 0, shard_rank: int = 0, shard_count: int = 1, wait_tables: bool = True):
        """groth16_cache_load.  A single-device key is usable before its fixed-base tables exist (they are built behind the
        first proofs); wait_tables=True (the default of this binding: tests and timings want the final layout) blocks until
        they are adopted, wait_tables=False returns like the C entry point does."""
        # zero-copy views of the caller's buffer (a 0.8 GB zkey must not be duplicated on the way in)
        if isinstance(zkey, np.ndarray):
            p = zkey.ctypes.data_as(C.c_void_p)
        elif isinstance(zkey, bytes):
            p = C.c_char_p(zkey)
        else:
            p = (C.c_char * len(zkey)).from_buffer(zkey)   # bytearray / writable memoryview
        _pcheck(lib().groth16_cache_load(self._h, key.encode(), p, C.c_size_t(len(zkey)), device_id, shard_rank, shard_count), "cache_load")
        if wait_tables:
            self.tables_ready(key, wait=True)

    def load_devices(self, key: str, zkey: bytes, device_ids):
        """one key over a GROUP of devices in this process (what groth16_prove builds for "HIP:0-7"): shard k lives on
        device_ids[k]; a device may be named several times (several shards on one GPU)."""
        ids = (C.c_int * len(device_ids))(*device_ids)
        p = zkey.ctypes.data_as(C.c_void_p) if isinstance(zkey, np.ndarray) else C.c_char_p(zkey) if isinstance(zkey, bytes) else (C.c_char * len(zkey)).from_buffer(zkey)
        _pcheck(lib().groth16_cache_load_devices(self._h, key.encode(), p, C.c_size_t(len(zkey)), ids, len(device_ids)), "cache_load_devices")

    def set_budget(self, bytes_per_device: int):
        lib().groth16_cache_set_budget(self._h, C.c_uint64(bytes_per_device))

    def load_file(self, key: str, path: str, device_id: int = 0, shard_rank: int = 0, shard_count: int = 1, wait_tables: bool = True):
        _pcheck(lib().groth16_cache_load_file(self._h, key.encode(), path.encode(), device_id, shard_rank, shard_count), "cache_load_file")
        if wait_tables:
            self.tables_ready(key, wait=True)

    def evict(self, key: str):
        lib().groth16_cache_evict(self._h, key.encode())

    def tables_ready(self, key: str, wait: bool = False) -> bool:
        """deferred fixed-base tables of a single-device key (groth16_cache_tables_ready): True once the key proves in its
        final layout; wait=True blocks until the build behind the first proofs has ended"""
        rc = lib().groth16_cache_tables_ready(self._h, key.encode(), int(wait))
        if rc < 0:
            _pcheck(rc, "cache_tables_ready")
        return rc == 1

    def info(self, key: str) -> CircuitInfo:
        ci = CircuitInfo()
        _pcheck(lib().groth16_cache_info(self._h, key.encode(), C.byref(ci)), "cache_info")
        return ci

    def group_describe(self, key: str) -> dict:
        """what the key runs on: shards, devices, exchange transport, RCCL ranks (groth16_group_describe)"""
        import json
        buf = C.create_string_buffer(2048)
        _pcheck(lib().groth16_group_describe(self._h, key.encode(), buf, C.c_size_t(len(buf))), "group_describe")
        return json.loads(buf.value.decode())

    def commitments(self, key: str, wtns: bytes | None):
        """groth16_commitments (incl. construct_r1cs) for this process's shard → (576-byte block, Timings).
        wtns=None re-uses the witness already resident on the device."""
        out = (C.c_uint8 * COMMITMENTS_BYTES)()
        tm = Timings()
        _pcheck(lib().groth16_commitments(self._h, key.encode(), wtns, C.c_size_t(len(wtns) if wtns else 0), out, C.byref(tm)), "commitments")
        return bytes(out), tm

    def assemble(self, key: str, wtns: bytes, points: bytes, r: int | None = None, s: int | None = None):
        pj, qj = C.create_string_buffer(1 << 14), C.create_string_buffer(1 << 20)
        rb = int(r).to_bytes(32, "little") if r is not None else None
        sb = int(s).to_bytes(32, "little") if s is not None else None
        _pcheck(lib().groth16_assemble_proof(self._h, key.encode(), wtns, C.c_size_t(len(wtns)), points, rb, sb,
                                             pj, C.c_size_t(len(pj)), qj, C.c_size_t(len(qj))), "assemble_proof")
        return pj.value.decode(), qj.value.decode()

    def prove_mem(self, key: str, wtns: bytes, r: int | None = None, s: int | None = None, resident: bool = False):
        """commitments + blinding + JSON in one call (the blinding terms are computed on a host thread while the GPU
        works).  resident=True re-uses the witness already uploaded by an earlier call."""
        if not hasattr(self, "_bufs"):
            self._bufs = (C.create_string_buffer(1 << 14), C.create_string_buffer(1 << 20))
        pj, qj = self._bufs
        rb = int(r).to_bytes(32, "little") if r is not None else None
        sb = int(s).to_bytes(32, "little") if s is not None else None
        tm = Timings()
        _pcheck(lib().groth16_prove_resident(self._h, key.encode(), wtns, C.c_size_t(len(wtns)), int(resident), rb, sb, pj,
                                             C.c_size_t(len(pj)), qj, C.c_size_t(len(qj)), C.byref(tm)), "prove_resident")
        return pj.value.decode(), qj.value.decode(), tm

    # ---- distributed QAP front end (2, 4 or 8 strided shards): see include/groth16_prover.h ----
    def dist_supported(self, key: str) -> bool:
        return bool(lib().groth16_dist_supported(self._h, key.encode()))

    def dist_stage1(self, key: str, wtns: bytes | None):
        """→ (send ptr, recv ptr, rows, row_bytes, chunk_bytes): device buffers of exchange 1.  wtns=None: the witness is
        already resident (upload_witness_slice + all-gather + witness_ready)"""
        send, recv = C.c_void_p(), C.c_void_p()
        rows, rb, cb = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _pcheck(lib().groth16_dist_stage1(self._h, key.encode(), wtns, C.c_size_t(len(wtns) if wtns else 0), C.byref(send), C.byref(recv), C.byref(rows), C.byref(rb), C.byref(cb)), "dist_stage1")
        return send.value, recv.value, rows.value, rb.value, cb.value

    def upload_witness_slice(self, key: str, wtns: bytes):
        """this rank's 1/shard_count of the witness → its place in the device witness buffer.
        → (device pointer of the whole buffer, bytes per rank of the in-place all-gather that completes it)"""
        ptr, sb = C.c_void_p(), C.c_uint64()
        _pcheck(lib().groth16_upload_witness_slice(self._h, key.encode(), wtns, C.c_size_t(len(wtns)), C.byref(ptr), C.byref(sb)), "upload_witness_slice")
        return ptr.value, sb.value

    def witness_ready(self, key: str):
        _pcheck(lib().groth16_witness_ready(self._h, key.encode()), "witness_ready")

    def dist_stage2(self, key: str):
        """→ (send ptr, recv ptr) of exchange 2 (same geometry as exchange 1)"""
        send, recv = C.c_void_p(), C.c_void_p()
        _pcheck(lib().groth16_dist_stage2(self._h, key.encode(), C.byref(send), C.byref(recv)), "dist_stage2")
        return send.value, recv.value

    def dist_exchange_done(self, key: str):
        """the caller confirms that exchange 2 delivered: the next commitments(key, None) finishes from those rows"""
        _pcheck(lib().groth16_dist_exchange_done(self._h, key.encode()), "dist_exchange_done")

    def prove_files(self, witness: str, zkey: str, proof: str, public: str, device: str = "HIP"):
        """groth16_prove — src/lib.rs:33-61: files in, files out (the reference's timed region)"""
        return self.prove(witness, zkey, proof, public, device)

    def last_timings(self, key: str) -> Timings:
        tm = Timings()
        _pcheck(lib().groth16_last_timings(self._h, key.encode(), C.byref(tm)), "last_timings")
        return tm

    def prove(self, witness: str, zkey: str, proof: str, public: str, device: str = "HIP"):
        """groth16_prove — src/lib.rs:33-61"""
        _pcheck(lib().groth16_prove(witness.encode(), zkey.encode(), proof.encode(), public.encode(), device.encode(), self._h), "groth16_prove")


PROVER_SYMBOLS = """
groth16_cache_manager_new groth16_cache_manager_free groth16_prove groth16_cache_load groth16_cache_load_file
groth16_cache_contains groth16_cache_evict groth16_commitments groth16_sum_commitments groth16_assemble_proof
groth16_prove_mem groth16_prove_resident groth16_cache_info groth16_last_error groth16_last_timings
groth16_dist_supported groth16_dist_stage1 groth16_dist_stage2 groth16_dist_exchange_done groth16_upload_witness_slice
groth16_witness_ready groth16_cache_load_devices groth16_parse_device groth16_cache_set_budget groth16_cache_info_sized
groth16_verify groth16_verify_json groth16_verify_last_error groth16_group_describe groth16_cache_tables_ready
groth16_cache_manager_prewarm
""".split()


def parse_device(device: str, cap: int = 64):
    """device string of groth16_prove → list of device ids ("HIP:0-7" → [0, …, 7]); raises ProverError for a bad string"""
    ids = (C.c_int * cap)()
    n = lib().groth16_parse_device(device.encode(), ids, cap)
    if n < 0:
        _pcheck(n, "parse_device")
    return [ids[i] for i in range(min(n, cap))]


def pairing(p_aff: np.ndarray, q_aff: np.ndarray) -> np.ndarray:
    """pairing — wrappers/rust/icicle-core/src/pairing/mod.rs:14-22.  Affine standard-form points in, the 12 Fq
    coefficients of e(P,Q) out (host computation, as in the reference)."""
    out = np.zeros((12, 4), dtype=np.uint64)
    check(lib().bn254_pairing(ptr_of(np.ascontiguousarray(p_aff)), ptr_of(np.ascontiguousarray(q_aff)), ptr_of(out)), "pairing")
    return out


def gt_op(op: str, a: np.ndarray, b=None) -> np.ndarray:
    """TargetField arithmetic (bn254_pairing_target_field_{add,sub,mul,inv,pow}); `b` is an int for pow."""
    out = np.zeros((12, 4), dtype=np.uint64)
    f = getattr(lib(), "bn254_pairing_target_field_" + op)
    a = np.ascontiguousarray(a)
    if op == "inv":
        f(ptr_of(a), ptr_of(out))
    elif op == "pow":
        f(ptr_of(a), C.c_int(int(b)), ptr_of(out))
    else:
        f(ptr_of(a), ptr_of(np.ascontiguousarray(b)), ptr_of(out))
    return out


def groth16_verify_json(proof_json: str, public_json: str, vk_json: str) -> bool:
    """groth16_verify_helper — src/proof_helper.rs:319-372 on JSON texts."""
    rc = lib().groth16_verify_json(proof_json.encode(), public_json.encode(), vk_json.encode())
    if rc < 0:
        lib().groth16_verify_last_error.restype = C.c_char_p
        raise ProverError(f"groth16_verify: {lib().groth16_verify_last_error().decode()} (code {rc})")
    return rc == 1


def groth16_verify(proof: str, public: str, vk: str):
    """groth16_verify — src/lib.rs:63-82 (paths in; raises on a rejected proof like the reference's assert)."""
    rc = lib().groth16_verify(proof.encode(), public.encode(), vk.encode())
    if rc != 0:
        lib().groth16_verify_last_error.restype = C.c_char_p
        raise ProverError(f"groth16_verify: {lib().groth16_verify_last_error().decode()} (code {rc})")


def sum_commitments(blocks: bytes, count: int) -> bytes:
    out = (C.c_uint8 * COMMITMENTS_BYTES)()
    _pcheck(lib().groth16_sum_commitments(blocks, count, out), "sum_commitments")
    return bytes(out)
