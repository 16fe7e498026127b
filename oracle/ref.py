"""ctypes front end of oracle/_ref/libicicle_ref.so — the reference's OWN sources compiled
where they lie (oracle/Makefile `ref`).  TEST INFRASTRUCTURE ONLY.

Used (a) to pin the C restatement (tests/test_oracle_vs_ref.py, tests/golden/make_golden.py)
and (b) as an oracle-independent accept/reject on proofs through the reference pairing
(groth16_verify_helper, src/proof_helper.rs:319-372).  The library is optional at run time:
`available()` is False when it has not been built (e.g. a checkout without /root/reference).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libicicle_ref.so")
_lib = None


def available() -> bool:
    # The reference build is used only where the reference tree itself is present (the build container): on the GPU box
    # the committed fixtures under tests/golden/ carry the pin, even if a prebuilt _ref travelled along.
    return os.path.exists(_PATH) and (os.path.isdir("/root/reference") or os.environ.get("ICICLE_SNARK_USE_REF") == "1")


class Device(C.Structure):
    _fields_ = [("type", C.c_char * 64), ("id", C.c_int)]


class VecOpsConfig(C.Structure):
    _fields_ = [("stream", C.c_void_p), ("is_a_on_device", C.c_bool), ("is_b_on_device", C.c_bool),
                ("is_result_on_device", C.c_bool), ("is_async", C.c_bool), ("batch_size", C.c_int),
                ("columns_batch", C.c_bool), ("ext", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        dev = Device(b"CPU", 0)
        rc = _lib.icicle_set_device(C.byref(dev))
        assert rc == 0, rc
    return _lib


def _p(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _new(*shape):
    return np.zeros(shape, dtype=np.uint64)


def _i2a(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint64).copy()


def _a2i(a):
    return int.from_bytes(np.ascontiguousarray(a).tobytes(), "little")


def fr_bin(op: str, a: int, b: int) -> int:
    out = _new(4)
    getattr(lib(), f"bn254_{op}")(_p(_i2a(a)), _p(_i2a(b)), _p(out))
    return _a2i(out)


def fr_inv(a: int) -> int:
    out = _new(4)
    lib().bn254_inv(_p(_i2a(a)), _p(out))
    return _a2i(out)


def get_root_of_unity(max_size: int) -> int:
    out = _new(4)
    rc = lib().bn254_get_root_of_unity(C.c_uint64(max_size), _p(out))
    assert rc == 0
    return _a2i(out)


_PRE = {"g1": "bn254_", "g2": "bn254_g2_"}
_DIMS = {"g1": (2, 3), "g2": (4, 6)}


def ec(group: str, op: str, *args):
    """op in ecadd/ecsub (proj,proj→proj), mul_scalar (proj, int → proj), to_affine, from_affine, generator"""
    f = getattr(lib(), _PRE[group] + op)
    na, npj = _DIMS[group]
    if op in ("ecadd", "ecsub"):
        out = _new(npj, 4)
        f(_p(np.ascontiguousarray(args[0])), _p(np.ascontiguousarray(args[1])), _p(out))
    elif op == "mul_scalar":
        out = _new(npj, 4)
        f(_p(np.ascontiguousarray(args[0])), _p(_i2a(args[1])), _p(out))
    elif op == "to_affine":
        out = _new(na, 4)
        f(_p(np.ascontiguousarray(args[0])), _p(out))
    elif op == "from_affine":
        out = _new(npj, 4)
        f(_p(np.ascontiguousarray(args[0])), _p(out))
    elif op == "generator":
        out = _new(npj, 4)
        f(_p(out))
    else:
        raise ValueError(op)
    return out


def ec_eq(group, a, b) -> bool:
    f = getattr(lib(), _PRE[group] + "eq")
    f.restype = C.c_bool
    return bool(f(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b))))


def ec_is_on_curve(group, a) -> bool:
    f = getattr(lib(), _PRE[group] + "is_on_curve")
    f.restype = C.c_bool
    return bool(f(_p(np.ascontiguousarray(a))))


def convert_montgomery(kind: str, a: np.ndarray, to_mont: bool) -> np.ndarray:
    """kind: 'g1' (n,2,4) affine, 'g2' (n,4,4) affine — CPU Montgomery-conversion backend of the reference
    (cpu_mont_conversion.cpp).  The scalar variant lives in cpu_vec_ops.cpp (Taskflow) and is not in this build."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    cfg = VecOpsConfig(None, False, False, False, False, 1, False, None)
    name = {"g1": "bn254_affine_convert_montgomery", "g2": "bn254_g2_affine_convert_montgomery"}[kind]
    n = a.shape[0]
    rc = getattr(lib(), name)(_p(a), C.c_uint64(n), C.c_bool(to_mont), C.byref(cfg), _p(out))
    assert rc == 0, rc
    return out


def msm_naive(group, scalars: np.ndarray, bases: np.ndarray):
    """Σ sᵢ·Pᵢ using only the reference's from_affine / mul_scalar / ecadd."""
    scalars = np.ascontiguousarray(scalars).reshape(-1, 4)
    acc = None
    for i in range(scalars.shape[0]):
        p = ec(group, "from_affine", bases[i])
        t = ec(group, "mul_scalar", p, _a2i(scalars[i]))
        acc = t if acc is None else ec(group, "ecadd", acc, t)
    return acc


def dft_naive(x: np.ndarray, w: int) -> np.ndarray:
    """out[k] = Σ_j x[j]·w^(jk) using only the reference's bn254_mul / bn254_add."""
    xs = [_a2i(r) for r in np.ascontiguousarray(x).reshape(-1, 4)]
    n = len(xs)
    out = []
    for k in range(n):
        step, wk, acc = 1, 1, 0
        for _ in range(k):
            step = fr_bin("mul", step, w)
        for j in range(n):
            acc = fr_bin("add", acc, fr_bin("mul", xs[j], wk))
            wk = fr_bin("mul", wk, step)
        out.append(acc)
    return np.stack([_i2a(v) for v in out])


# ----------------------------------------------------------------------------- verification
def _strs_to_g1(v):
    return np.stack([_i2a(int(v[0])), _i2a(int(v[1]))])


def _strs_to_g2(v):
    return np.stack([_i2a(int(v[0][0])), _i2a(int(v[0][1])), _i2a(int(v[1][0])), _i2a(int(v[1][1]))])


def pairing(p_aff: np.ndarray, q_aff: np.ndarray) -> np.ndarray:
    out = _new(12, 4)
    lib().bn254_pairing(_p(np.ascontiguousarray(p_aff)), _p(np.ascontiguousarray(q_aff)), _p(out))
    return out


def gt_mul(a, b):
    out = _new(12, 4)
    lib().bn254_pairing_target_field_mul(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
    return out


def gt_op(op: str, a, b=None):
    """bn254_pairing_target_field_{add,sub,mul,inv,pow} — icicle/src/fields/ffi_extern_pairing_extension.cpp"""
    out = _new(12, 4)
    f = getattr(lib(), "bn254_pairing_target_field_" + op)
    a = np.ascontiguousarray(a)
    if op == "inv":
        f(_p(a), _p(out))
    elif op == "pow":
        f(_p(a), C.c_int(int(b)), _p(out))
    else:
        f(_p(a), _p(np.ascontiguousarray(b)), _p(out))
    return out


def groth16_verify(proof: dict, public: list, vk: dict) -> bool:
    """groth16_verify_helper — src/proof_helper.rs:319-372:
    e(-A,B)·e(cpub,γ₂)·e(C,δ₂)·e(α₁,β₂) == 1.  `vk` holds affine numpy points
    (vk_alpha_1, vk_beta_2, vk_gamma_2, vk_delta_2, IC list) in standard form."""
    pi_a, pi_b, pi_c = _strs_to_g1(proof["pi_a"]), _strs_to_g2(proof["pi_b"]), _strs_to_g1(proof["pi_c"])
    cpub = ec("g1", "from_affine", vk["IC"][0])
    for i, s in enumerate(public):
        cpub = ec("g1", "ecadd", cpub, ec("g1", "mul_scalar", ec("g1", "from_affine", vk["IC"][i + 1]), int(s)))
    zero = _new(3, 4)
    zero[1, 0] = 1
    neg_a = ec("g1", "to_affine", ec("g1", "ecsub", zero, ec("g1", "from_affine", pi_a)))
    acc = pairing(neg_a, pi_b)
    acc = gt_mul(acc, pairing(ec("g1", "to_affine", cpub), vk["vk_gamma_2"]))
    acc = gt_mul(acc, pairing(pi_c, vk["vk_delta_2"]))
    acc = gt_mul(acc, pairing(vk["vk_alpha_1"], vk["vk_beta_2"]))
    one = _new(12, 4)
    one[0, 0] = 1
    return bool(np.array_equal(acc, one))
