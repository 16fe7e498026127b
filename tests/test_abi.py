"""C-ABI checks that need no GPU: the library loads, exports every symbol include/icicle_snark_hip.h
declares, struct layouts match the reference's repr(C) structs, and the host-side field/curve FFI
(host functions in the reference too, icicle/src/{fields,curves}/ffi_extern.cpp) reproduces the golden
vectors bit for bit.  No device compute is issued here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, unhex, unhex_int

DIMS = {"g1": (2, 3), "g2": (4, 6)}


def test_library_exports_every_declared_symbol(K):
    lib = K.lib()
    hdr = open(os.path.join(ROOT, "include", "icicle_snark_hip.h")).read()
    declared = set(re.findall(r"\b((?:icicle|bn254|config_extension|create_config|destroy_config|clone_config)\w*)\s*\(", hdr))
    declared -= {"icicle_snark_hip"}
    assert len(declared) > 70
    assert declared == set(K.DECLARED_SYMBOLS), declared ^ set(K.DECLARED_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), f"missing export {name}"


def test_library_exports_every_prover_symbol(K):
    lib = K.lib()
    hdr = open(os.path.join(ROOT, "include", "groth16_prover.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*|Groth16CacheManager\*)\s+(groth16_\w+)\s*\(", hdr, re.M))
    assert declared == set(K.PROVER_SYMBOLS), declared ^ set(K.PROVER_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), f"missing export {name}"


def test_struct_layouts_match_reference(K):
    # sizes/offsets of the reference's structs on x86-64 (icicle/include/icicle/{msm,ntt,vec_ops,device}.h)
    assert C.sizeof(K.Device) == 68
    assert C.sizeof(K.MSMConfig) == 40 and K.MSMConfig.ext.offset == 32 and K.MSMConfig.are_points_shared_in_batch.offset == 24
    assert C.sizeof(K.NTTConfig) == 64 and K.NTTConfig.coset_gen.offset == 8 and K.NTTConfig.ordering.offset == 48
    assert C.sizeof(K.VecOpsConfig) == 32 and K.VecOpsConfig.batch_size.offset == 12
    assert C.sizeof(K.NTTInitDomainConfig) == 24


def test_unknown_device_is_an_error_not_a_fallback(K):
    d = K.Device(b"CPU", 0)
    assert K.lib().icicle_set_device(C.byref(d)) == 1  # INVALID_DEVICE: there is no CPU backend here
    d = K.Device(b"TPU", 0)
    assert K.lib().icicle_set_device(C.byref(d)) == 1


def test_config_extension(K):
    lib = K.lib()
    lib.create_config_extension.restype = C.c_void_p
    lib.clone_config_extension.restype = C.c_void_p
    lib.config_extension_get_bool.restype = C.c_bool
    e = C.c_void_p(lib.create_config_extension())
    lib.config_extension_set_int(e, b"large_bucket_factor", 7)
    lib.config_extension_set_bool(e, b"is_big_triangle", True)
    e2 = C.c_void_p(lib.clone_config_extension(e))
    assert lib.config_extension_get_int(e2, b"large_bucket_factor") == 7
    assert lib.config_extension_get_bool(e2, b"is_big_triangle") is True
    lib.destroy_config_extension(e)
    lib.destroy_config_extension(e2)


def test_host_field_ffi_golden(K):
    for c in load_golden("field.json")["fr"]:
        a, b = unhex_int(c["a"]), unhex_int(c["b"])
        assert K.fr_op("add", a, b) == unhex_int(c["add"])
        assert K.fr_op("sub", a, b) == unhex_int(c["sub"])
        assert K.fr_op("mul", a, b) == unhex_int(c["mul"])
        assert K.fr_inv(a) == unhex_int(c["inv_a"])


def test_roots_of_unity_golden(K):
    for k, h in enumerate(load_golden("field.json")["roots"]):
        assert K.a2i(K.get_root_of_unity(1 << k)) == unhex_int(h)
    with pytest.raises(K.IcicleError):
        K.get_root_of_unity(1 << 29)


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_host_curve_ffi_golden(K, grp):
    g = load_golden("curve.json")[grp]
    na, npj = DIMS[grp]
    assert np.array_equal(K.ec(grp, "generator"), unhex(g["generator"], npj, 4))
    for c in g["cases"]:
        P, k = unhex(c["p"], npj, 4), unhex_int(c["k"])
        Q = K.ec(grp, "mul_scalar", P, k)
        assert np.array_equal(Q, unhex(c["mul"], npj, 4))
        assert np.array_equal(K.ec(grp, "to_affine", Q), unhex(c["mul_affine"], na, 4))
        assert np.array_equal(K.ec(grp, "ecadd", P, Q), unhex(c["add"], npj, 4))
        assert np.array_equal(K.ec(grp, "ecsub", P, Q), unhex(c["sub"], npj, 4))
        assert np.array_equal(K.ec(grp, "ecadd", Q, Q), unhex(c["dbl"], npj, 4))
        assert K.ec_eq(grp, Q, unhex(c["mul"], npj, 4)) and not K.ec_eq(grp, Q, P) or k == 1
        assert K.ec_is_on_curve(grp, Q)
        A = K.ec(grp, "to_affine", Q)
        assert np.array_equal(K.ec(grp, "from_affine", A)[:na // 2 * 2], A)
    zero = np.zeros((npj, 4), dtype=np.uint64)
    zero[npj // 3, 0] = 1
    assert not K.ec(grp, "to_affine", zero).any()
    assert np.array_equal(K.ec(grp, "from_affine", np.zeros((na, 4), dtype=np.uint64)), zero)
    assert not K.ec_eq(grp, np.zeros((npj, 4), dtype=np.uint64), np.zeros((npj, 4), dtype=np.uint64))


def test_generate_scalars_in_range(K, O):
    out = np.zeros((64, 4), dtype=np.uint64)
    K.lib().bn254_generate_scalars(out.ctypes.data_as(C.c_void_p), 64)
    vals = O.arr_to_ints(out)
    assert all(v < O.R_MOD for v in vals) and len(set(vals)) == 64


def _sections(kind, secs, version=1):
    out = kind + (version).to_bytes(4, "little") + len(secs).to_bytes(4, "little")
    for t, ln, data in secs:
        out += t.to_bytes(4, "little") + ln.to_bytes(8, "little") + data
    return out


def test_malformed_containers_are_format_errors_without_a_device(K):
    """The snarkjs container and the zkey header are validated before the device is touched: a hostile section length
    (pos + len wrapping past 2^64), a truncated table or a foreign magic must come back as a format error — never as
    an out-of-bounds read (the reference, in Rust, panics safely: src/file_wrapper.rs:45-103)."""
    cm = K.CacheManager()
    try:
        wrap = _sections(b"zkey", [(1, 4, (1).to_bytes(4, "little")), (2, (1 << 64) - 8, b"\0" * 64)])
        with pytest.raises(K.ProverError, match="exceeds the file"):
            cm.load("wrapped", wrap)
        with pytest.raises(K.ProverError, match="exceeds the file"):
            cm.load("huge", _sections(b"zkey", [(1, 1 << 40, b"")]))
        with pytest.raises(K.ProverError, match="truncated section table"):
            cm.load("trunc", b"zkey" + (1).to_bytes(4, "little") + (3).to_bytes(4, "little") + b"\1\0\0\0")
        with pytest.raises(K.ProverError, match="Invalid File format"):
            cm.load("magic", _sections(b"wtns", []))
        with pytest.raises(K.ProverError, match="Version not supported"):
            cm.load("version", _sections(b"zkey", [], version=9))
        with pytest.raises(K.ProverError, match="Missing section"):
            cm.load("nosec", _sections(b"zkey", [(1, 4, (1).to_bytes(4, "little"))]))
        with pytest.raises(K.ProverError, match="Protocol not supported"):
            cm.load("proto", _sections(b"zkey", [(1, 4, (2).to_bytes(4, "little"))]))
    finally:
        cm.close()


def test_golden_zkey_with_corrupt_point_section_is_rejected_before_the_device(K):
    import base64
    from conftest import load_golden
    z = bytearray(base64.b64decode(load_golden("groth16.json")["zkey"]))
    cm = K.CacheManager()
    try:
        with pytest.raises(K.ProverError, match="exceeds the file|size mismatch|too short"):
            cm.load("cut", bytes(z[:len(z) - 100]))
    finally:
        cm.close()


def test_device_strings_of_groth16_prove(K, monkeypatch):
    """device argument of groth16_prove / `prove --device` (src/lib.rs:25-31, src/main.rs:46-70): a type with an optional
    device list; more than one device selects the in-process device group.  Parsing needs no GPU."""
    monkeypatch.delenv("ICICLE_SNARK_DEVICES", raising=False)
    assert K.parse_device("HIP") == [0] and K.parse_device("CUDA") == [0]
    assert K.parse_device("HIP:3") == [3]
    assert K.parse_device("HIP:0-7") == list(range(8))
    assert K.parse_device("HIP:0,2,4-6") == [0, 2, 4, 5, 6]
    assert K.parse_device("HIP:0,0,0,0") == [0, 0, 0, 0]          # several shards on one device
    monkeypatch.setenv("ICICLE_SNARK_DEVICES", "1-2")
    assert K.parse_device("HIP") == [1, 2] and K.parse_device("HIP:5") == [5]
    monkeypatch.delenv("ICICLE_SNARK_DEVICES")
    for bad in ("CPU", "CPU:0", "", "HIP:", "HIP:a", "HIP:3-1", "HIP:0,,1", "HIP:0-", "HIP:-1", "HIP:0-99", "ROCM:0"):
        with pytest.raises(K.ProverError):
            K.parse_device(bad)
