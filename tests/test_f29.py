"""Checked host build of the lazy radix-2^29 field / XYZZ arithmetic used by the MSM hot loops (csrc/ff29.h,
csrc/ec29.h): compiled here with g++ -DF29_CHECK, which turns every bound stated in those headers (64-bit column
accumulators, limb domination of the borrow-proof constants, value bounds) into a recorded failure, and compares
all results with the 8x32-bit arithmetic of ff.h / ec.h.  No GPU."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from conftest import ROOT

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


@pytest.fixture(scope="module")
def chk():
    out = os.path.join(ROOT, "build", "f29_check.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(ROOT, "tests", "f29_check.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-DF29_CHECK", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "icicle-snark_amd", "csrc"), "-o", out, src], check=True)
    lib = C.CDLL(out)
    lib.f29_last_failure.restype = C.c_char_p
    return lib


def _arr(ints):
    return np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in ints), dtype=np.uint64).reshape(-1, 4).copy()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_field_ops_match_ff_and_respect_bounds(chk):
    rnd = random.Random(29)
    vals = [0, 1, 2, Q - 1, Q - 2, (1 << 253), (1 << 29) - 1, (1 << 232) - 1, Q >> 1]
    vals += [int("1" * 253, 2) % Q, sum(((1 << 29) - 1) << (29 * i) for i in range(9)) % Q]
    vals += [rnd.randrange(Q) for _ in range(4000)]
    rnd.shuffle(vals)
    a = _arr(vals)
    chk.f29_reset()
    rc = chk.f29_check_field(_p(a), len(vals))
    assert rc == 0, chk.f29_last_failure().decode()


def _points(O, grp, scalars):
    gen = O.ec_to_affine(grp, O.ec_generator(grp))
    std = O.fixed_base_mul(grp, gen, _arr(scalars))     # standard-form affine
    return std, O.fq_convert_montgomery(std, True)       # + Montgomery-256


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_madd_chains_match_ec_and_respect_bounds(chk, O, grp):
    rnd = random.Random(31 if grp == "g1" else 37)
    fn = getattr(chk, f"f29_check_{grp}_chain")
    n = 300 if grp == "g1" else 120
    ks = [rnd.randrange(1, O.R_MOD) for _ in range(n)]
    # special sequences: P then P (doubling branch), P then −P (cancellation), restart after the identity, repeated
    ks[10], ks[11] = 7, 7
    ks[20], ks[21] = 9, O.R_MOD - 9
    ks[40] = ks[41] = ks[42] = 5
    std, mont = _points(O, grp, ks)
    signs = np.array([rnd.randrange(2) for _ in range(n)], dtype=np.uint8)
    signs[10] = signs[11] = 0
    signs[20] = signs[21] = 1
    signs[40:43] = 0
    internal = np.empty_like(mont)
    chk.f29_to_internal(_p(mont), _p(internal), mont.size // 4)
    for form, enc in ((0, std), (1, mont), (2, internal)):
        chk.f29_reset()
        rc = fn(_p(mont), _p(signs), n, form, _p(np.ascontiguousarray(enc)))
        assert rc == 0, f"form {form}: {chk.f29_last_failure().decode()}"
    # a chain that starts by cancelling to the identity and a one-element chain
    chk.f29_reset()
    assert fn(_p(mont[20:]), _p(np.array([0, 0, 1], dtype=np.uint8)), 3, 1, _p(mont[20:])) == 0, chk.f29_last_failure().decode()
    # doubling branch: P + P + P from an empty accumulator, in every encoding
    for form, enc in ((0, std), (1, mont), (2, internal)):
        chk.f29_reset()
        assert fn(_p(mont[40:]), _p(np.zeros(3, dtype=np.uint8)), 3, form, _p(np.ascontiguousarray(enc[40:]))) == 0, chk.f29_last_failure().decode()
        chk.f29_reset()
        assert fn(_p(mont[40:]), _p(np.ones(3, dtype=np.uint8)), 3, form, _p(np.ascontiguousarray(enc[40:]))) == 0, chk.f29_last_failure().decode()


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_add_dbl_trees_match_ec_and_respect_bounds(chk, O, grp):
    rnd = random.Random(41)
    n = 120 if grp == "g1" else 50
    _, mont = _points(O, grp, [rnd.randrange(1, O.R_MOD) for _ in range(n)])
    chk.f29_reset()
    rc = getattr(chk, f"f29_check_{grp}_addtree")(_p(mont), n)
    assert rc == 0, chk.f29_last_failure().decode()
