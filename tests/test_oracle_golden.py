"""The CPU oracle against the committed golden vectors (outputs of the reference's own sources,
tests/golden/make_golden.py).  CPU only."""
import base64

import numpy as np
import pytest

from conftest import load_golden, unhex, unhex_int

DIMS = {"g1": (2, 3), "g2": (4, 6)}


def test_field_ops(O):
    g = load_golden("field.json")
    for c in g["fr"]:
        a, b = unhex_int(c["a"]), unhex_int(c["b"])
        assert O.fr_add(a, b) == unhex_int(c["add"])
        assert O.fr_sub(a, b) == unhex_int(c["sub"])
        assert O.fr_mul(a, b) == unhex_int(c["mul"])
        assert O.fr_inv(a) == unhex_int(c["inv_a"])


def test_roots_of_unity(O):
    g = load_golden("field.json")
    for k, h in enumerate(g["roots"]):
        w = unhex_int(h)
        assert O.fr_omega(k) == w
        assert O.get_root_of_unity(1 << k) == w
        # order exactly 2^k
        assert pow(w, 1 << k, O.R_MOD) == 1 and (k == 0 or pow(w, 1 << (k - 1), O.R_MOD) != 1)
    with pytest.raises(ValueError):
        O.fr_omega(29)


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_curve_ops_bit_identical(O, grp):
    g = load_golden("curve.json")[grp]
    na, npj = DIMS[grp]
    assert np.array_equal(O.ec_generator(grp), unhex(g["generator"], npj, 4))
    for c in g["cases"]:
        P = unhex(c["p"], npj, 4)
        k = unhex_int(c["k"])
        Q = O.ec_mul_scalar(grp, P, k)
        assert np.array_equal(Q, unhex(c["mul"], npj, 4))           # same projective representative
        assert np.array_equal(O.ec_to_affine(grp, Q), unhex(c["mul_affine"], na, 4))
        assert np.array_equal(O.ec_add(grp, P, Q), unhex(c["add"], npj, 4))
        assert np.array_equal(O.ec_sub(grp, P, Q), unhex(c["sub"], npj, 4))
        assert np.array_equal(O.ec_add(grp, Q, Q), unhex(c["dbl"], npj, 4))
        assert O.ec_eq(grp, O.ec_dbl(grp, Q), unhex(c["dbl"], npj, 4))
        assert O.ec_is_on_curve(grp, Q)
    assert np.array_equal(O.ec_to_affine(grp, O.ec_zero(grp)), unhex(g["zero_affine"], na, 4))
    assert not O.ec_to_affine(grp, unhex(g["p_minus_p"], npj, 4)).any()


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_point_montgomery_conversion(O, grp):
    g = load_golden("curve.json")[grp]
    na, _ = DIMS[grp]
    pts = unhex(g["mont_in"], 4, na, 4)
    assert np.array_equal(O.fq_convert_montgomery(pts, True), unhex(g["to_mont"], 4, na, 4))
    assert np.array_equal(O.fq_convert_montgomery(pts, False), unhex(g["from_mont"], 4, na, 4))


def test_scalar_montgomery_definition(O):
    rng = np.random.default_rng(1)
    xs = [int.from_bytes(rng.bytes(32), "little") % O.R_MOD for _ in range(16)] + [0, 1, O.R_MOD - 1]
    a = O.ints_to_arr(xs)
    Rm = 1 << 256
    assert O.arr_to_ints(O.fr_convert_montgomery(a, True)) == [x * Rm % O.R_MOD for x in xs]
    assert O.arr_to_ints(O.fr_convert_montgomery(a, False)) == [x * pow(Rm, -1, O.R_MOD) % O.R_MOD for x in xs]


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_msm_golden(O, grp):
    na, _ = DIMS[grp]
    for c in load_golden("msm.json")[grp]:
        n = c["n"]
        sc, bases = unhex(c["scalars"], n, 4), unhex(c["bases"], n, na, 4)
        want = unhex(c["result_affine"], na, 4)
        for cc in (0, 1, 4, 7, 13, 16):
            got = O.ec_to_affine(grp, O.msm(grp, sc, bases, c=cc))
            assert np.array_equal(got, want), (grp, n, c["kind"], cc)
        assert np.array_equal(O.ec_to_affine(grp, O.msm(grp, sc, bases, naive=True)), want)


def test_msm_empty_and_all_zero(O):
    sc = O.ints_to_arr([0, 0, 0])
    bases = O.fixed_base_mul("g1", O.ec_to_affine("g1", O.ec_generator("g1")), O.ints_to_arr([5, 6, 7]))
    assert not O.ec_to_affine("g1", O.msm("g1", sc, bases)).any()
    assert not O.ec_to_affine("g1", O.msm("g1", sc[:0], bases[:0])).any()


def test_ntt_golden(O):
    for c in load_golden("ntt.json"):
        n = c["n"]
        x = unhex(c["x"], n, 4)
        assert np.array_equal(O.fr_ntt(x, False), unhex(c["forward"], n, 4))
        assert np.array_equal(O.fr_ntt(x, True), unhex(c["inverse"], n, 4))
        # a larger domain must not change the result (domain stride semantics)
        assert np.array_equal(O.fr_ntt(x, False, domain_log=12), unhex(c["forward"], n, 4))
        assert np.array_equal(O.fr_ntt(O.fr_ntt(x, False), True), x)


def test_ntt_batch_and_sizes(O):
    rng = np.random.default_rng(3)
    for logn in (0, 1, 5, 10):
        n = 1 << logn
        x = O.ints_to_arr([int.from_bytes(rng.bytes(32), "little") % O.R_MOD for _ in range(3 * n)])
        y = O.fr_ntt(x, False, batch=3)
        for b in range(3):
            assert np.array_equal(y[b * n:(b + 1) * n], O.fr_ntt(x[b * n:(b + 1) * n], False))
        assert np.array_equal(O.fr_ntt(y, True, batch=3), x)
    n = 32
    x = O.ints_to_arr([int.from_bytes(rng.bytes(32), "little") % O.R_MOD for _ in range(n)])
    assert np.array_equal(O.fr_ntt(x, False), O.fr_dft_naive(x, O.fr_omega(5)))


def test_groth16_golden_proof(O):
    g = load_golden("groth16.json")
    zkey, wtns = base64.b64decode(g["zkey"]), base64.b64decode(g["wtns"])
    for c in g["cases"]:
        proof, public = O.groth16_prove(zkey, wtns, unhex_int(c["r"]), unhex_int(c["s"]))
        assert proof == c["proof"] and public == c["public"]
    assert g["cases"][0]["public"] == [str(pow(3, 64, O.R_MOD))]
