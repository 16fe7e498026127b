"""The CPU oracle against oracle/_ref — the reference's own sources compiled where they lie — on fresh
random inputs.  Skipped where the reference tree (hence the build) is absent; the committed golden
vectors (test_oracle_golden.py) carry the same pin everywhere else.  CPU only."""
import random

import numpy as np
import pytest


def test_field_random(O, R):
    rnd = random.Random(7)
    for _ in range(300):
        a, b = rnd.randrange(O.R_MOD), rnd.randrange(O.R_MOD)
        assert O.fr_mul(a, b) == R.fr_bin("mul", a, b) == a * b % O.R_MOD
        assert O.fr_add(a, b) == R.fr_bin("add", a, b)
        assert O.fr_sub(a, b) == R.fr_bin("sub", a, b)
    for a in (0, 1, 2, O.R_MOD - 1, rnd.randrange(O.R_MOD)):
        assert O.fr_inv(a) == R.fr_inv(a)


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_curve_random(O, R, grp):
    rnd = random.Random(11)
    P = O.ec_generator(grp)
    assert np.array_equal(P, R.ec(grp, "generator"))
    for _ in range(6):
        k = rnd.randrange(O.R_MOD)
        Q1, Q2 = O.ec_mul_scalar(grp, P, k), R.ec(grp, "mul_scalar", P, k)
        assert np.array_equal(Q1, Q2)
        assert np.array_equal(O.ec_add(grp, Q1, P), R.ec(grp, "ecadd", Q2, P))
        assert np.array_equal(O.ec_sub(grp, Q1, P), R.ec(grp, "ecsub", Q2, P))
        A = O.ec_to_affine(grp, Q1)
        assert np.array_equal(A, R.ec(grp, "to_affine", Q2))
        assert R.ec_eq(grp, O.ec_add_mixed(grp, Q1, A), R.ec(grp, "ecadd", Q2, Q2))
        assert R.ec_is_on_curve(grp, Q1) and O.ec_is_on_curve(grp, Q2)
        P = O.ec_add(grp, Q1, P)


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_msm_vs_reference_primitives(O, R, grp):
    rnd = random.Random(13)
    G = O.ec_to_affine(grp, O.ec_generator(grp))
    n = 40
    bases = O.fixed_base_mul(grp, G, O.ints_to_arr([rnd.randrange(O.R_MOD) for _ in range(n)]))
    bases[3] = 0
    sc = O.ints_to_arr([rnd.choice([0, 1, rnd.randrange(O.R_MOD)]) for _ in range(n)])
    want = R.ec(grp, "to_affine", R.msm_naive(grp, sc, bases))
    assert np.array_equal(O.ec_to_affine(grp, O.msm(grp, sc, bases)), want)


def test_ntt_vs_reference_primitives(O, R):
    rnd = random.Random(17)
    n = 32
    x = O.ints_to_arr([rnd.randrange(O.R_MOD) for _ in range(n)])
    assert np.array_equal(O.fr_ntt(x, False), R.dft_naive(x, R.get_root_of_unity(n)))


def test_proof_accepted_by_reference_pairing(O, R, S):
    G = {g: O.ec_to_affine(g, O.ec_generator(g)) for g in ("g1", "g2")}
    fbm = lambda g, sc: O.fixed_base_mul(g, G[g], sc)
    r1, w = S.squaring_chain(20)
    zkey, vk = S.setup(r1, fbm, points_to_mont=lambda a: O.fq_convert_montgomery(a, True))
    proof, public = O.groth16_prove(zkey, S.write_wtns(w), 1234567, 7654321)
    assert R.groth16_verify(proof, public, vk)
    bad = dict(proof)
    bad["pi_c"] = proof["pi_a"]
    assert not R.groth16_verify(bad, public, vk)
    r1, w = S.random_circuit(60, 2, 6)
    zkey, vk = S.setup(r1, fbm)
    proof, public = O.groth16_prove(zkey, S.write_wtns(w), 3, 5)
    assert R.groth16_verify(proof, public, vk)
