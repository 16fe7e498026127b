"""The scale-sized stand-in generator (BASELINE.json configs 4/5) — CPU checks at small size:
the random R1CS is satisfied by its witness, has the prescribed wire and row statistics, the vectorised setup writes
the same bytes as the generic one, and a proof of it (CPU oracle) passes the pairing check."""
import json

import numpy as np


class OracleVec:
    def __init__(self, O):
        self.mul = lambda a, b: O.fr_vector_mul(np.ascontiguousarray(a), np.ascontiguousarray(b))
        self.add = lambda a, b: O.fr_vector_add(np.ascontiguousarray(a), np.ascontiguousarray(b))
        self.intt = lambda a: O.fr_ntt(np.ascontiguousarray(a), True)


def test_standin_circuit_statistics_and_satisfiability(S):
    r, w = S.standin_circuit(30000, 3, 64, seed=5)
    assert r.n_vars == 1 + 3 + 64 + 30000 and len(w) == r.n_vars and w[0] == 1
    assert S.check_r1cs(r, w)
    kinds = np.array([0 if x in (0, 1) else (1 if x < (1 << 64) else 2) for x in w])
    assert (kinds == 0).mean() >= 0.70                       # bit wires
    assert 0.05 <= (kinds == 1).mean() <= 0.16               # small values
    assert 2.6 <= len(r.A[0]) / r.n_constraints <= 3.1       # ≈ 3 non-zeros per row of A
    assert 1.3 <= len(r.B[0]) / r.n_constraints <= 1.6       # ≈ 1.5 per row of B
    w2 = list(w)
    w2[-1] ^= 1
    assert not S.check_r1cs(r, w2)
    assert S.check_r1cs(r, w, sample=500)


def test_setup_sparse_equals_generic_setup_and_proves(S, O, K):
    r, w = S.standin_circuit(700, 2, 24, seed=9)
    Gaff = {g: O.ec_to_affine(g, O.ec_generator(g)) for g in ("g1", "g2")}
    fbm = lambda g, sc: O.fixed_base_mul(g, Gaff[g], sc)
    to_mont = lambda a: O.fq_convert_montgomery(a, True)
    zk_fast, vk = S.setup_sparse(r, OracleVec(O), fbm, points_to_mont=to_mont)
    zk_ref, _ = S.setup(r.to_lists(), fbm, points_to_mont=to_mont)
    assert zk_fast == zk_ref
    proof, public = O.groth16_prove(zk_fast, S.write_wtns(w), 3, 4)
    assert public == [str(w[1]), str(w[2])]
    assert K.groth16_verify_json(json.dumps(proof), json.dumps(public), S.vk_to_json(vk)) is True
