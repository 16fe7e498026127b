"""The identity behind the multi-GPU residue-class split of the H MSM (csrc/prover/qap.h: qap_coset_fold3), checked with
Python integers against the oracle's NTT: for n = G·m, coset generator g = ω_2n and any rank r < G,
    NTT_n(x_j · g^j)[r + G·k'] = NTT_m(z)[k'],   z[j'] = Σ_{t<G} x[j' + t·m] · ω_2n^((j'+tm) + 2m·(tr mod G) + 2·(j'r mod n)).
CPU only."""
import random

import numpy as np
import pytest


@pytest.mark.parametrize("logn,G", [(6, 2), (6, 4), (7, 8), (5, 32)])
def test_folded_coset_transform_equals_the_residue_class_of_the_full_one(O, logn, G):
    R = O.R_MOD
    n, m = 1 << logn, (1 << logn) // G
    rnd = random.Random(logn * 100 + G)
    x = [rnd.randrange(R) for _ in range(n)]
    w2n = O.fr_omega(logn + 1)
    full = O.arr_to_ints(O.fr_ntt(O.ints_to_arr([v * pow(w2n, j, R) % R for j, v in enumerate(x)]), False, domain_log=logn + 1))
    for r in range(G):
        z = []
        for jp in range(m):
            acc = 0
            for t in range(G):
                e = ((jp + t * m) + 2 * m * ((t * r) % G) + 2 * ((jp * r) % n)) % (2 * n)
                acc = (acc + x[jp + t * m] * pow(w2n, e, R)) % R
            z.append(acc)
        if m > 1:
            got = O.arr_to_ints(O.fr_ntt(O.ints_to_arr(z), False, domain_log=logn + 1))
        else:
            got = z
        assert got == [full[r + G * k] for k in range(m)], (logn, G, r)
