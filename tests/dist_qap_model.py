"""Distributed QAP front end for a power-of-two number of GPUs — the index algebra of the HIP path (csrc/prover/qap.hip:
qap_dist_*; csrc/prover/prover.cpp: shard_dist_stage1/2) restated on the CPU.  Test infrastructure (tests/test_dist_qap.py).

n = G·m.  The inverse transform of construct_r1cs (src/proof_helper.rs:116), the coset multiplication (:121-141) and the
forward transform (:145) are computed WITHOUT any rank holding a full row:

  j = j1 + G·j2 (input index of the inverse transform),  k = k1·m + k2 (coefficient index),  i = i1 + G·i2 (evaluation index)

  stage 1 (rank r = j1)   Y_r[k2]   = n⁻¹ · ω_n^{−r·k2} · Σ_{j2} x[r + G·j2] · ω_m^{−j2·k2}          size-m inverse DFT of the rank's rows
                                                                                                      (rows c ≡ r mod G of the spmv ONLY)
  exchange 1              rank r sends Y_r[k2 ∈ block b] to rank b,  block b = [b·m/G, (b+1)·m/G)
  stage 2 (rank b)        a[k1·m + k2] = Σ_{j1} ω_G^{−j1·k1} · Y_{j1}[k2]                              size-G inverse DFT across the sources
                          a'[k]        = a[k] · g^k,   g = ω_2n                                       coset keys
                          Z_{i1}[k2]   = ω_n^{k2·i1} · Σ_{k1} ω_G^{k1·i1} · a'[k1·m + k2]              size-G forward DFT
  exchange 2              rank b sends Z_{i1}[k2 ∈ block b] to rank i1
  stage 3 (rank r = i1)   E[r + G·i2]  = Σ_{k2} Z_r[k2] · ω_m^{k2·i2}                                  size-m forward DFT

so every rank ends with the coset evaluations at i ≡ r (mod G) — exactly the scalars of its residue-class shard of the H
bases (Shard.stride) — after two all-to-alls of 3·(n/G)·32·(G−1)/G bytes each and 1/G of the transform work.
"""
from __future__ import annotations

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def exchange_offsets(stage: int, row: int, peer: int, G: int, m: int):
    """(send offset, receive offset) in elements of the chunk (mb = m // G elements) a rank exchanges with `peer` for
    `row` (0..2) in exchange `stage` (1 or 2).  Buffers: Y / recv2 are [row][m]; recv1 / send2 are [row][peer][mb]."""
    mb = m // G
    flat = row * m + peer * mb            # == (row·G + peer)·mb: both layouts address the same offsets
    return (flat, flat)


def stage1(x_rows, r, G, n, omega_n, ntt_inverse):
    """x_rows: three lists of the m elements x[r + G·j2]; ntt_inverse(list) = size-m inverse DFT INCLUDING 1/m.
    Returns Y_r as three lists of m."""
    m = n // G
    ginv = pow(G, -1, R_MOD)
    out = []
    for x in x_rows:
        y = ntt_inverse(x)
        out.append([y[k2] * ginv % R_MOD * pow(omega_n, (-r * k2) % n, R_MOD) % R_MOD for k2 in range(m)])
    return out


def stage2(recv_rows, b, G, n, omega_n, omega_2n):
    """recv_rows[row][j1] = the mb elements Y_{j1}[k2], k2 in block b.  Returns send_rows[row][i1] = Z_{i1}[k2 in block b]."""
    m = n // G
    mb = m // G
    wG = pow(omega_n, m, R_MOD)
    out = []
    for rows in recv_rows:
        z = [[0] * mb for _ in range(G)]
        for t in range(mb):
            k2 = b * mb + t
            a = [sum(rows[j1][t] * pow(wG, (-j1 * k1) % G, R_MOD) for j1 in range(G)) % R_MOD for k1 in range(G)]
            ap = [a[k1] * pow(omega_2n, k1 * m + k2, R_MOD) % R_MOD for k1 in range(G)]
            for i1 in range(G):
                s = sum(ap[k1] * pow(wG, (k1 * i1) % G, R_MOD) for k1 in range(G)) % R_MOD
                z[i1][t] = s * pow(omega_n, (k2 * i1) % n, R_MOD) % R_MOD
        out.append(z)
    return out


def stage3(z_rows, ntt_forward):
    """z_rows: three lists of m (Z_r[k2]); returns d[i2] = E_A·E_B − E_C' for i = r + G·i2 (rows ordered [B | A | C'])."""
    eb, ea, ec = (ntt_forward(z) for z in z_rows)
    return [(a * b - c) % R_MOD for a, b, c in zip(ea, eb, ec)]
