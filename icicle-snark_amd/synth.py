"""Deterministic Groth16 key / witness synthesiser (snarkjs `.zkey` / `.wtns` writers).

No circom / snarkjs / ptau download is available offline, so the benchmark inputs
(`/root/reference/benchmark/<N>/circuit.circom:3-23`, input `{"a":"3"}`) are regenerated here from
a known-toxic-waste setup.  The files follow the snarkjs container the reference reads
(src/file_wrapper.rs:45-103, src/zkey.rs:47-85, src/cache.rs:126-181):

  zkey: sec1 protocol=1 | sec2 header (n8q q n8r r n_vars n_public domain α₁ β₁ β₂ γ₂ δ₁ δ₂)
        sec3 IC | sec4 coefficients {m:u32 c:u32 s:u32 value·R² (32 B)} | sec5 A | sec6 B1 | sec7 B2
        sec8 C | sec9 H | sec10 contributions (empty);  all coordinates Montgomery form, LE.
  wtns: sec1 n8 q n_witness | sec2 witness (standard form, LE)

Section-9 basis (SURVEY.md §8 a-H): with g = ω_{2n}, Z(x) = xⁿ − 1 the prover's H scalars are
d_j = (A·B − C)(g·ωʲ) = −2·h(g·ωʲ), hence H_j = [ L_j(τ/g) · Z(τ) / (−2δ) ]₁ with L_j the Lagrange
basis of the size-n domain.

The elliptic-curve work (many fixed-base scalar multiplications) is delegated to a caller-supplied
``fixed_base_mul(group, scalars_u64[n,4]) -> affine_u64[n,2|4,4]`` (standard form): tests pass the CPU
oracle's, bench.py passes the HIP library's.  This module itself is pure Python/numpy.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
# fp_config::rou, primitive 2^28-th root (icicle/include/icicle/fields/snark_fields/bn254_scalar.h:68-69)
ROU_28 = 0x2A3C09F0A58A7E8500E0A7EB8EF62ABC402D111E41112ED49BD61B6E725B19F0
MONT_R = 1 << 256
SEED = 0x1C1C1E


def omega(logn: int) -> int:
    w = ROU_28
    for _ in range(28 - logn):
        w = w * w % R_MOD
    return w


def ints_to_arr(xs) -> np.ndarray:
    return np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in xs), dtype=np.uint64).reshape(-1, 4).copy()


def arr_to_ints(a: np.ndarray):
    raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1, 32)
    return [int.from_bytes(r.tobytes(), "little") for r in raw]


class _Prng:
    """splitmix64-based deterministic stream of field elements (documented seed 0x1c1c1e)."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self) -> int:
        v = 0
        for _ in range(4):
            v = (v << 64) | self.u64()
        return v % R_MOD or 1


@dataclass
class R1CS:
    """Sparse R1CS: lists of (constraint, wire, value) per matrix. Wire 0 is the constant 1,
    wires 1..n_public are the public signals."""
    n_vars: int
    n_public: int
    n_constraints: int
    A: list = field(default_factory=list)
    B: list = field(default_factory=list)
    C: list = field(default_factory=list)


def squaring_chain(N: int, a: int = 3):
    """benchmark/<N>/circuit.circom:3-23 after circom's linear simplification:
    wires [1, c(=b[N-1]), a, b[0..N-2]], constraint j: prev·prev = cur."""
    r = R1CS(n_vars=N + 2, n_public=1, n_constraints=N)
    w = [0] * (N + 2)
    w[0], w[2] = 1, a % R_MOD
    prev_wire, prev_val = 2, w[2]
    for j in range(N):
        cur_wire = 3 + j if j < N - 1 else 1
        cur_val = prev_val * prev_val % R_MOD
        w[cur_wire] = cur_val
        r.A.append((j, prev_wire, 1))
        r.B.append((j, prev_wire, 1))
        r.C.append((j, cur_wire, 1))
        prev_wire, prev_val = cur_wire, cur_val
    return r, w


def random_circuit(n_constraints: int, n_public: int, n_inputs: int, bit_fraction: float = 0.7,
                   a_terms: int = 3, b_terms: int = 2, seed: int = 7):
    """Synthetic stand-in for the real RSA/SHA-heavy circuits (anon_aadhaar / keyless are not
    buildable offline): random sparse A, B rows over earlier wires, every constraint defines one fresh
    wire through C, a `bit_fraction` of constraints are booleanity-style (w·(w−1)=0 ⇒ wire ∈ {0,1})
    so that the witness is dominated by 0/1 values like real circuits."""
    rng = np.random.default_rng(seed)
    pr = _Prng(seed)
    n_vars = 1 + n_public + n_inputs + n_constraints
    r = R1CS(n_vars=n_vars, n_public=n_public, n_constraints=n_constraints)
    w = [0] * n_vars
    w[0] = 1
    first_free = 1 + n_public + n_inputs
    for i in range(1, first_free):
        w[i] = int(rng.integers(0, 2)) if rng.random() < bit_fraction else pr.fr()
    for j in range(n_constraints):
        out = first_free + j
        if rng.random() < bit_fraction:
            # out = x AND y for two earlier bit-ish wires:  x · y = out  (stays in {0,1} when inputs are bits)
            x, y = (int(v) for v in rng.integers(1, out, size=2))
            r.A.append((j, x, 1)); r.B.append((j, y, 1)); r.C.append((j, out, 1))
            w[out] = w[x] * w[y] % R_MOD
        else:
            sa, sb = 0, 0
            for x in rng.integers(0, out, size=a_terms):
                v = int(rng.integers(1, 1 << 16)); r.A.append((j, int(x), v)); sa += v * w[int(x)]
            for x in rng.integers(0, out, size=b_terms):
                v = int(rng.integers(1, 1 << 16)); r.B.append((j, int(x), v)); sb += v * w[int(x)]
            r.C.append((j, out, 1))
            w[out] = (sa % R_MOD) * (sb % R_MOD) % R_MOD
    # public signals: expose the last wires' values by aliasing  pub_i · 1 = wire  is not needed for a
    # stand-in; publics are free inputs already assigned above.
    return r, w


def _batch_inverse(xs):
    n = len(xs)
    pref = [1] * (n + 1)
    for i, x in enumerate(xs):
        pref[i + 1] = pref[i] * x % R_MOD
    inv = pow(pref[n], -1, R_MOD)
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = pref[i] * inv % R_MOD
        inv = inv * xs[i] % R_MOD
    return out


def lagrange_at(n: int, logn: int, y: int):
    """[L_j(y)]_j for the size-n domain: L_j(y) = ωʲ (yⁿ − 1) / (n (y − ωʲ))."""
    w = omega(logn)
    ws = [1] * n
    for j in range(1, n):
        ws[j] = ws[j - 1] * w % R_MOD
    dinv = _batch_inverse([(y - wj) % R_MOD for wj in ws])
    c = (pow(y, n, R_MOD) - 1) * pow(n, -1, R_MOD) % R_MOD
    return [ws[j] * c % R_MOD * dinv[j] % R_MOD for j in range(n)]


def _section(sid: int, payload: bytes) -> bytes:
    return struct.pack("<IQ", sid, len(payload)) + payload


def _mont_fq_bytes(aff: np.ndarray) -> bytes:
    """affine standard-form coordinates → Montgomery form bytes (pure Python; small arrays only)."""
    return b"".join((v * MONT_R % Q_MOD).to_bytes(32, "little") for v in arr_to_ints(aff))


def write_wtns(witness) -> bytes:
    hdr = struct.pack("<I", 32) + R_MOD.to_bytes(32, "little") + struct.pack("<I", len(witness))
    body = b"".join(int(x).to_bytes(32, "little") for x in witness)
    return b"wtns" + struct.pack("<II", 2, 2) + _section(1, hdr) + _section(2, body)


def _toxic(seed):
    pr = _Prng(seed)
    return tuple(pr.fr() for _ in range(5))  # tau, alpha, beta, gamma, delta


def _finish(m, npub, n, a_tau, b_tau, c_s, ic_s, h_s, mcs, vals_r2, toxic, fixed_base_mul, points_to_mont):
    """Common tail of setup(): scalar arrays (numpy (k,4) u64, standard form) → points → zkey bytes."""
    tau, alpha, beta, gamma, delta = toxic
    head = ints_to_arr([alpha, beta, delta])
    g1 = fixed_base_mul("g1", np.concatenate([head, ic_s, a_tau, b_tau, c_s, h_s]))
    g2 = fixed_base_mul("g2", np.concatenate([ints_to_arr([beta, gamma, delta]), b_tau]))
    vk_alpha_1, vk_beta_1, vk_delta_1 = g1[0], g1[1], g1[2]
    o = 3
    IC = g1[o:o + npub + 1]; o += npub + 1
    A = g1[o:o + m]; o += m
    B1 = g1[o:o + m]; o += m
    Cp = g1[o:o + m - npub - 1]; o += m - npub - 1
    H = g1[o:o + n]
    vk_beta_2, vk_gamma_2, vk_delta_2 = g2[0], g2[1], g2[2]
    B2 = g2[3:]
    if points_to_mont is None:
        def points_to_mont(arr):
            flat = arr_to_ints(arr.reshape(-1, 4))
            return ints_to_arr([v * MONT_R % Q_MOD for v in flat]).reshape(arr.shape)
    pm = lambda arr: np.ascontiguousarray(points_to_mont(np.ascontiguousarray(arr))).tobytes()

    hdr = struct.pack("<I", 32) + Q_MOD.to_bytes(32, "little") + struct.pack("<I", 32) + R_MOD.to_bytes(32, "little")
    hdr += struct.pack("<III", m, npub, n)
    hdr += pm(vk_alpha_1) + pm(vk_beta_1) + pm(vk_beta_2) + pm(vk_gamma_2) + pm(vk_delta_1) + pm(vk_delta_2)
    ncoef = mcs.shape[0]
    rec = np.zeros((ncoef, 44), dtype=np.uint8)
    rec[:, 0:12] = np.ascontiguousarray(mcs, dtype=np.uint32).view(np.uint8).reshape(-1, 12)
    rec[:, 12:44] = np.ascontiguousarray(vals_r2, dtype=np.uint64).view(np.uint8).reshape(-1, 32)
    sec4 = struct.pack("<I", ncoef) + rec.tobytes()
    body = _section(1, struct.pack("<I", 1)) + _section(2, hdr) + _section(3, pm(IC)) + _section(4, sec4)
    body += _section(5, pm(A)) + _section(6, pm(B1)) + _section(7, pm(B2)) + _section(8, pm(Cp)) + _section(9, pm(H))
    body += _section(10, struct.pack("<I", 0))
    zkey = b"zkey" + struct.pack("<II", 1, 10) + body
    vk = dict(vk_alpha_1=vk_alpha_1, vk_beta_2=vk_beta_2, vk_gamma_2=vk_gamma_2, vk_delta_2=vk_delta_2,
              IC=[IC[i] for i in range(npub + 1)], n_public=npub)
    return zkey, vk


def setup(r1cs: R1CS, fixed_base_mul, points_to_mont=None, seed: int = SEED):
    """Groth16 setup with toxic waste (τ, α, β, γ, δ) = first five outputs of the fixed-seed PRNG.
    Returns (zkey_bytes, vk dict with standard-form affine numpy points).  Generic (any R1CS), pure Python
    field arithmetic — fine up to ~10^5 constraints; setup_squaring_chain() is the vectorised path for the
    benchmark sizes.

    points_to_mont(arr_u64[..., 4]) -> same shape: optional fast Fq std→Montgomery converter for the big
    point arrays (the HIP library's or the oracle's); defaults to pure Python."""
    toxic = _toxic(seed)
    tau, alpha, beta, gamma, delta = toxic
    m, npub, nc = r1cs.n_vars, r1cs.n_public, r1cs.n_constraints
    n = 1
    while n < nc + npub + 1:
        n <<= 1
    logn = n.bit_length() - 1
    L = lagrange_at(n, logn, tau)
    a_tau, b_tau, c_tau = [0] * m, [0] * m, [0] * m
    coeffs = []  # (m, c, s, value)
    for (j, i, v) in r1cs.A:
        a_tau[i] = (a_tau[i] + v * L[j]) % R_MOD
        coeffs.append((0, j, i, v))
    for (j, i, v) in r1cs.B:
        b_tau[i] = (b_tau[i] + v * L[j]) % R_MOD
        coeffs.append((1, j, i, v))
    for (j, i, v) in r1cs.C:
        c_tau[i] = (c_tau[i] + v * L[j]) % R_MOD
    for s in range(npub + 1):  # snarkjs' extra rows binding the public inputs (A only)
        a_tau[s] = (a_tau[s] + L[nc + s]) % R_MOD
        coeffs.append((0, nc + s, s, 1))
    dinv, ginv = pow(delta, -1, R_MOD), pow(gamma, -1, R_MOD)
    comb = [(beta * a_tau[i] + alpha * b_tau[i] + c_tau[i]) % R_MOD for i in range(m)]
    ic_s = [comb[i] * ginv % R_MOD for i in range(npub + 1)]
    c_s = [comb[i] * dinv % R_MOD for i in range(npub + 1, m)]
    g = omega(logn + 1)
    Lc = lagrange_at(n, logn, tau * pow(g, -1, R_MOD) % R_MOD)
    zt = (pow(tau, n, R_MOD) - 1) * pow((-2 * delta) % R_MOD, -1, R_MOD) % R_MOD
    h_s = [x * zt % R_MOD for x in Lc]
    R2 = MONT_R * MONT_R % R_MOD
    mcs = np.array([(c[0], c[1], c[2]) for c in coeffs], dtype=np.uint32).reshape(-1, 3)
    cache = {}
    for c in coeffs:
        if c[3] not in cache:
            cache[c[3]] = (c[3] * R2 % R_MOD).to_bytes(32, "little")
    vals = np.frombuffer(b"".join(cache[c[3]] for c in coeffs), dtype=np.uint64).reshape(-1, 4)
    _e = lambda xs: ints_to_arr(xs) if len(xs) else np.zeros((0, 4), dtype=np.uint64)
    return _finish(m, npub, n, _e(a_tau), _e(b_tau), _e(c_s), _e(ic_s), _e(h_s), mcs, vals, toxic, fixed_base_mul, points_to_mont)


def vk_to_json(vk: dict) -> str:
    """snarkjs verification_key.json text for the `vk` dict `setup` returns (fields read by the reference:
    src/cache.rs:84-106; projective third coordinates as snarkjs writes them)."""
    import json

    def g1(a):
        x, y = arr_to_ints(a)
        return [str(x), str(y), "1"]

    def g2(a):
        x0, x1, y0, y1 = arr_to_ints(a)
        return [[str(x0), str(x1)], [str(y0), str(y1)], ["1", "0"]]

    return json.dumps({
        "protocol": "groth16", "curve": "bn128", "nPublic": int(vk["n_public"]),
        "vk_alpha_1": g1(vk["vk_alpha_1"]), "vk_beta_2": g2(vk["vk_beta_2"]), "vk_gamma_2": g2(vk["vk_gamma_2"]),
        "vk_delta_2": g2(vk["vk_delta_2"]), "IC": [g1(p) for p in vk["IC"]],
    }, indent=1)


def squaring_chain_witness(N: int, a: int = 3):
    w = [0] * (N + 2)
    w[0], w[2] = 1, a % R_MOD
    v = w[2]
    for j in range(N):
        v = v * v % R_MOD
        w[3 + j if j < N - 1 else 1] = v
    return w


def setup_squaring_chain(N: int, vec, fixed_base_mul, points_to_mont=None, seed: int = SEED):
    """Vectorised setup for benchmark/<N>: byte-identical to setup(squaring_chain(N)[0], …) (tested), with all
    O(n) field work delegated to `vec`, an object offering (numpy (k,4) u64 standard-form arrays):
        vec.mul(a, b), vec.add(a, b)      element-wise Fr
        vec.intt(a)                        inverse NTT, natural order, size len(a) (power of two)
    Uses  [L_j(y)]_j = iNTT([y^k]_k)  — the Lagrange basis at y is the inverse DFT of the powers of y."""
    toxic = _toxic(seed)
    tau, alpha, beta, gamma, delta = toxic
    m, npub, nc = N + 2, 1, N
    n = 1
    while n < nc + npub + 1:
        n <<= 1
    logn = n.bit_length() - 1
    one = ints_to_arr([1])

    def bcast(x, k):
        return np.broadcast_to(ints_to_arr([x]), (k, 4)).copy()

    def powers(y):
        p = one.copy()
        step, k = y % R_MOD, 1
        while k < n:
            p = np.concatenate([p, vec.mul(p, bcast(step, k))])
            step = step * step % R_MOD
            k <<= 1
        return p

    L = vec.intt(powers(tau))
    g = omega(logn + 1)
    Lc = vec.intt(powers(tau * pow(g, -1, R_MOD) % R_MOD))
    z = lambda k: np.zeros((k, 4), dtype=np.uint64)
    a_tau, b_tau, c_tau = z(m), z(m), z(m)
    a_tau[2:2 + N] = L[0:N]
    a_tau[0], a_tau[1] = L[N], L[N + 1]
    b_tau[2:2 + N] = L[0:N]
    c_tau[3:3 + N - 1] = L[0:N - 1]
    c_tau[1] = L[N - 1]
    comb = vec.add(vec.add(vec.mul(a_tau, bcast(beta, m)), vec.mul(b_tau, bcast(alpha, m))), c_tau)
    ic_s = vec.mul(comb[0:2].copy(), bcast(pow(gamma, -1, R_MOD), 2))
    c_s = vec.mul(comb[2:].copy(), bcast(pow(delta, -1, R_MOD), m - 2))
    zt = (pow(tau, n, R_MOD) - 1) * pow((-2 * delta) % R_MOD, -1, R_MOD) % R_MOD
    h_s = vec.mul(Lc, bcast(zt, n))
    j = np.arange(N, dtype=np.uint32)
    mcs = np.concatenate([
        np.stack([np.zeros(N, np.uint32), j, j + 2], axis=1),
        np.stack([np.ones(N, np.uint32), j, j + 2], axis=1),
        np.array([[0, N, 0], [0, N + 1, 1]], dtype=np.uint32)])
    R2 = MONT_R * MONT_R % R_MOD
    vals = bcast(R2, 2 * N + 2)
    return _finish(m, npub, n, a_tau, b_tau, c_s, ic_s, h_s, mcs, vals, toxic, fixed_base_mul, points_to_mont)


# ---------------------------------------------------------------------------------------------------------------------
# Scale-sized stand-ins for BASELINE.json configs 4 (anon_aadhaar) and 5 (Aptos keyless).  The real circuits need circom,
# circomlib and input files that are not in the reference checkout (benchmark/anon_aadhaar has no input.json,
# benchmark/keyless/README.md is a pointer), so a random sparse R1CS of published scale stands in:
# ≥ 70 % of the wires are bits, ~10 % are below 2^64, the rest full width; ≈ 3 non-zeros per row of A and ≈ 1.5 per row
# of B.  Everything produced from it is labelled "synthetic stand-in".
# Where the sizes come from (none of it checkable offline — no circom, no circomlib, no snarkjs; they are estimates and say so):
#  * anon_aadhaar: `component main = AadhaarVerifier(121, 17, 512 * 3)` (benchmark/anon_aadhaar/circuit.circom, last line): an
#    RSA-2048 signature check over 17 limbs of 121 bits on a SHA-256 of up to 1536 bytes = 24 compression blocks; circomlib's
#    SHA-256 is ≈ 29 k constraints per block (≈ 0.7 M), the big-integer exponentiation to 65537 and the QR-data extractor make up the
#    rest: ≈ 1.0 M constraints, domain 2^20.  The reference's own chart agrees: its ≈ 150 ms bar lies between its 800K (≈ 100 ms) and
#    1600K (≈ 180 ms) bars (figures/4090_cache.png).  Public signals: nullifierSeed, signalHash + 7 outputs in the real circuit; 4 here.
#  * keyless: benchmark/keyless/README.md only points at aptos-labs/keyless-zk-proofs; its ≈ 230 ms bar lies between the 1600K and
#    3200K bars of the same chart, and a bit-heavy circuit proves faster per constraint than the dense benchmark chain, so 1.4 M
#    constraints (domain 2^21) is a lower-end estimate of that scale.
#  * sparsity / witness mix: SHA-256 and RSA limb arithmetic are dominated by boolean wires (XOR / AND / MUX of bits, bit
#    decompositions) with a minority of range-limited limbs and full-width products — the five constraint kinds of standin_circuit.
STANDIN_SIZES = {
    # name: (constraints, public signals, free inputs) — domain 2^20 / 2^21
    "aadhaar_standin": (1_000_000, 4, 4096),
    "keyless_standin": (1_400_000, 1, 8192),
}


@dataclass
class SparseR1CS:
    """R1CS as flat numpy arrays (row, wire, small signed coefficient) per matrix — the list-of-tuples R1CS above costs
    ~100 bytes per entry, too much at 10^6 constraints."""
    n_vars: int
    n_public: int
    n_constraints: int
    A: tuple  # (rows int64[k], wires int64[k], coeffs int64[k])
    B: tuple
    C: tuple

    def to_lists(self) -> R1CS:
        r = R1CS(self.n_vars, self.n_public, self.n_constraints)
        for name in "ABC":
            rows, wires, vals = getattr(self, name)
            getattr(r, name).extend((int(j), int(i), int(v) % R_MOD) for j, i, v in zip(rows.tolist(), wires.tolist(), vals.tolist()))
        return r


def standin_circuit(n_constraints: int, n_public: int, n_inputs: int, seed: int = 7):
    """Random satisfiable R1CS with the wire and row statistics of an RSA/SHA-style circuit.  Every constraint defines one
    fresh wire (`out`, through C).  Five constraint kinds (AND alone would drive the density of ones towards zero — the
    XOR / MUX share keeps it near 3/8):
      10 %  AND    x · y = out                       x, y earlier BIT wires              A 1, B 1   out ∈ {0,1}
      25 %  XOR    2x · y = x + y − out              x, y earlier bit wires              A 1, B 1   out ∈ {0,1}
      35 %  MUX    (x − y) · s = out − y             x, y, s earlier bit wires           A 2, B 1   out ∈ {0,1}
      10 %  PACK   (Σ_{k<8} 2^k·b_k) · 1 = out       b_k earlier bit wires               A 8, B 1   out < 2^9: a small value
      20 %  MULADD (Σ_5 a_k x_k) · (Σ_3 b_k y_k) = out   any earlier wires, a, b < 2^16   A 5, B 3   out full width
    Average non-zeros per row: A 2.85, B 1.4.  Returns (SparseR1CS, witness list of Python ints)."""
    rng = np.random.default_rng(seed)
    pr = _Prng(seed)
    first = 1 + n_public + n_inputs
    n_vars = first + n_constraints
    w = [0] * n_vars
    w[0] = 1
    is_bit = np.zeros(n_vars, dtype=bool)
    # free inputs: 3/4 bits, the others full width (public signals are full-width field elements)
    for i in range(1, first):
        if i > n_public and (i & 3):
            w[i] = int(rng.integers(0, 2))
            is_bit[i] = True
        else:
            w[i] = pr.fr()
    kind = rng.choice(5, size=n_constraints, p=[0.10, 0.35, 0.10, 0.20, 0.25])   # AND, MUX, PACK, MULADD, XOR
    is_bit[first:] = (kind < 2) | (kind == 4)
    # bit_rank[i] = number of bit wires among wires [0, i): a random earlier bit wire is bits[floor(u · bit_rank[out])]
    bits = np.flatnonzero(is_bit)
    bit_rank = np.cumsum(is_bit) - is_bit
    assert bit_rank[first] >= 8, "need at least 8 bit inputs"
    u = rng.random((n_constraints, 8))
    any_u = rng.random((n_constraints, 8))
    coef = rng.integers(1, 1 << 16, size=(n_constraints, 8))
    outs = first + np.arange(n_constraints)
    nb = bit_rank[outs]                                       # bit wires available to each constraint
    bit_pick = bits[(u * nb[:, None]).astype(np.int64)]       # 8 candidate bit wires per constraint
    any_pick = (any_u * outs[:, None]).astype(np.int64)       # 8 candidate arbitrary earlier wires (incl. the constant 1)
    Ar, Aw, Av, Br, Bw, Bv, Cr, Cw, Cv = ([] for _ in range(9))
    j_all = np.arange(n_constraints)
    for k in range(5):
        sel = j_all[kind == k]
        if k == 0:      # AND
            Ar.append(sel); Aw.append(bit_pick[sel, 0]); Av.append(np.ones(len(sel), np.int64))
            Br.append(sel); Bw.append(bit_pick[sel, 1]); Bv.append(np.ones(len(sel), np.int64))
            Cr.append(sel); Cw.append(outs[sel]); Cv.append(np.ones(len(sel), np.int64))
        elif k == 1:    # MUX
            Ar += [sel, sel]; Aw += [bit_pick[sel, 0], bit_pick[sel, 1]]; Av += [np.ones(len(sel), np.int64), -np.ones(len(sel), np.int64)]
            Br.append(sel); Bw.append(bit_pick[sel, 2]); Bv.append(np.ones(len(sel), np.int64))
            Cr += [sel, sel]; Cw += [outs[sel], bit_pick[sel, 1]]; Cv += [np.ones(len(sel), np.int64), -np.ones(len(sel), np.int64)]
        elif k == 2:    # PACK
            for t in range(8):
                Ar.append(sel); Aw.append(bit_pick[sel, t]); Av.append(np.full(len(sel), 1 << t, np.int64))
            Br.append(sel); Bw.append(np.zeros(len(sel), np.int64)); Bv.append(np.ones(len(sel), np.int64))
            Cr.append(sel); Cw.append(outs[sel]); Cv.append(np.ones(len(sel), np.int64))
        elif k == 4:    # XOR
            ones = np.ones(len(sel), np.int64)
            Ar.append(sel); Aw.append(bit_pick[sel, 0]); Av.append(2 * ones)
            Br.append(sel); Bw.append(bit_pick[sel, 1]); Bv.append(ones)
            Cr += [sel, sel, sel]; Cw += [bit_pick[sel, 0], bit_pick[sel, 1], outs[sel]]; Cv += [ones, ones, -ones]
        else:           # MULADD
            for t in range(5):
                Ar.append(sel); Aw.append(any_pick[sel, t]); Av.append(coef[sel, t])
            for t in range(3):
                Br.append(sel); Bw.append(any_pick[sel, 5 + t]); Bv.append(coef[sel, 5 + t])
            Cr.append(sel); Cw.append(outs[sel]); Cv.append(np.ones(len(sel), np.int64))
    cat = lambda xs: np.concatenate(xs).astype(np.int64)
    A = (cat(Ar), cat(Aw), cat(Av)); B = (cat(Br), cat(Bw), cat(Bv)); C = (cat(Cr), cat(Cw), cat(Cv))
    # witness: constraints in order (each reads earlier wires only)
    kl = kind.tolist()
    bp = bit_pick.tolist()
    ap = any_pick.tolist()
    cf = coef.tolist()
    for j in range(n_constraints):
        out = first + j
        k = kl[j]
        b = bp[j]
        if k == 0:
            w[out] = w[b[0]] & w[b[1]]
        elif k == 1:
            w[out] = w[b[0]] if w[b[2]] else w[b[1]]
        elif k == 2:
            v = 0
            for t in range(8):
                v += w[b[t]] << t
            w[out] = v
        elif k == 4:
            w[out] = w[b[0]] ^ w[b[1]]
        else:
            a, c = ap[j], cf[j]
            sa = c[0] * w[a[0]] + c[1] * w[a[1]] + c[2] * w[a[2]] + c[3] * w[a[3]] + c[4] * w[a[4]]
            sb = c[5] * w[a[5]] + c[6] * w[a[6]] + c[7] * w[a[7]]
            w[out] = (sa % R_MOD) * (sb % R_MOD) % R_MOD
    return SparseR1CS(n_vars, n_public, n_constraints, A, B, C), w


def check_r1cs(r: SparseR1CS, w, sample: int = 0) -> bool:
    """(A·w) ∘ (B·w) = C·w on all (sample = 0) or on `sample` random constraints — used by the tests of the generator."""
    rows = range(r.n_constraints)
    if sample:
        rows = np.random.default_rng(1).integers(0, r.n_constraints, size=sample).tolist()
    want = set(rows)
    acc = {name: {} for name in "ABC"}
    for name in "ABC":
        jr, iw, vv = getattr(r, name)
        d = acc[name]
        if sample:
            m = np.isin(jr, np.fromiter(want, dtype=np.int64))
            jr, iw, vv = jr[m], iw[m], vv[m]
        for j, i, v in zip(jr.tolist(), iw.tolist(), vv.tolist()):
            d[j] = (d.get(j, 0) + v * w[i]) % R_MOD
    return all(acc["A"].get(j, 0) * acc["B"].get(j, 0) % R_MOD == acc["C"].get(j, 0) for j in want)


def setup_sparse(r: SparseR1CS, vec, fixed_base_mul, points_to_mont=None, seed: int = SEED):
    """setup() for a SparseR1CS at 10^6 constraints: byte-identical to setup(r.to_lists(), …) (tested at small size).  The
    two Lagrange bases come from `vec.intt` (as in setup_squaring_chain), the O(n) scalings from `vec.mul/add`; the sparse
    accumulations a_i(τ) = Σ_j v_ji·L_j(τ) run over Python integers (a few seconds per million non-zeros)."""
    toxic = _toxic(seed)
    tau, alpha, beta, gamma, delta = toxic
    m, npub, nc = r.n_vars, r.n_public, r.n_constraints
    n = 1
    while n < nc + npub + 1:
        n <<= 1
    logn = n.bit_length() - 1
    one = ints_to_arr([1])

    def bcast(x, k):
        return np.broadcast_to(ints_to_arr([x]), (k, 4)).copy()

    def powers(y):
        p = one.copy()
        step, k = y % R_MOD, 1
        while k < n:
            p = np.concatenate([p, vec.mul(p, bcast(step, k))])
            step = step * step % R_MOD
            k <<= 1
        return p

    L = arr_to_ints(vec.intt(powers(tau)))
    g = omega(logn + 1)
    Lc = vec.intt(powers(tau * pow(g, -1, R_MOD) % R_MOD))
    tau_polys = []
    for name in "ABC":
        acc = [0] * m
        rows, wires, vals = getattr(r, name)
        for j, i, v in zip(rows.tolist(), wires.tolist(), vals.tolist()):
            acc[i] += v * L[j]
        if name == "A":
            for s in range(npub + 1):     # snarkjs' extra rows binding the public inputs (A only)
                acc[s] += L[nc + s]
        tau_polys.append(ints_to_arr([x % R_MOD for x in acc]))
    a_tau, b_tau, c_tau = tau_polys
    comb = vec.add(vec.add(vec.mul(a_tau, bcast(beta, m)), vec.mul(b_tau, bcast(alpha, m))), c_tau)
    ic_s = vec.mul(comb[0:npub + 1].copy(), bcast(pow(gamma, -1, R_MOD), npub + 1))
    c_s = vec.mul(comb[npub + 1:].copy(), bcast(pow(delta, -1, R_MOD), m - npub - 1))
    zt = (pow(tau, n, R_MOD) - 1) * pow((-2 * delta) % R_MOD, -1, R_MOD) % R_MOD
    h_s = vec.mul(Lc, bcast(zt, n))
    # coefficient section: A entries, then B entries (order of setup()), then the public-input rows
    pub = np.arange(npub + 1, dtype=np.int64)
    mm = np.concatenate([np.zeros(len(r.A[0]), np.int64), np.ones(len(r.B[0]), np.int64), np.zeros(npub + 1, np.int64)])
    cc = np.concatenate([r.A[0], r.B[0], nc + pub])
    ss = np.concatenate([r.A[1], r.B[1], pub])
    vv = np.concatenate([r.A[2], r.B[2], np.ones(npub + 1, np.int64)])
    mcs = np.stack([mm, cc, ss], axis=1).astype(np.uint32)
    R2 = MONT_R * MONT_R % R_MOD
    uniq, inv = np.unique(vv, return_inverse=True)
    table = ints_to_arr([int(v) % R_MOD * R2 % R_MOD for v in uniq.tolist()])
    vals = table[inv]
    return _finish(m, npub, n, a_tau, b_tau, c_s, ic_s, h_s, mcs, vals, toxic, fixed_base_mul, points_to_mont)
