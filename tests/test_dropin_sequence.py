"""The reference's own call sequence through the drop-in boundary.

`lib/dropin_host` (csrc/tools/dropin_host.cc) restates the reference's Rust host call for call — src/cache.rs:117-256
(cache build with device-side from_mont, domain from points_a.len()), src/proof_helper.rs:31-241 (host gather →
bn254_scalar_convert_montgomery → bn254_vector_mul to a HOST result → serial bn254_add scatter → H2D → bn254_ntt ×2 with
batch 3 → seven vec ops → five bn254_msm / bn254_g2_msm on two streams over interior-pointer slices) and :274-316
(blinding through the host EC FFI) — using ONLY `icicle_*` / `bn254_*` exports.  Its proofs must equal the CPU oracle's
for the same (r, s), and the fused host's (groth16_prove_mem)."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "icicle-snark_amd", "lib", "dropin_host")


def _run(tmp_path, zkey, wtns, r, s, iters=1):
    (tmp_path / "c.zkey").write_bytes(zkey)
    (tmp_path / "w.wtns").write_bytes(wtns)
    cmd = [EXE, str(tmp_path / "c.zkey"), str(tmp_path / "w.wtns"), str(tmp_path / "proof.json"), str(tmp_path / "public.json"),
           "--rs", str(r), str(s), "--iters", str(iters), "--keys-dir", str(tmp_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("proof took:") == iters
    return (tmp_path / "proof.json").read_text(), (tmp_path / "public.json").read_text(), out.stdout


def test_only_abi_symbols_are_used():
    """the driver must not reach into the library beyond the reference's FFI surface"""
    syms = subprocess.run(["nm", "-D", "--undefined-only", EXE], capture_output=True, text=True).stdout.split()
    ours = [s for s in syms if s.startswith(("icicle_", "bn254_", "groth16_", "msm_", "qap_")) or "config_extension" in s]
    assert ours and all(s.startswith(("icicle_", "bn254_")) or s.endswith("config_extension") for s in ours), ours
    assert not any(s.startswith("groth16_") for s in ours)
    allowed_extra = {"icicle_snark_last_error"}            # diagnostics only
    assert {s for s in ours if s.startswith("icicle_snark_")} <= allowed_extra


def test_dropin_sequence_3000_matches_oracle(gpu, O, S, tmp_path):
    K = gpu
    r1, w = S.squaring_chain(3000)
    zkey, vk = S.setup(r1, lambda g, k: K.generator_mul(g, k), points_to_mont=lambda a: O.fq_convert_montgomery(a, True))
    wtns = S.write_wtns(w)
    pj, qj, _ = _run(tmp_path, zkey, wtns, 5, 7, iters=2)   # second iteration: warm cache, key file present
    proof, public = O.groth16_prove(zkey, wtns, 5, 7)
    assert json.loads(pj) == proof and json.loads(qj) == public
    cm = K.CacheManager()
    cm.load("k", zkey)
    fused, fpub, _ = cm.prove_mem("k", wtns, 5, 7)
    assert fused == pj and fpub == qj                          # byte-identical files from both hosts
    cm.close()
    K.release_domain()
    assert K.groth16_verify_json(pj, qj, S.vk_to_json(vk))


def test_dropin_sequence_random_circuit_matches_oracle(gpu, O, S, tmp_path):
    """several public signals, rows with many non-zeros, bit-heavy witness (duplicates in c + m·n exercise the bn254_add path)"""
    K = gpu
    r, w = S.standin_circuit(5000, 3, 40, seed=3)
    zkey, vk = S.setup(r.to_lists(), lambda g, k: K.generator_mul(g, k), points_to_mont=lambda a: O.fq_convert_montgomery(a, True))
    wtns = S.write_wtns(w)
    pj, qj, _ = _run(tmp_path, zkey, wtns, 11, 13)
    proof, public = O.groth16_prove(zkey, wtns, 11, 13)
    assert json.loads(pj) == proof and json.loads(qj) == public
    assert K.groth16_verify_json(pj, qj, S.vk_to_json(vk))


def test_dropin_sequence_100k_matches_oracle(gpu, O, tmp_path):
    K = gpu
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    zkey, wtns = bench.make_inputs(K, S, 100_000)
    pj, qj, log = _run(tmp_path, zkey, wtns, 1, 1, iters=3)
    proof, public = O.groth16_prove(zkey, wtns, 1, 1)
    assert json.loads(pj) == proof and json.loads(qj) == public
    print(log)
