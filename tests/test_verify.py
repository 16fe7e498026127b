"""groth16_verify and bn254_pairing — host code, no GPU.

* the library's pairing against the reference's own outputs (tests/golden/pairing.json, generated from
  oracle/_ref) bit for bit, and against oracle/_ref live on further seeded points when it is built;
* bilinearity e(aP, Q) = e(P, aQ) = e(P,Q)^a-style checks as in the reference's test
  (wrappers/rust/icicle-core/src/pairing/tests.rs: check_pairing_bilinearity);
* groth16_verify accepts the golden proofs, rejects tampered proofs / public inputs, reports malformed input.
"""
import json
import random

import numpy as np
import pytest

from conftest import load_golden, unhex, unhex_int


def test_pairing_matches_reference_golden(K):
    for c in load_golden("pairing.json")["cases"]:
        e = K.pairing(unhex(c["p"], 2, 4), unhex(c["q"], 4, 4))
        assert np.array_equal(e, unhex(c["e"], 12, 4)), (c["a"], c["b"])


def test_target_field_ffi_matches_reference_golden(K):
    g = load_golden("pairing.json")["target_field"]
    a, b = unhex(g["a"], 12, 4), unhex(g["b"], 12, 4)
    for op in ("add", "sub", "mul"):
        assert np.array_equal(K.gt_op(op, a, b), unhex(g[op], 12, 4)), op
    assert np.array_equal(K.gt_op("inv", a), unhex(g["inv"], 12, 4))
    assert np.array_equal(K.gt_op("pow", a, 5), unhex(g["pow5"], 12, 4))
    one = np.zeros((12, 4), dtype=np.uint64)
    K.lib().bn254_pairing_target_field_from_u32(1, K.ptr_of(one))
    assert np.array_equal(K.gt_op("mul", a, K.gt_op("inv", a)), one)
    rnd = np.zeros((3, 12, 4), dtype=np.uint64)
    K.lib().bn254_pairing_target_field_generate_scalars(K.ptr_of(rnd), 3)
    q = int.from_bytes(bytes.fromhex("47fd7cd8168c203c8dca7168916a81975d588181b64550b829a031e1724e6430"), "little")
    vals = [int.from_bytes(r.tobytes(), "little") for r in rnd.reshape(-1, 4)]
    assert len(set(vals)) == 36 and all(v < q for v in vals)


def test_pairing_matches_reference_live(K, R):
    rnd = random.Random(77)
    g1, g2 = R.ec("g1", "generator"), R.ec("g2", "generator")
    for _ in range(4):
        P = R.ec("g1", "to_affine", R.ec("g1", "mul_scalar", g1, rnd.randrange(1 << 254)))
        Q = R.ec("g2", "to_affine", R.ec("g2", "mul_scalar", g2, rnd.randrange(1 << 254)))
        assert np.array_equal(K.pairing(P, Q), R.pairing(P, Q))


def test_pairing_bilinearity_and_identity(K):
    g1, g2 = K.ec("g1", "generator"), K.ec("g2", "generator")
    P, Q = K.ec("g1", "to_affine", g1), K.ec("g2", "to_affine", g2)
    a = 0x1234567890ABCDEF1234567
    aP = K.ec("g1", "to_affine", K.ec("g1", "mul_scalar", g1, a))
    aQ = K.ec("g2", "to_affine", K.ec("g2", "mul_scalar", g2, a))
    e1, e2 = K.pairing(aP, Q), K.pairing(P, aQ)
    assert np.array_equal(e1, e2)
    assert not np.array_equal(e1, K.pairing(P, Q))
    one = np.zeros((12, 4), dtype=np.uint64)
    one[0, 0] = 1
    assert np.array_equal(K.pairing(np.zeros((2, 4), dtype=np.uint64), Q), one)
    assert np.array_equal(K.pairing(P, np.zeros((4, 4), dtype=np.uint64)), one)
    # e(P, Q)·e(−P, Q) = 1 is what groth16_verify relies on: checked through verify below


def _golden_vk_json(S):
    g = load_golden("groth16.json")
    v = g["vk"]
    vk = dict(vk_alpha_1=unhex(v["vk_alpha_1"], 2, 4), vk_beta_2=unhex(v["vk_beta_2"], 4, 4), vk_gamma_2=unhex(v["vk_gamma_2"], 4, 4),
              vk_delta_2=unhex(v["vk_delta_2"], 4, 4), IC=[unhex(p, 2, 4) for p in v["IC"]], n_public=len(v["IC"]) - 1)
    return g, S.vk_to_json(vk)


def test_verify_accepts_golden_and_rejects_tampering(K, S, tmp_path):
    g, vkj = _golden_vk_json(S)
    for c in g["cases"]:
        pj, qj = json.dumps(c["proof"]), json.dumps(c["public"])
        assert K.groth16_verify_json(pj, qj, vkj) is True
        bad_pub = [str(int(c["public"][0]) + 1)] + c["public"][1:]
        assert K.groth16_verify_json(pj, json.dumps(bad_pub), vkj) is False
        bad = json.loads(pj)
        bad["pi_a"], bad["pi_c"] = [bad["pi_c"][0], bad["pi_c"][1], "1"], [bad["pi_a"][0], bad["pi_a"][1], "1"]
        assert K.groth16_verify_json(json.dumps(bad), qj, vkj) is False
    # a proof for (r, s) verified against the other case's public signals is still fine (same statement)
    assert K.groth16_verify_json(json.dumps(g["cases"][0]["proof"]), json.dumps(g["cases"][1]["public"]), vkj) is True
    # files in, as the reference's API takes them
    c = g["cases"][0]
    (tmp_path / "proof.json").write_text(json.dumps(c["proof"]))
    (tmp_path / "public.json").write_text(json.dumps(c["public"]))
    (tmp_path / "vk.json").write_text(vkj)
    K.groth16_verify(str(tmp_path / "proof.json"), str(tmp_path / "public.json"), str(tmp_path / "vk.json"))
    (tmp_path / "public.json").write_text(json.dumps(["5"]))
    with pytest.raises(K.ProverError, match="Verification failed"):
        K.groth16_verify(str(tmp_path / "proof.json"), str(tmp_path / "public.json"), str(tmp_path / "vk.json"))
    with pytest.raises(K.ProverError, match="cannot read"):
        K.groth16_verify(str(tmp_path / "nope.json"), str(tmp_path / "public.json"), str(tmp_path / "vk.json"))


def test_verify_reports_malformed_input(K, S):
    g, vkj = _golden_vk_json(S)
    c = g["cases"][0]
    pj, qj = json.dumps(c["proof"]), json.dumps(c["public"])
    with pytest.raises(K.ProverError, match="malformed"):
        K.groth16_verify_json(pj[:-5], qj, vkj)
    with pytest.raises(K.ProverError, match="bad point"):
        K.groth16_verify_json(json.dumps({"pi_a": ["x", "1"], "pi_b": c["proof"]["pi_b"], "pi_c": c["proof"]["pi_c"]}), qj, vkj)
    with pytest.raises(K.ProverError, match="length mismatch"):
        K.groth16_verify_json(pj, "[]", vkj)


def test_verify_agrees_with_reference_on_random_circuit(K, S, O, R):
    """a second statement (random R1CS with 3 public inputs) proved by the oracle: the library and the reference's
    pairing check must agree on accept and on reject."""
    Gaff = {g: O.ec_to_affine(g, O.ec_generator(g)) for g in ("g1", "g2")}
    r1, w = S.random_circuit(40, 3, 5)
    zkey, vk = S.setup(r1, lambda g, sc: O.fixed_base_mul(g, Gaff[g], sc))
    proof, public = O.groth16_prove(zkey, S.write_wtns(w), 11, 12)
    vkj = S.vk_to_json(vk)
    assert R.groth16_verify(proof, public, vk)
    assert K.groth16_verify_json(json.dumps(proof), json.dumps(public), vkj) is True
    public2 = list(public)
    public2[2] = str(int(public2[2]) ^ 1)
    assert not R.groth16_verify(proof, public2, vk)
    assert K.groth16_verify_json(json.dumps(proof), json.dumps(public2), vkj) is False


def test_cli_verify(K, S, tmp_path):
    """`verify --proof … --public … --vk …` through the REPL worker (src/main.rs:83-116,168-178)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "icicle-snark_amd", "lib", "prove")
    g, vkj = _golden_vk_json(S)
    c = g["cases"][1]
    (tmp_path / "proof.json").write_text(json.dumps(c["proof"]))
    (tmp_path / "public.json").write_text(json.dumps(c["public"]))
    (tmp_path / "bad.json").write_text(json.dumps(["1"]))
    (tmp_path / "verification_key.json").write_text(vkj)
    cmds = (f"verify --proof {tmp_path}/proof.json --public {tmp_path}/public.json --vk {tmp_path}/verification_key.json\n"
            f"verify --system groth16 --proof {tmp_path}/proof.json --public {tmp_path}/bad.json --vk {tmp_path}/verification_key.json\n"
            "exit\n")
    out = subprocess.run([exe], input=cmds, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    lines = [ln.replace("> ", "") for ln in out.stdout.splitlines()]
    assert lines[:4] == ["VERIFY_OK", "COMMAND_COMPLETED", "VERIFY_FAILED", "COMMAND_COMPLETED"], out.stdout
    assert "Verification failed" in out.stderr
