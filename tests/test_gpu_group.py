"""The in-process multi-GPU prove (device string "HIP:a-b", csrc/prover/multi.cpp) on ONE MI355X: a device may be named
several times, so "HIP:0,0,…" runs the real G-way code — one host thread per shard, point-range / residue-class shards,
witness slices + device all-gather, the two all-to-alls of the distributed QAP front end enqueued on the shards' streams,
host sum of the partial commitments — with every "device" aliased to GPU 0.  Proofs must equal the single-device prover's
and the CPU oracle's bit for bit (src/lib.rs:33-61 is the entry point; the reference itself is single-device)."""
import importlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(K, N):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    return bench.make_inputs(K, S, N) + (S,)


@pytest.mark.parametrize("N,G", [(100_000, 2), (100_000, 4), (100_000, 8), (30_001, 3), (400_000, 8)])
def test_group_prove_equals_single_device_and_oracle(gpu, O, N, G):
    K = gpu
    zkey, wtns, S = _inputs(K, N)
    cm = K.CacheManager()
    cm.load("full", zkey)
    want, want_pub, _ = cm.prove_mem("full", wtns, 5, 9)
    if N <= 100_000:
        proof, pub = O.groth16_prove(zkey, wtns, 5, 9)
        assert json.loads(want) == proof and json.loads(want_pub) == pub
    cm.load_devices("grp", zkey, [0] * G)
    info = cm.info("grp")
    assert info.shards == G and info.n_vars == cm.info("full").n_vars
    for rep in range(3):                                   # buffers, events and worker threads are reused
        got, pub, tm = cm.prove_mem("grp", wtns, 5, 9)
        assert got == want and pub == want_pub, (N, G, rep)
    # witness already resident on every shard: the distributed stages run again, the upload does not
    got, pub, _ = cm.prove_mem("grp", wtns, 5, 9, resident=True)
    assert got == want
    # groth16_commitments on a group = the SUM of the shards' commitments = what one device computes
    blk, _ = cm.commitments("grp", wtns)
    one, _ = cm.commitments("full", wtns)
    assert cm.assemble("full", wtns, blk, 5, 9)[0] == cm.assemble("full", wtns, one, 5, 9)[0] == want
    # a different witness through the same group (the chain started at 5 instead of 3 does not satisfy the public
    # output of the first, so only equality with the single-device prover is asserted)
    cm.evict("grp")
    assert not cm.contains("grp")
    cm.close()
    K.release_domain()


@pytest.mark.parametrize("mode", ["pull", "memcpy"])
def test_group_exchange_transports(gpu, mode):
    """both peer transports carry the same prove (the third, rccl, needs distinct devices: test_rccl_in_process_world1)"""
    code = r'''
import importlib, json, os, sys
sys.path.insert(0, %r)
K = importlib.import_module("icicle-snark_amd")
S = importlib.import_module("icicle-snark_amd.synth")
bench = importlib.import_module("bench")
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, 60000)
cm = K.CacheManager()
cm.load("full", zkey)
want = cm.prove_mem("full", wtns, 2, 3)[0]
cm.load_devices("g", zkey, [0, 0, 0, 0])
assert cm.prove_mem("g", wtns, 2, 3)[0] == want
d = cm.group_describe("g")
assert d["shards"] == 4 and d["devices"] == [0, 0, 0, 0] and d["distinct_devices"] == 1 and d["rccl_ranks"] == 0, d
assert d["transport"] == os.environ["ICICLE_SNARK_EXCHANGE"] and d["transport_forced_by_env"] is True, d
assert cm.group_describe("full")["shards"] == 0
print("GROUP_OK")
''' % ROOT
    env = dict(os.environ, ICICLE_SNARK_EXCHANGE=mode, ICICLE_SNARK_VERBOSE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert "GROUP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"{mode} exchange" in out.stderr


def test_group_errors_leave_the_group_usable(gpu):
    K = gpu
    zkey, wtns, S = _inputs(K, 20_000)
    cm = K.CacheManager()
    cm.load_devices("g", zkey, [0, 0])
    with pytest.raises(K.ProverError, match="none resident"):
        cm.commitments("g", None)
    bad = bytearray(wtns)
    bad[8] ^= 0xff                                          # section count
    with pytest.raises(K.ProverError):
        cm.prove_mem("g", bytes(bad), 1, 1)
    short = S.write_wtns(S.squaring_chain_witness(19_000))
    with pytest.raises(K.ProverError, match="Invalid witness length"):
        cm.prove_mem("g", short, 1, 1)
    cm.load("full", zkey)
    assert cm.prove_mem("g", wtns, 1, 1)[0] == cm.prove_mem("full", wtns, 1, 1)[0]
    with pytest.raises(K.ProverError, match="does not exist"):
        cm.load_devices("nodev", zkey, [0, 99])
    cm.close()
    K.release_domain()


def test_cli_and_file_entry_with_device_list(gpu, O, tmp_path):
    """groth16_prove(…, device = "HIP:0,0,0,0") and the REPL worker's --device take the group path (src/main.rs:46-70)"""
    K = gpu
    zkey, wtns, S = _inputs(K, 50_000)
    zp, wp = tmp_path / "c.zkey", tmp_path / "w.wtns"
    zp.write_bytes(zkey)
    wp.write_bytes(wtns)
    cm = K.CacheManager()
    cm.prove_files(str(wp), str(zp), str(tmp_path / "p1.json"), str(tmp_path / "q1.json"), "HIP")
    cm.prove_files(str(wp), str(zp), str(tmp_path / "p4.json"), str(tmp_path / "q4.json"), "HIP:0,0,0,0")
    assert cm.info(f"{zp}_HIP:0,0,0,0").shards == 4
    assert (tmp_path / "q1.json").read_text() == (tmp_path / "q4.json").read_text()
    # random r, s: the proofs differ; both must pass the pairing check
    from test_gpu_fullsize import _vk_of
    vk = S.vk_to_json(_vk_of(O, zkey))
    for n in ("1", "4"):
        assert K.groth16_verify_json((tmp_path / f"p{n}.json").read_text(), (tmp_path / f"q{n}.json").read_text(), vk)
    cm.close()
    K.release_domain()
    exe = os.path.join(ROOT, "icicle-snark_amd", "lib", "prove")
    cmd = f"prove --witness {wp} --zkey {zp} --proof {tmp_path / 'pc.json'} --public {tmp_path / 'qc.json'} --device HIP:0,0\nexit\n"
    out = subprocess.run([exe], input=cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, ICICLE_SNARK_QUIET="1"))
    assert out.returncode == 0 and out.stdout.count("COMMAND_COMPLETED") == 2, out.stdout + out.stderr
    assert json.loads((tmp_path / "qc.json").read_text()) == json.loads((tmp_path / "q1.json").read_text())
    assert K.groth16_verify_json((tmp_path / "pc.json").read_text(), (tmp_path / "qc.json").read_text(), vk)
