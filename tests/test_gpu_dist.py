"""The distributed QAP front end on ONE MI355X: G ∈ {2, 4, 8} shard caches in one process play the ranks, the two
all-to-alls are emulated by copying the chunks between the shards' device buffers through the host, and the sum of the
partial commitments must assemble into the same proof as the single-GPU prover (which the oracle pins elsewhere)."""
import importlib
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exchange(K, bufs, rows, rb, cb):
    """bufs[r] = (send ptr, recv ptr) of rank r: chunk (q, p) of r's send buffer → offset (q, r) of p's receive buffer"""
    G = len(bufs)
    sends = [K.raw_to_host(s, rows * rb) for s, _ in bufs]
    for p in range(G):
        recv = bytearray(rows * rb)
        for q in range(rows):
            for r in range(G):
                recv[q * rb + r * cb:q * rb + (r + 1) * cb] = sends[r][q * rb + p * cb:q * rb + (p + 1) * cb]
        K.raw_to_device(bufs[p][1], bytes(recv))


@pytest.mark.parametrize("N,G", [(100_000, 2), (100_000, 4), (100_000, 8), (400_000, 8)])
def test_distributed_front_end_equals_single_gpu(gpu, O, N, G):
    K = gpu
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    zkey, wtns = bench.make_inputs(K, S, N)
    cm = K.CacheManager()
    cm.load("full", zkey)
    want, public, _ = cm.prove_mem("full", wtns, 5, 9)
    if N <= 100_000:
        proof, pub = O.groth16_prove(zkey, wtns, 5, 9)
        assert json.loads(want) == proof and json.loads(public) == pub
    keys = [f"s{r}" for r in range(G)]
    for r, k in enumerate(keys):
        cm.load(k, zkey, shard_rank=r, shard_count=G)
        assert cm.dist_supported(k)
    for rep in range(2):                                      # twice: the buffers and tables are reused
        st = [cm.dist_stage1(k, wtns) for k in keys]
        rows, rb, cb = st[0][2:]
        _exchange(K, [(s[0], s[1]) for s in st], rows, rb, cb)
        st2 = [cm.dist_stage2(k) for k in keys]
        _exchange(K, st2, rows, rb, cb)
        for k in keys:
            cm.dist_exchange_done(k)
        blocks = b"".join(cm.commitments(k, None)[0] for k in keys)
        got, _ = cm.assemble("full", wtns, K.sum_commitments(blocks, G), 5, 9)
        assert got == want, (N, G, rep)
    # the replicated path on the same shard caches still works afterwards (and gives the same commitments)
    blocks = b"".join(cm.commitments(k, wtns)[0] for k in keys)
    got, _ = cm.assemble("full", wtns, K.sum_commitments(blocks, G), 5, 9)
    assert got == want
    assert not cm.dist_supported("full")
    with pytest.raises(K.ProverError, match="not a strided shard"):
        cm.dist_stage1("full", wtns)
    cm.close()
    K.release_domain()


@pytest.mark.parametrize("N,G", [(100_000, 4), (30_001, 3)])
def test_witness_slices_and_all_gather_equal_full_uploads(gpu, O, N, G):
    """multi-GPU witness distribution on one device: every shard cache uploads only its 1/G of the witness, the in-place
    all-gather is emulated by copying the slices between the shards' device buffers, and the proof is the single-GPU one —
    with the distributed front end (power-of-two G) and with the replicated one (G = 3, n_vars not a multiple of G)"""
    K = gpu
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    zkey, wtns = bench.make_inputs(K, S, N)
    cm = K.CacheManager()
    cm.load("full", zkey)
    want, _, _ = cm.prove_mem("full", wtns, 5, 9)
    keys = [f"s{r}" for r in range(G)]
    for r, k in enumerate(keys):
        cm.load(k, zkey, shard_rank=r, shard_count=G)
    n_vars = cm.info("full").n_vars
    for rep in range(2):
        up = [cm.upload_witness_slice(k, wtns) for k in keys]
        sb = up[0][1]
        assert all(u[1] == sb for u in up) and sb * G >= n_vars * 32 and (sb // 32 - 1) * G < n_vars
        # before witness_ready the shard refuses to run on a half-filled buffer
        with pytest.raises(K.ProverError, match="none resident"):
            cm.commitments(keys[0], None)
        slices = [K.raw_to_host(up[r][0] + r * sb, sb) for r in range(G)]
        for r in range(G):
            K.raw_to_device(up[r][0], b"".join(slices))
            cm.witness_ready(keys[r])
        if cm.dist_supported(keys[0]):
            st = [cm.dist_stage1(k, None) for k in keys]
            rows, rb, cb = st[0][2:]
            _exchange(K, [(s[0], s[1]) for s in st], rows, rb, cb)
            _exchange(K, [cm.dist_stage2(k) for k in keys], rows, rb, cb)
            for k in keys:
                cm.dist_exchange_done(k)
        blocks = b"".join(cm.commitments(k, None)[0] for k in keys)
        got, _ = cm.assemble("full", wtns, K.sum_commitments(blocks, G), 5, 9)
        assert got == want, (N, G, rep)
    cm.close()
