"""Parity at BASELINE.json's full sizes (needs an MI355X), through size-independent properties where the oracle
would take minutes: MSM linearity and sharding additivity at L = 2^21, NTT round trip and linearity at 3 × 2^21,
and a complete benchmark/1600k prove checked (a) for determinism under fixed (r, s), (b) against the expected
public signal 3^(2^N) and (c) by the reference pairing check when oracle/_ref is present."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_fr(rng, n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 61) - 1)
    return a



def _vk_of(O, zkey):
    """verification key (standard-form affine numpy points) read back from the zkey header and section 3"""
    z = O.parse_zkey(zkey)
    conv = lambda a: O.fq_convert_montgomery(a, False)
    sec = O.read_sections(zkey, b"zkey")
    ic = np.frombuffer(O._section(zkey, sec, 3), dtype=np.uint64).reshape(-1, 2, 4)
    return dict(vk_alpha_1=conv(z["vk_alpha_1"]), vk_beta_2=conv(z["vk_beta_2"]), vk_gamma_2=conv(z["vk_gamma_2"]),
                vk_delta_2=conv(z["vk_delta_2"]), IC=[conv(p) for p in ic], n_public=len(ic) - 1)


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_msm_linearity_and_additivity_full_size(gpu, O, grp):
    K = gpu
    n = 1 << 21 if grp == "g1" else 1 << 19
    rng = np.random.default_rng(42)
    s1, s2 = rand_fr(rng, n), rand_fr(rng, n)
    pts = K.generator_mul(grp, rand_fr(rng, n))
    ssum = K.add_scalars(s1, s2)
    d_p = K.DeviceVec.from_host(pts)
    m = lambda sc, lo=0, hi=n: K.msm(grp, K.DeviceVec.from_host(sc[lo:hi]), d_p.slice(lo * pts[0].nbytes, (hi - lo) * pts[0].nbytes), size=hi - lo)
    r1, r2, r12 = m(s1), m(s2), m(ssum)
    assert K.ec_eq(grp, K.ec(grp, "ecadd", r1, r2), r12)                       # MSM(s1) + MSM(s2) = MSM(s1 + s2)
    parts = [m(s1, lo, hi) for lo, hi in ((0, n // 3), (n // 3, n // 2), (n // 2, n))]
    acc = parts[0]
    for p in parts[1:]:
        acc = K.ec(grp, "ecadd", acc, p)
    assert K.ec_eq(grp, acc, r1)                                                 # point-range shards add up
    # Σ sᵢ·(kᵢ·G) = (Σ sᵢ kᵢ)·G on a slice the oracle can finish in seconds
    k = 4096
    want = O.ec_to_affine(grp, O.msm(grp, s1[:k], pts[:k]))
    assert np.array_equal(K.ec(grp, "to_affine", m(s1, 0, k)), want)


@pytest.mark.parametrize("grp,logn", [("g1", 20), ("g2", 18)])
def test_classic_msm_equals_oracle_at_size(gpu, O, grp, logn):
    """bn254_msm / bn254_g2_msm (classic layout: the path the reference's host reaches) against the CPU oracle's Pippenger at
    2^20 (G1) / 2^18 (G2) random scalars — distinct random bases, two identities, the extremal scalars 0, 1, r − 1 —
    on device-resident inputs, standard and Montgomery form"""
    K = gpu
    O.calibrate_threads()
    n = 1 << logn
    rng = np.random.default_rng(1000 + logn)
    sc = rand_fr(rng, n)
    sc[0] = 0
    sc[1] = 0; sc[1, 0] = 1
    sc[2] = np.frombuffer((O.R_MOD - 1).to_bytes(32, "little"), dtype=np.uint64)
    pts = K.generator_mul(grp, rand_fr(rng, n))
    pts[7] = 0
    pts[n - 3] = 0
    want = O.ec_to_affine(grp, O.msm(grp, sc, pts))
    d_s, d_p = K.DeviceVec.from_host(sc), K.DeviceVec.from_host(pts)
    assert np.array_equal(K.ec(grp, "to_affine", K.msm(grp, d_s, d_p, size=n)), want)
    assert np.array_equal(K.ec(grp, "to_affine", K.msm(grp, d_s, d_p, size=n, c=13)), want)
    K.scalar_convert_montgomery(d_s, True)
    K.affine_convert_montgomery(grp, d_p, True)
    assert np.array_equal(K.ec(grp, "to_affine", K.msm(grp, d_s, d_p, size=n, scalars_mont=True, points_mont=True)), want)
    d_s.free(); d_p.free()


def test_batched_ntt_3x2p21_equals_oracle(gpu, O):
    """the prover's transform configuration at benchmark/1600k size — 3 rows of 2^21, in place on the device, domain 2^22 —
    against the oracle's radix-2 transform, both directions"""
    K = gpu
    O.calibrate_threads()
    n = 1 << 21
    K.release_domain()
    K.initialize_domain(K.get_root_of_unity(2 * n))
    rng = np.random.default_rng(21)
    x = rand_fr(rng, 3 * n)
    d = K.DeviceVec.from_host(x)
    K.ntt(d, True, batch_size=3)
    assert np.array_equal(d.to_host(x.shape), O.fr_ntt(x, True, batch=3, domain_log=22))
    d.copy_from_host(x)
    K.ntt(d, False, batch_size=3)
    assert np.array_equal(d.to_host(x.shape), O.fr_ntt(x, False, batch=3, domain_log=22))
    d.free()
    K.release_domain()


def test_ntt_round_trip_and_linearity_full_size(gpu):
    K = gpu
    n = 1 << 21
    K.release_domain()
    K.initialize_domain(K.get_root_of_unity(2 * n))
    rng = np.random.default_rng(7)
    x, y = rand_fr(rng, 3 * n), rand_fr(rng, 3 * n)
    d = K.DeviceVec.from_host(x)
    K.ntt(d, False, batch_size=3)
    fx = d.to_host(x.shape)
    K.ntt(d, True, batch_size=3)
    assert np.array_equal(d.to_host(x.shape), x)                                 # iNTT(NTT(x)) = x
    fy = K.ntt(y, False, batch_size=3)
    fxy = K.ntt(K.add_scalars(x, y), False, batch_size=3)
    assert np.array_equal(fxy, K.add_scalars(fx, fy))                            # NTT(x + y) = NTT(x) + NTT(y)
    # a 2^21 transform inside the 2^22 domain equals the same transform in its own domain
    K.release_domain()
    K.initialize_domain(K.get_root_of_unity(n))
    assert np.array_equal(K.ntt(x[:n].copy(), False), fx[:n])
    d.free()
    K.release_domain()


def test_benchmark_1600k_prove(gpu, O):
    K = gpu
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    N = 1_600_000
    zkey, wtns = bench.make_inputs(K, S, N)
    cm = K.CacheManager()
    cm.load("full", zkey)
    info = cm.info("full")
    assert (info.n_vars, info.n_public, info.domain_size, info.n_coef) == (N + 2, 1, 1 << 21, 2 * N + 2)
    p1, q1, _ = cm.prove_mem("full", wtns, 1, 1)              # `no-randomness` semantics
    p2, q2, _ = cm.prove_mem("full", wtns, 1, 1)
    assert p1 == p2 and q1 == q2                                # deterministic (bucket order inside atomics is not)
    assert json.loads(q1) == [str(pow(3, 1 << N, S.R_MOD))]
    vk = _vk_of(O, zkey)                                         # vk from the synthesised key itself
    vkj = S.vk_to_json(vk)
    p3, q3, _ = cm.prove_mem("full", wtns)                      # random blinding
    assert p3 != p1
    assert K.groth16_verify_json(p1, q1, vkj) and K.groth16_verify_json(p3, q3, vkj)   # the library's own pairing check
    import ref as R
    if R.available():                                            # … and the reference's
        assert R.groth16_verify(json.loads(p1), json.loads(q1), vk)
        assert R.groth16_verify(json.loads(p3), json.loads(q3), vk)
    cm.close()
    K.release_domain()


def test_bit_heavy_witness_100k_matches_oracle(gpu, O):
    """witness shape of the RSA/SHA-style circuits (70 % of the wires in {0,1}, 10 % below 2^64) at 100 k constraints:
    fixed-base tables with buckets holding a third of all entries, large-bucket path at scale, range and residue-class
    shards.  The vector does not satisfy the circuit — prover and oracle compute the same algebra on it all the same,
    so the two proofs must be identical."""
    K = gpu
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    N = 100_000
    zkey, wtns = bench.make_inputs(K, S, N)
    rng = np.random.default_rng(3)
    w = np.frombuffer(wtns, dtype=np.uint8).copy()
    body = w[len(w) - 32 * (N + 2):].view(np.uint64).reshape(-1, 4)
    kind = rng.random(N + 2)
    bits = kind < 0.7
    body[bits] = 0
    body[bits, 0] = rng.integers(0, 2, size=int(bits.sum()), dtype=np.uint64)
    small = (kind >= 0.7) & (kind < 0.8)
    body[small, 1:] = 0
    body[0] = 0
    body[0, 0] = 1
    skewed = w.tobytes()
    cm = K.CacheManager()
    cm.load("k", zkey)
    pj, qj, _ = cm.prove_mem("k", skewed, 5, 7)
    proof, public = O.groth16_prove(zkey, skewed, 5, 7)
    assert json.loads(pj) == proof and json.loads(qj) == public
    for count in (3, 4):
        blocks = b""
        for rank in range(count):
            cm.load(f"s{count}{rank}", zkey, shard_rank=rank, shard_count=count)
            blk, _ = cm.commitments(f"s{count}{rank}", skewed)
            blocks += blk
            cm.evict(f"s{count}{rank}")
        got, _ = cm.assemble("k", skewed, K.sum_commitments(blocks, count), 5, 7)
        assert got == pj, count
    cm.close()
    K.release_domain()


def test_repeated_prove_cache_loop(gpu, O, S):
    """config 5 pattern (examples/rust/src/main.rs:3-4,20-36): one process, cached zkey, 2 warm-up + 10 proves,
    on a bit-heavy stand-in circuit; every proof equals the oracle's for the same (r, s)."""
    K = gpu
    r1, w = S.random_circuit(4000, 2, 30, bit_fraction=0.85, seed=11)
    zkey, _ = S.setup(r1, lambda g, sc: K.generator_mul(g, sc), points_to_mont=lambda a: O.fq_convert_montgomery(a, True))
    wtns = S.write_wtns(w)
    cm = K.CacheManager()
    cm.load("loop", zkey)
    cache = O.build_cache(O.parse_zkey(zkey))
    want = {}
    for i in range(12):
        r, s = 1000 + i % 3, 77
        pj, qj, _ = cm.prove_mem("loop", wtns, r, s)
        if (r, s) not in want:
            want[(r, s)] = O.groth16_prove(zkey, wtns, r, s, cache=cache)
        assert json.loads(pj) == want[(r, s)][0] and json.loads(qj) == want[(r, s)][1]
    cm.close()


def test_rccl_exchange_single_rank(gpu):
    """the RCCL data plane with world size 1 (a 1-GPU box cannot host two ranks): unique id, communicator, all-gather
    and max-reduce through csrc/comm/rccl_comm.cpp on this library's HIP runtime, then the device all-to-all and the in-place
    device all-gather on buffers of the size one of eight ranks moves, both host-synchronous (one process per GPU) and
    enqueued on the caller's stream through communicators of ncclCommInitAll (device group in one process).  Runs in a fresh interpreter
    because our librccl must be loaded before torch's bundled one (parallel.preload_rccl)."""
    import subprocess
    code = r'''
import importlib, os, socket, sys
sys.path.insert(0, %r)
P = importlib.import_module("icicle-snark_amd.parallel")
K = importlib.import_module("icicle-snark_amd")
P.preload_rccl()
import torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("gloo", rank=0, world_size=1)
K.set_device("HIP", 0)
ex = P.RcclExchange(0)
blk = bytes(range(256)) * 2 + bytes(64)
assert ex.allgather(blk) == blk
assert ex.max(3.25) == 3.25
# the two DEVICE collectives of a sharded prove at the sizes one of eight ranks moves at 1.6 M constraints: the all-to-all of
# the distributed QAP front end (3 rows of 2^18 elements = 24 MiB) and the in-place witness all-gather (6.4 MB slice)
import numpy as np
rng = np.random.default_rng(5)
rows, row_bytes = 3, (1 << 18) * 32
payload = rng.integers(0, 256, size=rows * row_bytes, dtype=np.uint8).tobytes()
d_send, d_recv = K.DeviceVec(rows * row_bytes), K.DeviceVec(rows * row_bytes)
K.raw_to_device(d_send.ptr, payload)
ex.alltoall_rows(d_send.ptr, d_recv.ptr, rows, row_bytes, row_bytes)
assert K.raw_to_host(d_recv.ptr, rows * row_bytes) == payload
sl = 6_400_032
wit = rng.integers(0, 256, size=sl, dtype=np.uint8).tobytes()
d_w = K.DeviceVec(sl)
K.raw_to_device(d_w.ptr, wit)
ex.allgather_device(d_w.ptr, sl)
assert K.raw_to_host(d_w.ptr, sl) == wit
# the same collectives the way a device group uses them (csrc/prover/multi.cpp): communicators of ncclCommInitAll, enqueued on
# the CALLER's stream without a host wait, then read back in stream order
import ctypes as C
lib = ex.lib
comms = (C.c_void_p * 1)()
devs = (C.c_int * 1)(0)
assert lib.icicle_snark_rccl_init_all(1, devs, comms) == 0, lib.icicle_snark_rccl_last_error()
st = K.IcicleStream()
K.raw_to_device(d_recv.ptr, bytes(rows * row_bytes))
assert lib.icicle_snark_rccl_alltoall_rows_on(C.c_void_p(comms[0]), C.c_void_p(d_send.ptr), C.c_void_p(d_recv.ptr), rows, C.c_size_t(row_bytes), C.c_size_t(row_bytes), C.c_void_p(st.handle)) == 0
assert lib.icicle_snark_rccl_allgather_device_on(C.c_void_p(comms[0]), C.c_void_p(d_w.ptr), C.c_size_t(sl), C.c_void_p(st.handle)) == 0
st.synchronize()
assert K.raw_to_host(d_recv.ptr, rows * row_bytes) == payload and K.raw_to_host(d_w.ptr, sl) == wit
lib.icicle_snark_rccl_destroy(C.c_void_p(comms[0]))
st.destroy(); d_send.free(); d_recv.free(); d_w.free()
ex.barrier(); ex.close()
dist.destroy_process_group()
print("RCCL_OK", [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l][:1])
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "RCCL_OK" in out.stdout, out.stdout + out.stderr
    assert "/opt/rocm" in out.stdout, out.stdout


def test_bench_two_ranks_on_one_gpu(gpu):
    """bench.py's N > 1 control flow end to end under torchrun with two ranks pinned to the one GPU of the box: rank 0 proves
    through the in-process device group ("HIP:0,0"), then both ranks run the rank-per-GPU host with the gloo exchange standing
    in for RCCL (RCCL refuses two ranks on one device): point-range shards, all-gather of the 576-byte blocks, group sum,
    blinding/JSON, max-over-ranks timing, one JSON line from rank 0."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ICICLE_SNARK_BENCH_DEVICE="0", ICICLE_SNARK_BENCH_EXCHANGE="gloo", ICICLE_SNARK_BENCH_DEVICES="0,0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--constraints", "100000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    # both ranks are pinned to the ONE GPU of the box: the line must not claim two (round-4 advisor)
    assert d["n_gpus"] == 1 and d["config"]["requested_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["msm_sharding"] == "point-range x2"
    assert d["value"] > 0 and d["roofline"]["achieved"] > 0
    # `value` comes from the library's own entry (rank 0: groth16_prove with the device list "HIP:0,0"), the rank-per-GPU host is timed beside it
    assert d["config"]["device_group"]["shards"] == 2 and d["config"]["device_group"].get("error") is None
    assert d["config"]["host"].startswith("one process, one host thread per GPU") and d["config"]["prove_ms_rank_per_gpu"] > 0


def test_bench_gpus_without_a_launcher(gpu):
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset (no torchrun): the library's own multi-device entry — one process,
    groth16_prove with the device list "HIP:0,0" — is timed in a child of bench.py; rc 0, ONE JSON line, and the line says which
    transport moved the exchanges, how many devices the group touched, how many RCCL ranks took part (0: the pull / memcpy
    transports do not go through RCCL) and that the group's proofs equal the single-device ones (two witnesses)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ICICLE_SNARK_BENCH_DEVICES"] = "0,0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--constraints", "100000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d["config"]
    # two shards aliased to ONE device: n_gpus says what was used, config.requested_gpus what was asked for
    assert d["n_gpus"] == 1 and c["requested_gpus"] == 2 and d["fallback"] is False
    assert d["scaling"] == "strong" and d["value"] > 0 and d["roofline"]["achieved"] > 0
    assert c["msm_sharding"] == "point-range x2" and c["launcher"].startswith("none")
    assert c["exchange"] in ("pull", "memcpy") and c["rccl_ranks"] == 0 and c["devices_touched"] == 1
    g = c["device_group"]
    assert g["shards"] == 2 and g["devices"] == [0, 0] and g["equals_single_device_proof"] is True
    assert g["attempts"][-1]["error"] is None and len(g["attempts"]) == 1


def test_bench_gpus_without_a_launcher_falls_back(gpu):
    """the same entry when every device-group attempt fails (a device that does not exist): the ladder — library default
    transports, memcpy forced, rccl forced — is walked, the line still appears from the single-device prove and says so"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ICICLE_SNARK_BENCH_DEVICES"] = "0,97"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--constraints", "100000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 3, out.stderr[-3000:]     # the line is printed, the exit code says that `--gpus 2` did not happen
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert d["value"] > 0 and "FALLBACK" in c["msm_sharding"]
    assert d["fallback"] is True and d["n_gpus"] == 1 and c["requested_gpus"] == 2
    assert [a["error"] is None for a in c["device_group"]["attempts"]] == [False, False, False, True]


@pytest.mark.parametrize("fault", ["abort", "hang", "no_such_device"])
def test_bench_survives_a_failing_device_group(gpu, fault):
    """The device-group leg of bench.py runs in a child process of rank 0: when that process aborts, hangs (ended after
    ICICLE_SNARK_GROUP_TIMEOUT) or fails (a device that does not exist), the launcher's ranks agree on the fallback and the one
    JSON line carries the one-process-per-GPU host's numbers with the error recorded."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ICICLE_SNARK_BENCH_DEVICE="0", ICICLE_SNARK_BENCH_EXCHANGE="gloo", ICICLE_SNARK_BENCH_DEVICES="0,0", ICICLE_SNARK_GROUP_TIMEOUT="20")
    if fault == "no_such_device":
        env["ICICLE_SNARK_BENCH_DEVICES"] = "0,97"
    else:
        env["ICICLE_SNARK_BENCH_GROUP_FAULT"] = fault
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--constraints", "100000", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["device_group"]["error"] and d["config"]["host"] == "one process per GPU"
    assert abs(d["ms_per_step"] - d["config"]["prove_ms_rank_per_gpu"]) < 1e-9


def test_benchmark_3200k_sharded_commitments(gpu, O):
    """BASELINE config 3 (benchmark/3200k, MSMs sharded over 8 GPUs) through the entry a caller uses — one key over a device
    group of EIGHT shards in one process (csrc/prover/multi.cpp; all eight "devices" are GPU 0 on this box): point-range shards
    of A, B1, B2, C, residue-class shards of H, 1/8 witness slices + device all-gather, the distributed QAP front end with its
    two all-to-alls.  The proof equals the unsharded prover's and the CPU oracle's and passes the reference pairing check."""
    K = gpu
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    N = 3_200_000
    zkey, wtns = bench.make_inputs(K, S, N)
    cm = K.CacheManager()
    cm.load("full", zkey)
    assert cm.info("full").domain_size == 1 << 22
    want, public, _ = cm.prove_mem("full", wtns, 5, 9)
    cm.evict("full")
    cm.load_devices("g8", zkey, [0] * 8)
    info = cm.info("g8")
    assert info.shards == 8 and info.domain_size == 1 << 22
    for rep in range(2):
        got, pub, _ = cm.prove_mem("g8", wtns, 5, 9)
        assert got == want and pub == public, rep
    assert json.loads(public) == [str(pow(3, 1 << N, S.R_MOD))]
    # … and the proof itself equals the CPU oracle's (≈ 15 s of host time at this size)
    O.calibrate_threads()
    proof, pub = O.groth16_prove(zkey, wtns, 5, 9)
    assert json.loads(want) == proof and json.loads(public) == pub
    vk = _vk_of(O, zkey)
    assert K.groth16_verify_json(want, public, S.vk_to_json(vk))
    import ref as R
    if R.available():
        assert R.groth16_verify(json.loads(want), json.loads(public), vk)
    cm.close()
    K.release_domain()


def test_prove_at_domain_2p23_table_and_classic_layouts(gpu, O, monkeypatch):
    """Above every BASELINE size: a squaring chain of 6.4 M constraints (domain 2^23, 3.1 GB zkey, 205 MB witness).  The 32-bit
    sort entry has no room for a 23-bit point index beside 20-bit digits: tables would fall back to narrower digits, which is no
    faster than the classic layout with one bucket set per window (what the reference's bucket method does when it lowers c,
    cuda_msm.cuh:1204-1254), so a key of this size stays classic by default; ICICLE_SNARK_TABLES=2 forces the tables.  No oracle at this size: the checks are the size-independent ones — the public
    signal 3^(2^N), determinism under fixed (r, s), the table-mode and the classic-layout provers agreeing bit for bit, the
    library's pairing check and the reference's (when oracle/_ref is present)."""
    K = gpu
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    N = 6_400_000
    zkey, wtns = bench.make_inputs(K, S, N)
    cm = K.CacheManager()
    monkeypatch.setenv("ICICLE_SNARK_TABLES", "2")     # tables although the key is above the size where they pay (cache.cpp)
    cm.load("big", zkey)
    info = cm.info("big")
    assert (info.n_vars, info.domain_size) == (N + 2, 1 << 23)
    import time
    t0 = time.perf_counter()
    p1, q1, tm = cm.prove_mem("big", wtns, 5, 9)
    first_ms = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        p2, q2, tm = cm.prove_mem("big", wtns, 5, 9, resident=True)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert p2 == p1 and q2 == q1
    geom = K.msm_profile(0)[1]
    print(f"[2^23] table layout: c = {geom['c']}, W = {geom['W']}, {info.device_bytes / 1e9:.1f} GB; first prove {first_ms:.1f} ms, resident {sorted(ts)[1]:.1f} ms (qap {tm.qap_ms:.1f}, msm {tm.msm_ms:.1f})")
    assert json.loads(q1) == [str(pow(3, 1 << N, S.R_MOD))]
    vk = _vk_of(O, zkey)
    assert K.groth16_verify_json(p1, q1, S.vk_to_json(vk))
    import ref as R
    if R.available():
        assert R.groth16_verify(json.loads(p1), json.loads(q1), vk)
    cm.evict("big")
    # the classic layout (one bucket set per window, 16-bit digits) — what a key of this size gets by default: the same proof
    monkeypatch.delenv("ICICLE_SNARK_TABLES")
    cm.load("classic", zkey)
    assert cm.info("classic").device_bytes < 6e9
    t0 = time.perf_counter()
    p3, q3, _ = cm.prove_mem("classic", wtns, 5, 9)
    t0 = time.perf_counter()
    p3b, _, tm3 = cm.prove_mem("classic", wtns, 5, 9, resident=True)
    classic_ms = (time.perf_counter() - t0) * 1e3
    geom3 = K.msm_profile(0)[1]
    print(f"[2^23] classic layout: c = {geom3['c']}, W = {geom3['W']}, {cm.info('classic').device_bytes / 1e9:.1f} GB; resident {classic_ms:.1f} ms (qap {tm3.qap_ms:.1f}, msm {tm3.msm_ms:.1f})")
    assert p3 == p1 and q3 == q1 and p3b == p1
    cm.close()
    K.release_domain()


def test_first_proves_on_adopted_tables_are_warm(gpu, O, tmp_path):
    """round-5 verdict item 3c: the proves right after a key's deferred fixed-base tables have been adopted (cache.cpp: adopt_tables) must
    not pay for anything the build could have prepared — the first three file-to-file proves on the tables stay within 1.25 × the
    median of the eight that follow (benchmark/1600k, cold file prove → proves beside the build → adoption)."""
    import importlib
    import time
    K = gpu
    bench = importlib.import_module("bench")
    S = importlib.import_module("icicle-snark_amd.synth")
    zkey, wtns = bench.make_inputs(K, S, 1_600_000)
    zp, wp, pp, qp = (str(tmp_path / n) for n in ("c.zkey", "w.wtns", "proof.json", "public.json"))
    open(zp, "wb").write(zkey)
    open(wp, "wb").write(wtns)
    cm = K.CacheManager()
    try:
        key = f"{zp}_HIP"
        # a cost the adoption leaves behind shows in EVERY cycle, a stall of the box's host (shared with other jobs) in one: two
        # cycles must pass, at most two may miss
        passed, seen = 0, []
        for cycle in range(4):
            cm.prove_files(wp, zp, pp, qp)                      # cold: sections cross PCIe beside the first proof
            while not cm.tables_ready(key):
                cm.prove_files(wp, zp, pp, qp)                  # classic layout beside the build
            ts = []
            for _ in range(11):
                t = time.perf_counter()
                cm.prove_files(wp, zp, pp, qp)
                ts.append((time.perf_counter() - t) * 1e3)
            med = sorted(ts[3:])[4]
            seen.append([round(x, 2) for x in ts])
            passed += max(ts[:3]) <= 1.25 * med
            cm.evict(key)
            if passed == 2:
                break
        print(f"[adoption] first three / median of the next eight per cycle: {[round(max(t[:3]) / sorted(t[3:])[4], 3) for t in seen]}")
        assert passed == 2, seen
    finally:
        cm.close()
        K.release_domain()
