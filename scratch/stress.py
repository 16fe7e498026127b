"""randomised parity campaign on the GPU box (not part of the test suite): MSMs of random length / scalar shape / flags
against the oracle, proves of random squaring chains with random shard counts against the oracle's proof.
usage: stress.py [seconds] [seed]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
R = O.R_MOD

def scalars(n, kind):
    if kind == "uniform":
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 61) - 1)
    elif kind == "bits":
        a = np.zeros((n, 4), dtype=np.uint64); a[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint64)
    elif kind == "small":
        a = np.zeros((n, 4), dtype=np.uint64); a[:, 0] = rng.integers(0, 1 << 20, size=n, dtype=np.uint64)
    elif kind == "top":   # r - 1 - small
        vals = [(R - 1 - int(x)) for x in rng.integers(0, 1 << 30, size=n)]
        a = O.ints_to_arr(vals)
    else:                 # all equal
        a = np.tile(O.ints_to_arr([int(rng.integers(1, 1 << 62))])[0], (n, 1)).copy()
    return a

G = {g: O.ec_to_affine(g, O.ec_generator(g)) for g in ("g1", "g2")}
t0 = time.time(); n_msm = n_prove = 0
while time.time() - t0 < budget:
    grp = "g1" if rng.random() < 0.6 else "g2"
    n = int(rng.integers(1, 200000 if grp == "g1" else 40000))
    if rng.random() < 0.3:
        n = int(2 ** rng.integers(1, 17)) + int(rng.integers(-2, 3))
        n = max(1, n)
    kind = ["uniform", "bits", "small", "top", "equal"][int(rng.integers(0, 5))]
    sc = scalars(n, kind)
    pts = O.fixed_base_mul(grp, G[grp], scalars(min(n, 64), "uniform"))
    bases = np.concatenate([pts] * ((n + len(pts) - 1) // len(pts)))[:n].copy()
    if n > 3 and rng.random() < 0.5:
        bases[int(rng.integers(0, n))] = 0
    want = O.ec_to_affine(grp, O.msm(grp, sc, bases))
    c = int(rng.choice([0, 0, 0, 7, 10, 13, 16]))
    got = K.ec(grp, "to_affine", K.msm(grp, sc, bases, c=c))
    assert np.array_equal(got, want), ("msm", grp, n, kind, c)
    n_msm += 1
    if n_msm % 3 == 0:
        N = int(rng.integers(2, 6000))
        count = int(rng.choice([1, 1, 2, 3, 4, 8]))
        r1, w = S.squaring_chain(N)
        zkey, vk = S.setup(r1, lambda g, k: K.generator_mul(g, k), points_to_mont=lambda a: O.fq_convert_montgomery(a, True))
        wtns = S.write_wtns(w)
        r, s = int(rng.integers(1, 1 << 62)), int(rng.integers(1, 1 << 62))
        proof, public = O.groth16_prove(zkey, wtns, r, s)
        cm = K.CacheManager()
        if count == 1:
            cm.load("k", zkey)
            pj, qj, _ = cm.prove_mem("k", wtns, r, s)
        else:
            blocks = b""
            for rank in range(count):
                cm.load(f"s{rank}", zkey, shard_rank=rank, shard_count=count)
                blk, _ = cm.commitments(f"s{rank}", wtns); blocks += blk
            cm.load("k", zkey)
            pj, qj = cm.assemble("k", wtns, K.sum_commitments(blocks, count), r, s)
        assert json.loads(pj) == proof and json.loads(qj) == public, ("prove", N, count)
        assert K.groth16_verify_json(pj, qj, S.vk_to_json(vk))
        cm.close(); n_prove += 1
print(f"stress ok: {n_msm} MSMs, {n_prove} proves in {time.time() - t0:.0f} s (seed {seed})")
