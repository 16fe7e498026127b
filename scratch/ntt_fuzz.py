"""random sizes / batches / directions / value patterns of bn254_ntt against the CPU oracle (the radix-2^29 passes from 2^11 on, the
8x32-bit kernels below and for ragged plans): python scratch/ntt_fuzz.py [iterations]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
K = importlib.import_module("icicle-snark_amd"); O = importlib.import_module("oracle")
K.set_device("HIP", 0)
K.release_domain(); K.initialize_domain(K.get_root_of_unity(1 << 20))
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
R = O.R_MOD
def fr(v): return np.frombuffer(int(v % R).to_bytes(32, "little"), dtype=np.uint64)
bad = 0
for it in range(n_it):
    logn = int(rng.choice([rng.integers(0, 11), rng.integers(11, 17), rng.integers(11, 17), rng.integers(17, 20)]))
    batch = int(rng.integers(1, 4)) if logn <= 17 else 1
    n = 1 << logn
    kind = int(rng.integers(0, 4))
    if kind == 0:
        raw = rng.integers(0, 1 << 63, size=(batch * n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(batch * n, 4), dtype=np.uint64)
        raw[:, 3] &= np.uint64((1 << 61) - 1)
        x = raw
    else:
        pool = np.stack([fr(R - 1), fr(R - 2), fr(0), fr(1), fr(R >> 1), fr((R >> 1) + 1)])
        x = np.ascontiguousarray(pool[rng.integers(0, len(pool) if kind == 1 else 2, size=batch * n)])
    inverse = bool(rng.integers(0, 2))
    got = K.ntt(x, inverse, batch_size=batch)
    want = O.fr_ntt(x, inverse, batch=batch, domain_log=20)
    if not np.array_equal(got, want):
        bad += 1
        print(f"MISMATCH logn={logn} batch={batch} inverse={inverse} kind={kind}", flush=True)
print(f"ntt fuzz: {n_it} transforms, {bad} mismatches")
