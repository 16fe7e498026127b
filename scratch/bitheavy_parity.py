"""bit-heavy witness at 100k constraints (table mode, large buckets at scale): GPU proof == oracle proof for the same
(not circuit-satisfying) witness vector — both sides compute the same algebra, no verification involved."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
zkey, wtns = bench.make_inputs(K, S, N)
rng = np.random.default_rng(3)
w = np.frombuffer(wtns, dtype=np.uint8).copy()
body = w[len(w) - 32 * (N + 2):].view(np.uint64).reshape(-1, 4)
kind = rng.random(N + 2)
bits = kind < 0.7
body[bits] = 0; body[bits, 0] = rng.integers(0, 2, size=int(bits.sum()), dtype=np.uint64)
small = (kind >= 0.7) & (kind < 0.8); body[small, 1:] = 0
body[0] = 0; body[0, 0] = 1
skewed = w.tobytes()
cm = K.CacheManager(); cm.load("k", zkey)
pj, qj, _ = cm.prove_mem("k", skewed, 5, 7)
O.set_num_threads(O.calibrate_threads()) if hasattr(O, "set_num_threads") else None
proof, public = O.groth16_prove(zkey, skewed, 5, 7)
assert json.loads(pj) == proof and json.loads(qj) == public
for count in (2, 4):
    blocks = b""
    for rank in range(count):
        cm.load(f"s{count}{rank}", zkey, shard_rank=rank, shard_count=count)
        blk, _ = cm.commitments(f"s{count}{rank}", skewed); blocks += blk
        cm.evict(f"s{count}{rank}")
    got, _ = cm.assemble("k", skewed, K.sum_commitments(blocks, count), 5, 7)
    assert got == pj, count
print("bit-heavy parity ok at N =", N)
