"""one rank's work of an N-way sharded prove of benchmark/1600k on a single GPU (no exchange): groth16_commitments of
shard `rank` of `count` with the witness resident.  usage: shard_rank_time.py [count] [rank]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
zkey, wtns = bench.make_inputs(K, S, 1_600_000)
cm = K.CacheManager()
cm.load("s", zkey, shard_rank=rank, shard_count=count)
cm.commitments("s", wtns)
for _ in range(3):
    cm.commitments("s", None)
K.check(K.lib().icicle_device_synchronize())
t = time.perf_counter(); n = 10
acc = [0.0, 0.0]
for _ in range(n):
    _, tm = cm.commitments("s", None)
    acc[0] += tm.qap_ms; acc[1] += tm.msm_ms
K.check(K.lib().icicle_device_synchronize())
print("shard %d/%d: %.2f ms per commitments() call (qap %.2f, msm phase %.2f)" % (rank, count, (time.perf_counter() - t) * 1e3 / n, acc[0] / n, acc[1] / n))
