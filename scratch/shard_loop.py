"""loop of one rank's work of an 8-way (argv[1]) sharded prove of benchmark/1600k with the distributed front end (buffers exchanged
with themselves: timing only) — for rocprofv3 timelines of a shard"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
cache = f"/tmp/isnark_inputs_{N}"
if os.path.exists(cache + ".zkey"):
    zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
else:
    zkey, wtns = bench.make_inputs(K, S, N)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
cm = K.CacheManager()
cm.load("s", zkey, shard_rank=0, shard_count=count)
ts = []
for _ in range(n):
    t = time.perf_counter()
    cm.upload_witness_slice("s", wtns); cm.witness_ready("s")
    cm.dist_stage1("s", None); cm.dist_stage2("s"); cm.dist_exchange_done("s")
    _, tm = cm.commitments("s", None)
    ts.append((time.perf_counter() - t) * 1e3)
ts.sort()
print(f"shard 0/{count}: median {ts[len(ts) // 2]:.3f} ms per rank step (slice upload + stage 1 + stage 2 + finish), finish qap {tm.qap_ms:.3f} msm {tm.msm_ms:.3f}")
