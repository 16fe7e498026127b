"""one rank's work of an N-way sharded prove of benchmark/1600k on a single GPU (no exchange): (a) groth16_commitments of
shard `rank` of `count` with the witness resident and the replicated front end, (b) the distributed front end — stage 1,
stage 2 (their buffers exchanged with THEMSELVES: timing only, wrong numbers) and the finishing commitments call.
usage: shard_rank_time.py [count] [rank]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
zkey, wtns = bench.make_inputs(K, S, int(os.environ.get("LOOP_CONSTRAINTS", "1600000")))
cm = K.CacheManager()
cm.load("s", zkey, shard_rank=rank, shard_count=count)
sync = lambda: K.check(K.lib().icicle_device_synchronize())
cm.commitments("s", wtns)
for _ in range(3):
    cm.commitments("s", None)
sync()
t = time.perf_counter(); n = 10
acc = [0.0, 0.0]
for _ in range(n):
    _, tm = cm.commitments("s", None)
    acc[0] += tm.qap_ms; acc[1] += tm.msm_ms
sync()
print("shard %d/%d replicated front end: %.2f ms per commitments() call, witness resident (qap %.2f, msm phase %.2f)" % (rank, count, (time.perf_counter() - t) * 1e3 / n, acc[0] / n, acc[1] / n))
t = time.perf_counter()
for _ in range(n):
    cm.commitments("s", wtns)
sync()
rep_host = (time.perf_counter() - t) * 1e3 / n
if cm.dist_supported("s"):
    for _ in range(2):
        cm.dist_stage1("s", wtns); cm.dist_stage2("s"); cm.dist_exchange_done("s"); cm.commitments("s", None)
    sync()
    t1 = t2 = t3 = 0.0
    for _ in range(n):
        a = time.perf_counter(); cm.dist_stage1("s", wtns); b = time.perf_counter(); cm.dist_stage2("s"); cm.dist_exchange_done("s"); c = time.perf_counter()
        _, tm = cm.commitments("s", None); d = time.perf_counter()
        t1 += b - a; t2 += c - b; t3 += d - c
    print("shard %d/%d distributed front end (host witness each time): stage 1 %.2f ms (upload + spmv + size-n/G inverse transform), stage 2 %.2f ms, "
          "finish %.2f ms; total %.2f ms + two all-to-alls  vs  %.2f ms replicated with the same witness upload"
          % (rank, count, t1 * 1e3 / n, t2 * 1e3 / n, t3 * 1e3 / n, (t1 + t2 + t3) * 1e3 / n, rep_host))
    # witness distribution: this rank uploads 1/count of the witness (the all-gather is not timed: the other slices are already there)
    t0 = t1 = t2 = t3 = 0.0
    for _ in range(n):
        z = time.perf_counter(); cm.upload_witness_slice("s", wtns); cm.witness_ready("s")
        a = time.perf_counter(); cm.dist_stage1("s", None); b = time.perf_counter(); cm.dist_stage2("s"); cm.dist_exchange_done("s"); c = time.perf_counter()
        _, tm = cm.commitments("s", None); d = time.perf_counter()
        t0 += a - z; t1 += b - a; t2 += c - b; t3 += d - c
    print("shard %d/%d with 1/%d witness upload: slice upload %.2f ms, stage 1 %.2f ms, stage 2 %.2f ms, finish %.2f ms; total %.2f ms + all-gather + two all-to-alls"
          % (rank, count, count, t0 * 1e3 / n, t1 * 1e3 / n, t2 * 1e3 / n, t3 * 1e3 / n, (t0 + t1 + t2 + t3) * 1e3 / n))
