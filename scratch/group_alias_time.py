"""The in-process device group (csrc/prover/multi.cpp) with EVERY shard on GPU 0: G host threads, 6·G streams and the three
device-side exchanges share one device, so the time is the SUM of the shards' work plus the orchestration — a correctness-path
timing (what a 1-GPU box can measure), not a scaling figure.  Prints resident and host-witness prove times for G = 1, 2, 4, 8."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
zkey, wtns = bench.make_inputs(K, S, N)
cm = K.CacheManager()
want = None
for G in (1, 2, 4, 8):
    key = f"g{G}"
    t = time.perf_counter()
    cm.load_devices(key, zkey, [0] * G) if G > 1 else cm.load(key, zkey)
    cold = (time.perf_counter() - t) * 1e3
    p, _, _ = cm.prove_mem(key, wtns, 3, 4)
    want = want or p
    assert p == want
    res, host = [], []
    for _ in range(12):
        t = time.perf_counter(); cm.prove_mem(key, wtns, 3, 4, resident=True); res.append((time.perf_counter() - t) * 1e3)
    for _ in range(8):
        t = time.perf_counter(); cm.prove_mem(key, wtns, 3, 4); host.append((time.perf_counter() - t) * 1e3)
    res.sort(); host.sort()
    tm = cm.last_timings(key)
    print(f"benchmark/{N}: {G} shard(s) on ONE GPU: cache build {cold:.0f} ms, {cm.info(key).device_bytes / 1e6:.0f} MB | prove resident median {res[len(res) // 2]:.2f} ms, host witness median {host[len(host) // 2]:.2f} ms "
          f"(slowest shard: upload {tm.h2d_ms:.2f}, qap {tm.qap_ms:.2f}, msm {tm.msm_ms:.2f})")
    cm.evict(key)
cm.close()
