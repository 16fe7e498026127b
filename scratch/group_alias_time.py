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
    # (round 6: ten warm-up proves, then the two series INTERLEAVED — round 5 timed twelve resident proves right after the load and the
    #  host-witness ones behind them, and at eight shards the first ones after a load are slow: "resident 28.5 / host 25.0" was that order)
    res, host, first = [], [], []
    for i in range(10):
        t = time.perf_counter(); cm.prove_mem(key, wtns, 3, 4, resident=i % 2 == 1); first.append((time.perf_counter() - t) * 1e3)
    for _ in range(10):
        t = time.perf_counter(); cm.prove_mem(key, wtns, 3, 4, resident=True); res.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter(); cm.prove_mem(key, wtns, 3, 4); host.append((time.perf_counter() - t) * 1e3)
    res.sort(); host.sort()
    print(f"   first ten proves after the load: {' '.join(f'{x:.1f}' for x in first)}")
    tm = cm.last_timings(key)
    print(f"benchmark/{N}: {G} shard(s) on ONE GPU: cache build {cold:.0f} ms, {cm.info(key).device_bytes / 1e6:.0f} MB | prove resident median {res[len(res) // 2]:.2f} (min {res[0]:.2f}) ms, host witness median {host[len(host) // 2]:.2f} (min {host[0]:.2f}) ms "
          f"(slowest shard: upload {tm.h2d_ms:.2f}, qap {tm.qap_ms:.2f}, msm {tm.msm_ms:.2f})")
    cm.evict(key)
cm.close()
