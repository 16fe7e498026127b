"""distribution of the prove time with the witness handed over as a host buffer (staged upload on three lanes):
looks for the occasional stall of an upload worker; prints min / median / p90 / max over 60 calls per size"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
for N in (400_000, 1_600_000):
    zkey, wtns = bench.make_inputs(K, S, N)
    cm = K.CacheManager(); cm.load("k", zkey)
    cm.prove_mem("k", wtns)
    w, h = [], []
    for i in range(60):
        t = time.perf_counter()
        _, _, tm = cm.prove_mem("k", wtns)
        w.append((time.perf_counter() - t) * 1e3); h.append(tm.h2d_ms)
    ws, hs = sorted(w), sorted(h)
    print(f"N={N}: wall min {ws[0]:.2f} median {ws[30]:.2f} p90 {ws[54]:.2f} max {ws[-1]:.2f} ms; h2d min {hs[0]:.2f} median {hs[30]:.2f} p90 {hs[54]:.2f} max {hs[-1]:.2f} ms")
    print("   outliers (wall, h2d):", [(round(a, 1), round(b, 1)) for a, b in zip(w, h) if a > 1.3 * ws[30]])
    cm.close()
print(open("/proc/loadavg").read())
