"""N resident proves of benchmark/1600k in one process; prints median / min ms and the phase times (A/B helper: the library
reads its ICICLE_SNARK_* knobs once per process)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
cache = f"/tmp/isnark_inputs_{N}"
if os.path.exists(cache + ".zkey"):
    zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
else:
    zkey, wtns = bench.make_inputs(K, S, N)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
cm = K.CacheManager(); cm.load("k", zkey)
cm.prove_mem("k", wtns)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
res, host, q, m = [], [], 0.0, 0.0
for i in range(n):
    t = time.perf_counter(); _, _, tm = cm.prove_mem("k", wtns, resident=True); res.append((time.perf_counter() - t) * 1e3)
    q += tm.qap_ms; m += tm.msm_ms
for i in range(10):
    t = time.perf_counter(); cm.prove_mem("k", wtns); host.append((time.perf_counter() - t) * 1e3)
res.sort(); host.sort()
print(f"resident median {res[n // 2]:.3f} min {res[0]:.3f} | host-witness median {host[5]:.3f} | qap {q / n:.3f} msm {m / n:.3f}")
