"""N resident proves of a bench workload (default aadhaar_standin) in one process: median ms, phase times and the HIP-event
profile of the five MSMs of the last prove (A/B helper for ICICLE_SNARK_* knobs, which the library reads once per process or
per cache build)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
wl = os.environ.get("LOOP_WORKLOAD", "aadhaar_standin")
cache = f"/tmp/isnark_inputs_{wl}"
if os.path.exists(cache + ".zkey"):
    zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
else:
    zkey, wtns, N, what, standin = bench.workload_inputs(K, S, wl)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
cm = K.CacheManager(); cm.load("k", zkey)
info = cm.info("k")
cm.prove_mem("k", wtns)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
res, q, m = [], 0.0, 0.0
for i in range(n):
    t = time.perf_counter(); _, _, tm = cm.prove_mem("k", wtns, resident=True); res.append((time.perf_counter() - t) * 1e3)
    q += tm.qap_ms; m += tm.msm_ms
host = []
for i in range(n):
    t = time.perf_counter(); cm.prove_mem("k", wtns); host.append((time.perf_counter() - t) * 1e3)
res.sort(); host.sort()
print(f"{wl}: host-witness median {host[n // 2]:.3f} min {host[0]:.3f}")
print(f"{wl}: b_bases {info.b_bases} of {info.n_vars} | resident median {res[n // 2]:.3f} min {res[0]:.3f} | qap {q / n:.3f} msm {m / n:.3f}")
for back, name in zip(range(4, -1, -1), ("A", "B1", "B2", "C", "H")):
    ms, g = K.msm_profile(back)
    print(f"   {name:2s} L={g['L']:8d} c={g['c']} W={g['W']} sort+wait {ms[0]:.3f} acc {ms[1]:.3f} reduce {ms[2]:.3f} total {ms[3]:.3f} sort-only {ms[4]:.3f}")
