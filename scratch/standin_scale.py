"""resident prove time of the aadhaar-style stand-in circuit at a given scale (fraction of 1.0 M constraints); the knobs of the
library are read at cache build"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
scale = float(sys.argv[1])
cache = f"/tmp/isnark_standin_{scale}"
if os.path.exists(cache + ".zkey"):
    zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
else:
    zkey, wtns, _, nc = bench.make_standin_inputs(K, S, "aadhaar_standin", scale=scale)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
cm = K.CacheManager(); cm.load("k", zkey); info = cm.info("k")
cm.prove_mem("k", wtns)
res = []
for i in range(40):
    t = time.perf_counter(); cm.prove_mem("k", wtns, resident=True); res.append((time.perf_counter() - t) * 1e3)
res.sort()
print(f"scale {scale}: n_vars {info.n_vars} b_bases {info.b_bases} | resident median {res[20]:.3f} min {res[0]:.3f}")
if os.environ.get("SHOW_PROFILE"):
    _, _, tm = cm.prove_mem("k", wtns, resident=True)
    print(f"   qap {tm.qap_ms:.3f} msm {tm.msm_ms:.3f}")
    for back, name in zip(range(4, -1, -1), ("A", "B1", "B2", "C", "H")):
        ms, g = K.msm_profile(back)
        print(f"   {name:2s} L={g['L']:8d} c={g['c']} W={g['W']} sort+wait {ms[0]:.3f} acc {ms[1]:.3f} reduce {ms[2]:.3f} total {ms[3]:.3f} sort-only {ms[4]:.3f}")
