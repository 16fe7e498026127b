import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(sys.argv[1])
zkey, wtns = bench.make_inputs(K, S, N)
cm = K.CacheManager(); cm.load("k", zkey)
for _ in range(5): cm.prove_mem("k", wtns)
_, _, tm = cm.prove_mem("k", wtns, resident=True)
ms, g = K.msm_profile(2)
print(f"chain {N}: B2 acc {ms[1]:.3f} ms (L={g['L']}, c={g['c']}) qap {tm.qap_ms:.2f} msm {tm.msm_ms:.2f}")
