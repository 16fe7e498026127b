import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, 1_600_000)
cm = K.CacheManager(); t = time.time(); cm.load("k", zkey); print("load ms", (time.time() - t) * 1e3)
for i in range(16):
    t = time.perf_counter()
    _, _, tm = cm.prove_mem("k", wtns, resident=(i % 2 == 1))
    print("resident" if i % 2 else "upload  ", "wall %.2f ms" % ((time.perf_counter() - t) * 1e3), "h2d %.2f qap %.2f msm %.2f total %.2f" % (tm.h2d_ms, tm.qap_ms, tm.msm_ms, tm.total_ms))
