"""cold path from a FILE (groth16_cache_load_file, as the first groth16_prove of a key does): ms per build, page cache warm"""
import importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, 1_600_000)
d = tempfile.mkdtemp()
zp, wp = d + "/c.zkey", d + "/w.wtns"
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
ts = []
for i in range(4):
    cm = K.CacheManager()
    t = time.perf_counter(); cm.prove_files(wp, zp, d + "/proof.json", d + "/public.json"); ts.append((time.perf_counter() - t) * 1e3)
    cm.close()
print("first groth16_prove of a key (cache build from the file + prove): " + ", ".join("%.0f" % x for x in ts) + " ms")
