"""the reference's timed region (groth16_prove on files, warm cache) with the host-side marks (ICICLE_SNARK_TRACE_HOST=1)"""
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
cm = K.CacheManager()
cm.load(zp + "_HIP", zkey)
for _ in range(3):
    cm.prove_files(wp, zp, d + "/proof.json", d + "/public.json")
ts = []
for _ in range(20):
    t = time.perf_counter(); cm.prove_files(wp, zp, d + "/proof.json", d + "/public.json"); ts.append((time.perf_counter() - t) * 1e3)
ts.sort()
print("groth16_prove on files, warm cache: median %.3f ms, min %.3f ms" % (ts[10], ts[0]))
