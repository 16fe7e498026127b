"""per-step wall clock of bench.py's N = 1 timed loop (same sequence: cache load, warm-up, K file-to-file proves), to see whether
the mean the driver reads is the median of the steps or carries outliers"""
import importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, 1_600_000)
d = tempfile.mkdtemp(prefix="isnark_bench_")
zp, wp = d + "/circuit.zkey", d + "/witness.wtns"
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
cm = K.CacheManager(); key = zp + "_HIP"; cm.load(key, zkey)
for rep in range(3):
    for _ in range(2):
        cm.prove_files(wp, zp, d + "/proof.json", d + "/public.json")
    K.check(K.lib().icicle_device_synchronize())
    ts = []
    t0 = time.perf_counter()
    for _ in range(10):
        t = time.perf_counter(); cm.prove_files(wp, zp, d + "/proof.json", d + "/public.json"); ts.append((time.perf_counter() - t) * 1e3)
    K.check(K.lib().icicle_device_synchronize())
    print("mean %.3f | " % ((time.perf_counter() - t0) * 1e3 / 10) + " ".join("%.2f" % x for x in ts))
