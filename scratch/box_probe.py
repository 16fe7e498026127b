"""what kind of box is this: access-pattern rates + the table build of benchmark/1600k alone + the digit sort of 2^21 dense scalars"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
print("probes:", K.access_probes(), "copy/mad:", tuple(round(x, 1) for x in K.microbench()), flush=True)
zkey, wtns = bench.make_inputs(K, S, 1600000)
cm = K.CacheManager()
for rep in range(2):
    cm.load("k", zkey, wait_tables=False)
    t1 = time.perf_counter()
    cm.tables_ready("k", wait=True)
    t2 = time.perf_counter()
    ts = []
    for _ in range(6):
        t = time.perf_counter(); cm.prove_mem("k", wtns); ts.append((time.perf_counter() - t) * 1e3)
    sort_ms = K.msm_profile(2)[0][4]
    print(f"rep {rep}: tables alone {1e3*(t2-t1):.0f} ms, proves {min(ts):.2f}-{max(ts):.2f} ms, witness tail sort in the prove {sort_ms:.2f} ms", flush=True)
    cm.evict("k")
print("probes again:", K.access_probes(), flush=True)
