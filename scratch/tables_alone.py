"""time of the deferred fixed-base table build with nothing running beside it (load without tables, then wait for them)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
zkey, wtns = bench.make_inputs(K, S, N)
cm = K.CacheManager()
for rep in range(4):
    t = time.perf_counter()
    cm.load("k", zkey, wait_tables=False)
    t1 = time.perf_counter()
    cm.tables_ready("k", wait=True)
    t2 = time.perf_counter()
    cm.prove_mem("k", wtns)
    t3 = time.perf_counter()
    cm.evict("k")
    t4 = time.perf_counter()
    print(f"rep {rep}: load {1e3*(t1-t):.1f} ms, tables alone {1e3*(t2-t1):.1f} ms, first prove {1e3*(t3-t2):.1f} ms, evict {1e3*(t4-t3):.1f} ms", flush=True)
