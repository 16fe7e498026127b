"""host-side trace of the file-to-file prove (groth16_prove) at benchmark/1600k: ICICLE_SNARK_TRACE_HOST=1 marks + wall clock per call"""
import importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
zkey, wtns = bench.make_inputs(K, S, N)
d = tempfile.mkdtemp()
zp, wp = d + "/c.zkey", d + "/w.wtns"
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
cm = K.CacheManager()
for i in range(int(os.environ.get('CALLS', '8'))):
    if i == 1:
        cm.tables_ready(zp + "_HIP", wait=True)
    t = time.perf_counter()
    cm.prove_files(wp, zp, d + "/p.json", d + "/q.json")
    print(f"call {i}: {1e3 * (time.perf_counter() - t):.3f} ms", file=sys.stderr, flush=True)
print("---- host-witness (prove_mem) ----", file=sys.stderr, flush=True)
for i in range(4):
    t = time.perf_counter()
    cm.prove_mem(zp + "_HIP", wtns)
    print(f"mem call {i}: {1e3 * (time.perf_counter() - t):.3f} ms", file=sys.stderr, flush=True)
