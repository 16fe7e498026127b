"""cold path timing: zkey bytes / file → device-resident cache (run on the GPU box)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_600_000
zkey, wtns = bench.make_inputs(K, S, N)
os.environ["ICICLE_SNARK_TRACE_COLD"] = "1"
cm = K.CacheManager()
for i in range(3):
    t = time.time(); cm.load(f"k{i}", zkey); print(f"load from memory #{i}: {(time.time()-t)*1e3:.1f} ms", flush=True)
    cm.evict(f"k{i}")
path = "/tmp/cold.zkey"
open(path, "wb").write(zkey)
for i in range(2):
    t = time.time(); cm.load_file(f"f{i}", path); print(f"load from file (page cache warm) #{i}: {(time.time()-t)*1e3:.1f} ms", flush=True)
    cm.evict(f"f{i}")
os.remove(path)
