"""device / host memory over many (a) cold proves from files + evict, (b) load + prove + evict cycles of one small key"""
import importlib, os, sys, time, resource, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "100000"))
zkey, wtns = bench.make_inputs(K, S, N)
d = tempfile.mkdtemp()
zp, wp = d + "/c.zkey", d + "/w.wtns"
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
cm = K.CacheManager()
def rss():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024
import ctypes as C
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 1e6
mode = sys.argv[1] if len(sys.argv) > 1 else "cold"
wait = os.environ.get("WAIT_TABLES", "0") == "1"
for i in range(int(os.environ.get("CYCLES", "400")) + 1):
    if i % 50 == 0:
        K.check(K.lib().icicle_device_synchronize())
        print(f"{mode} cycle {i}: device free {free_mb():.0f} MB, host RSS {rss():.0f} MB", flush=True)
    if mode == "cold":
        cm.prove_files(wp, zp, d + "/p.json", d + "/q.json")
        if wait:
            cm.tables_ready(zp + "_HIP", wait=True)
        cm.evict(zp + "_HIP")
    else:
        cm.load("k", zkey, wait_tables=wait)
        cm.prove_mem("k", wtns)
        cm.evict("k")
