import importlib, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, 400_000)
def free():
    a, b = C.c_size_t(), C.c_size_t()
    K.check(K.lib().icicle_get_available_memory(C.byref(a), C.byref(b)))
    return b.value / 1e6
cm = K.CacheManager()
base = None
for i in range(6):
    cm.load("k", zkey); cm.prove_mem("k", wtns); cm.prove_mem("k", wtns, resident=True); cm.evict("k")
    f = free()
    if i == 1: base = f
    print("cycle %d: free %.0f MB" % (i, f), flush=True)
assert abs(free() - base) < 64, "device memory keeps shrinking across load/prove/evict cycles"
print("no leak")
