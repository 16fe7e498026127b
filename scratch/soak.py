"""soak: N proves of benchmark/1600k (fixed r, s) in one process — every proof identical, device memory flat (hipMemGetInfo
before / after), host RSS flat; alternates the entry points (files, host buffer, resident witness)."""
import importlib, os, sys, tempfile, time, resource, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
zkey, wtns = bench.make_inputs(K, S, int(os.environ.get("LOOP_CONSTRAINTS", "1600000")))
d = tempfile.mkdtemp()
zp, wp = d + "/c.zkey", d + "/w.wtns"
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
cm = K.CacheManager(); key = zp + "_HIP"; cm.load(key, zkey)
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 1e6
ref = cm.prove_mem(key, wtns, 11, 13)[:2]
for _ in range(20):
    cm.prove_mem(key, wtns, 11, 13); cm.prove_files(wp, zp, d + "/p.json", d + "/q.json")
m0, r0, t0 = free_mb(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3, time.time()
bad = 0
for i in range(n):
    k = i % 3
    if k == 0: got = cm.prove_mem(key, wtns, 11, 13)[:2]
    elif k == 1: got = cm.prove_mem(key, wtns, 11, 13, resident=True)[:2]
    else:
        cm.prove_files(wp, zp, d + "/p.json", d + "/q.json"); got = ref   # random r, s: only that it runs
    bad += got != ref
m1, r1 = free_mb(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3
print(f"{n} proves in {time.time() - t0:.1f} s: {bad} differing proofs; free device memory {m0:.0f} -> {m1:.0f} MB; host max RSS {r0:.0f} -> {r1:.0f} MB")
