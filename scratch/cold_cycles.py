"""Evict-and-prove cycles of one key (round-5 verdict item 3: the outliers of the cold path).  Per cycle: cold file-to-file prove,
proves beside the deferred table build (median / max), ms until the tables are adopted, the first three proves on the adopted
tables; then `load` alone: ms until usable / ms of the table build with nothing beside it.
usage: cold_cycles.py [cycles] [pre]   pre = none | malloc9 (hipMalloc + memset + hipFree of 9 GB in this process before every cycle)
                                            | child9 (a child process that allocates, touches and frees 9 GB, then exits, before every cycle)"""
import ctypes as C, importlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pre = sys.argv[2] if len(sys.argv) > 2 else "none"
cache = f"/tmp/isnark_inputs_{N}"
if os.path.exists(cache + ".zkey"):
    zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
else:
    zkey, wtns = bench.make_inputs(K, S, N)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
tmp = tempfile.mkdtemp(prefix="cold_")
zp, wp, pp, qp = (os.path.join(tmp, n) for n in ("c.zkey", "w.wtns", "proof.json", "public.json"))
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]; hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

def free_gb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 1e9

def pre_action():
    if pre == "malloc9":
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), 9 << 30) == 0
        hip.hipMemset(p, 1, 9 << 30); hip.hipDeviceSynchronize(); hip.hipFree(p)
    elif pre == "child9":
        subprocess.run([sys.executable, "-c", "import ctypes as C; h=C.CDLL('libamdhip64.so'); p=C.c_void_p(); h.hipMalloc.argtypes=[C.POINTER(C.c_void_p),C.c_size_t]; h.hipMemset.argtypes=[C.c_void_p,C.c_int,C.c_size_t]; assert h.hipMalloc(C.byref(p), 9<<30)==0; h.hipMemset(p,1,9<<30); h.hipDeviceSynchronize()"], check=True)

cm = K.CacheManager()
key = f"{zp}_HIP"
t = lambda f: (lambda t0: (f(), (time.perf_counter() - t0) * 1e3)[1])(time.perf_counter())
for c in range(cycles):
    pre_action()
    fb = free_gb()
    cold = t(lambda: cm.prove_files(wp, zp, pp, qp))
    during, t0 = [], time.perf_counter()
    while not cm.tables_ready(key):
        during.append(t(lambda: cm.prove_files(wp, zp, pp, qp)))
    adopted = (time.perf_counter() - t0) * 1e3
    after = [t(lambda: cm.prove_files(wp, zp, pp, qp)) for _ in range(4)]
    d = sorted(during)
    print(f"cycle {c} [{pre}] free {fb:6.1f} GB | cold prove {cold:7.1f} ms | beside build: n {len(d):3d} median {d[len(d)//2] if d else 0:6.1f} max {d[-1] if d else 0:6.1f} | tables adopted after {adopted:7.1f} ms | "
          f"then {' '.join(f'{x:.1f}' for x in after)}", flush=True)
    cm.evict(key)
for c in range(3):
    pre_action()
    t0 = time.perf_counter()
    cm.load(key, zkey, wait_tables=False)
    usable = (time.perf_counter() - t0) * 1e3
    cm.tables_ready(key, wait=True)
    print(f"load alone [{pre}]: usable after {usable:6.1f} ms, tables after another {(time.perf_counter() - t0) * 1e3 - usable:7.1f} ms", flush=True)
    cm.evict(key)
