"""soak of the round-5 host paths: keys loaded with their tables DEFERRED, proved while the worker thread builds them, evicted at random
points of the build (before it started, in the middle, after the adoption), two keys alive at a time, the three entry points, a second
thread asking groth16_cache_tables_ready meanwhile; every fixed-(r, s) proof identical to the first one; free device memory and host RSS
before / after.   usage: soak_r05.py [cycles]   (LOOP_CONSTRAINTS, default 400000)"""
import importlib, os, random, sys, tempfile, threading, time, resource, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N = int(os.environ.get("LOOP_CONSTRAINTS", "400000"))
zkey, wtns = bench.make_inputs(K, S, N)
d = tempfile.mkdtemp()
zp, wp = d + "/c.zkey", d + "/w.wtns"
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 1e6
cm = K.CacheManager()
cm.load("ref", zkey)
ref = cm.prove_mem("ref", wtns, 11, 13)[:2]
cm.evict("ref")
rnd = random.Random(5)
stop = [False]
def poller():
    while not stop[0]:
        for k in ("a", "b"):
            try:
                cm.tables_ready(k)
            except Exception:
                pass   # the key is not there at the moment
        time.sleep(0.0005)
th = threading.Thread(target=poller, daemon=True); th.start()
K.check(K.lib().icicle_device_synchronize(), "sync")
m0, r0, t0 = free_mb(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3, time.time()
bad = proves = 0
stats = {"evicted before the build": 0, "evicted in the build": 0, "evicted after adoption": 0}
for c in range(cycles):
    key = "ab"[c % 2]
    cm.load(key, zkey, wait_tables=False)
    mode = rnd.randrange(3)
    if mode == 0:                       # no prove: the build has not started (grace time) → evict cancels it
        stats["evicted before the build"] += 1
    else:
        n = rnd.randrange(1, 4) if mode == 1 else 10 ** 6
        i = 0
        while i < n:
            rdy = cm.tables_ready(key)
            got = (cm.prove_mem(key, wtns, 11, 13, resident=(i % 2 == 1 and i > 0)) if i % 3 else cm.prove_mem(key, wtns, 11, 13))[:2]
            bad += got != ref; proves += 1; i += 1
            if rdy and mode == 2 and i >= 3:
                break
        stats["evicted in the build" if mode == 1 and not cm.tables_ready(key) else "evicted after adoption"] += 1
    # the OTHER key (if alive) proves once while this one is in whatever state
    other = "ab"[(c + 1) % 2]
    if cm.contains(other):
        bad += cm.prove_mem(other, wtns, 11, 13)[:2] != ref; proves += 1
        cm.evict(other)
    if c % 7 == 6:
        cm.prove_files(wp, zp, d + "/p.json", d + "/q.json"); proves += 1
        cm.evict(zp + "_HIP")
for k in ("a", "b"):
    cm.evict(k)
stop[0] = True; th.join()
K.check(K.lib().icicle_device_synchronize(), "sync")
m1, r1 = free_mb(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3
cur = [int(l.split()[1]) / 1024 for l in open("/proc/self/status") if l.startswith("VmRSS")][0]
print(f"{cycles} load / prove / evict cycles at {N} constraints, {proves} proves in {time.time() - t0:.1f} s: {bad} differing proofs; {stats}; "
      f"free device memory {m0:.0f} -> {m1:.0f} MB; host max RSS {r0:.0f} -> {r1:.0f} MB (resident at the end: {cur:.0f} MB)")
cm.close()
