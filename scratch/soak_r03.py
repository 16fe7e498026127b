"""round-3 soak: the paths added this round, N iterations each in one process, device memory and host RSS before / after —
(1) a device group (4 shards on GPU 0) proving benchmark/400k through groth16_prove with a device list, host buffer and resident
witness; (2) bn254_msm / bn254_g2_msm over device-resident bases with the automatic tables (classic → build → hits), the bases
rewritten every 50 calls; (3) a witness-light stand-in key crossing its table rebuild.  Every fixed-(r, s) proof identical."""
import importlib, os, sys, tempfile, time, resource, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 1e6
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3

# (1) device group
zkey, wtns = bench.make_inputs(K, S, 400_000)
d = tempfile.mkdtemp(); zp, wp = d + "/c.zkey", d + "/w.wtns"
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
cm = K.CacheManager()
dev = "HIP:0,0,0,0"
cm.prove_files(wp, zp, d + "/p.json", d + "/q.json", dev)
key = f"{zp}_{dev}"
ref = cm.prove_mem(key, wtns, 11, 13)[:2]
for _ in range(10): cm.prove_mem(key, wtns, 11, 13)
m0, r0, t0, bad = free_mb(), rss(), time.time(), 0
for i in range(n):
    k = i % 3
    if k == 0: got = cm.prove_mem(key, wtns, 11, 13)[:2]
    elif k == 1: got = cm.prove_mem(key, wtns, 11, 13, resident=True)[:2]
    else: cm.prove_files(wp, zp, d + "/p.json", d + "/q.json", dev); got = ref
    bad += got != ref
print(f"(1) device group x4: {n} proves in {time.time() - t0:.1f} s: {bad} differing proofs; free device memory {m0:.0f} -> {free_mb():.0f} MB; host max RSS {r0:.0f} -> {rss():.0f} MB")
cm.evict(key)

# (2) automatic tables
rng = np.random.default_rng(3)
L = 1 << 17
sc = rng.integers(0, 1 << 62, size=(L, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)
b1 = K.generator_mul("g1", sc[::-1].copy()); b2 = K.generator_mul("g2", sc[::-1].copy())
d_s, d_b1, d_b2 = K.DeviceVec.from_host(sc), K.DeviceVec.from_host(b1), K.DeviceVec.from_host(b2)
r1 = K.ec("g1", "to_affine", K.msm("g1", d_s, d_b1)); r2 = K.ec("g2", "to_affine", K.msm("g2", d_s, d_b2))
for _ in range(5): K.msm("g1", d_s, d_b1); K.msm("g2", d_s, d_b2)
m0, r0, t0, bad = free_mb(), rss(), time.time(), 0
for i in range(n):
    if i % 50 == 49:          # rewrite the bases through the API (same contents): the tables are retired and rebuilt
        d_b1.copy_from_host(b1); d_b2.copy_from_host(b2)
    bad += not np.array_equal(K.ec("g1", "to_affine", K.msm("g1", d_s, d_b1)), r1)
    bad += not np.array_equal(K.ec("g2", "to_affine", K.msm("g2", d_s, d_b2)), r2)
print(f"(2) automatic tables: {2 * n} MSMs in {time.time() - t0:.1f} s: {bad} differing results; free device memory {m0:.0f} -> {free_mb():.0f} MB; host max RSS {r0:.0f} -> {rss():.0f} MB")
d_s.free(); d_b1.free(); d_b2.free()

# (3) witness-light key
zk, wt, _, _ = bench.make_standin_inputs(K, S, "aadhaar_standin", scale=0.25)
cm.load("light", zk)
ref = cm.prove_mem("light", wt, 5, 7)[:2]
m0, r0, t0, bad = None, rss(), time.time(), 0
for i in range(n):
    bad += cm.prove_mem("light", wt, 5, 7, resident=bool(i % 2))[:2] != ref
    if i == 3: m0 = free_mb()        # after the one-time table rebuild
print(f"(3) witness-light key: {n} proves in {time.time() - t0:.1f} s: {bad} differing proofs; free device memory {m0:.0f} -> {free_mb():.0f} MB; host max RSS {r0:.0f} -> {rss():.0f} MB; b_bases {cm.info('light').b_bases}")
cm.close()
