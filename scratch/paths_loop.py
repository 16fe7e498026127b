"""N proves of a benchmark size through the three entry points in one process — files in / files out (the reference's timed
region), host buffer in, witness resident — median ms each + the phase times of the file path.  A/B helper: the library
reads its ICICLE_SNARK_* knobs once per process.   usage: paths_loop.py [proves]   (LOOP_CONSTRAINTS, default 1600000)"""
import importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
cache = f"/tmp/isnark_inputs_{N}"
if not os.path.exists(cache + ".zkey"):
    zkey, wtns = bench.make_inputs(K, S, N)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
zp, wp = cache + ".zkey", cache + ".wtns"
d = tempfile.mkdtemp()
key = zp + "_HIP"
cm = K.CacheManager(); cm.load(key, zkey)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for _ in range(3):
    cm.prove_files(wp, zp, d + "/proof.json", d + "/public.json")
ref = cm.prove_mem(key, wtns, 3, 4)[0]
assert cm.prove_mem(key, wtns, 3, 4, resident=True)[0] == ref
def med(f):
    ts = []
    q = m = h = 0.0
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
        tm = cm.last_timings(key); q += tm.qap_ms; m += tm.msm_ms; h += tm.h2d_ms
    ts.sort()
    return ts[n // 2], ts[0], q / n, m / n, h / n
f = med(lambda: cm.prove_files(wp, zp, d + "/proof.json", d + "/public.json"))
hst = med(lambda: cm.prove_mem(key, wtns))
r = med(lambda: cm.prove_mem(key, wtns, resident=True))
f2 = med(lambda: cm.prove_files(wp, zp, d + "/proof.json", d + "/public.json"))
print(f"N={N} files {f[0]:.3f}/{f2[0]:.3f} (min {min(f[1], f2[1]):.3f}; h2d {f[4]:.2f} qap {f[2]:.2f} msm {f[3]:.2f}) | host {hst[0]:.3f} (h2d {hst[4]:.2f} qap {hst[2]:.2f} msm {hst[3]:.2f}) | resident {r[0]:.3f} (qap {r[2]:.2f} msm {r[3]:.2f})")
