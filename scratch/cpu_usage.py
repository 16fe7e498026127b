"""How many host CPUs does a prove loop keep busy?  (user + system CPU seconds of the process / wall seconds) for file-to-file, host-buffer
and resident proves of benchmark/1600k; the container's CPU quota (cgroup) is printed beside it."""
import importlib, os, resource, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = 1600000
cache = f"/tmp/isnark_inputs_{N}"
if os.path.exists(cache + ".zkey"):
    zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
else:
    zkey, wtns = bench.make_inputs(K, S, N)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
    try: print(f, open(f).read().replace("\n", " | "))
    except OSError: pass
tmp = tempfile.mkdtemp()
zp, wp, pp, qp = (os.path.join(tmp, n) for n in ("c.zkey", "w.wtns", "p.json", "q.json"))
open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
cm = K.CacheManager()
cm.prove_files(wp, zp, pp, qp)
key = f"{zp}_HIP"
cm.tables_ready(key, wait=True)
for _ in range(5): cm.prove_files(wp, zp, pp, qp)
def run(name, f, n=100):
    r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    print(f"{name}: {dt / n * 1e3:.3f} ms per prove, {cpu / dt:.2f} CPUs busy (user {(r1.ru_utime - r0.ru_utime) / dt:.2f} + sys {(r1.ru_stime - r0.ru_stime) / dt:.2f}), invol ctx switches {r1.ru_nivcsw - r0.ru_nivcsw}", flush=True)
run("files   ", lambda: cm.prove_files(wp, zp, pp, qp))
run("host buf", lambda: cm.prove_mem(key, wtns))
run("resident", lambda: cm.prove_mem(key, wtns, resident=True))
try: print("cpu.stat after:", open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " | "))
except OSError: pass
print("loadavg", os.getloadavg())
