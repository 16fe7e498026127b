"""Is the deferred table build paced by its host thread?  `load` + wait for the tables, alone and with H busy host processes beside it
(the build thread waits for an event every slice and enqueues three launches per slice: a descheduled thread lets the stream run dry)."""
import importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
cache = f"/tmp/isnark_inputs_{N}"
if os.path.exists(cache + ".zkey"):
    zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
else:
    zkey, wtns = bench.make_inputs(K, S, N)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
cm = K.CacheManager()
print("cpus visible", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for hogs in (0, 0, 16, 32, 64, 0):
    ps = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(hogs)]
    time.sleep(0.3)
    for rep in range(2):
        t0 = time.perf_counter()
        cm.load("k", zkey, wait_tables=False)
        usable = (time.perf_counter() - t0) * 1e3
        cm.tables_ready("k", wait=True)
        print(f"{hogs:3d} busy processes: usable after {usable:6.1f} ms, tables after another {(time.perf_counter() - t0) * 1e3 - usable:7.1f} ms", flush=True)
        cm.evict("k")
    for p in ps: p.kill()
    for p in ps: p.wait()
