"""cold prove: zkey FILE -> proof.json with nothing cached, then proves while the deferred tables are built, then warm ones.
usage: cold_prove.py [constraints]   (ICICLE_SNARK_TRACE_COLD=1 / ICICLE_SNARK_TRACE_HOST=1 for the breakdown)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
os.environ["ICICLE_SNARK_QUIET"] = "1"
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1600000
d = f"/tmp/isnark_cold_{N}"
os.makedirs(d, exist_ok=True)
zp, wp, pp, qp = (os.path.join(d, x) for x in ("circuit.zkey", "witness.wtns", "proof.json", "public.json"))
if not os.path.exists(zp):
    zkey, wtns = bench.make_inputs(K, S, N)
    open(zp, "wb").write(zkey); open(wp, "wb").write(wtns)
    if os.environ.get("PRE_READ") == "1":          # the files read once before the first prove (a key that has been used before)
        with open(zp, "rb") as f:
            while f.read(1 << 24):
                pass
    if os.environ.get("PRE_READ") == "2":          # … or just given time (write-back of the fresh file)
        import time as _t; os.sync(); _t.sleep(2)
    del zkey, wtns
K.check(K.lib().icicle_device_synchronize(), "sync")
for rnd in range(2):
    cm = K.CacheManager()   # (with a device current: the manager prewarms streams and staging buffers on a helper thread)
    time.sleep(0.3)         # a worker process waiting for its first command
    key = f"{zp}_HIP"
    t = time.perf_counter(); cm.prove_files(wp, zp, pp, qp); cold = (time.perf_counter() - t) * 1e3
    during = []
    t_all = time.perf_counter()
    while not cm.tables_ready(key):
        t = time.perf_counter(); cm.prove_files(wp, zp, pp, qp); during.append((time.perf_counter() - t) * 1e3)
    build = (time.perf_counter() - t_all) * 1e3
    warm = []
    for i in range(8):
        t = time.perf_counter(); cm.prove_files(wp, zp, pp, qp); warm.append((time.perf_counter() - t) * 1e3)
    print(f"round {rnd} ({'first key of the process' if rnd == 0 else 'process warm'}): cold prove {cold:.1f} ms | {len(during)} proves during the table build ({build:.0f} ms): "
          f"{' '.join(f'{x:.1f}' for x in during[:30])} | after: {' '.join(f'{x:.1f}' for x in warm)}", flush=True)
    assert json.loads(open(qp).read()) == [str(pow(3, 1 << N, S.R_MOD))]
    cm.evict(key); cm.close()
# tables built with nothing beside them
os.environ["ICICLE_SNARK_DEFER_TABLES"] = "0"
cm = K.CacheManager()
t = time.perf_counter(); cm.prove_files(wp, zp, pp, qp); print(f"ICICLE_SNARK_DEFER_TABLES=0: cold prove {(time.perf_counter() - t) * 1e3:.1f} ms")
