"""per-prove time and window geometry of a stand-in key over its first proves (witness-following digit width)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ICICLE_SNARK_QUIET"] = "1"
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "aadhaar_standin"
zkey, wtns, vk, nc = bench.make_standin_inputs(K, S, name)
cm = K.CacheManager()
if os.environ.get("COLD_FIRST", "0") != "0":
    import tempfile
    d = tempfile.mkdtemp()
    open(d + "/c.zkey", "wb").write(zkey); open(d + "/w.wtns", "wb").write(wtns)
    for rep in range(int(os.environ["COLD_FIRST"])):
        cm.prove_files(d + "/w.wtns", d + "/c.zkey", d + "/p.json", d + "/q.json")
        cm.tables_ready(d + "/c.zkey_HIP", wait=True)
        for _ in range(3):
            cm.prove_files(d + "/w.wtns", d + "/c.zkey", d + "/p.json", d + "/q.json")
        print("cold phase: A c =", K.msm_profile(4)[1]["c"], flush=True)
        cm.evict(d + "/c.zkey_HIP")
cm.load("k", zkey, wait_tables=(os.environ.get("WAIT_TABLES", "1") == "1"))
for i in range(int(os.environ.get("PROVES", "14"))):
    rdy = cm.tables_ready("k")
    t = time.perf_counter()
    cm.prove_mem("k", wtns, resident=(i > 0 and os.environ.get("RESIDENT", "0") == "1"))
    ms = (time.perf_counter() - t) * 1e3
    tm = cm.last_timings("k")
    g = K.msm_profile(4)[1]
    gh = K.msm_profile(0)[1]
    print(f"prove {i}: {ms:7.2f} ms  tables_ready={rdy}  qap {tm.qap_ms:.2f} msm {tm.msm_ms:.2f}  A: c={g['c']} W={g['W']} L={g['L']} nb={g['nbuckets']}  H: c={gh['c']} L={gh['L']}", flush=True)
    if i == 4:
        cm.tables_ready("k", wait=True)
