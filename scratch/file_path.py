"""groth16_prove on files (the reference's API: witness.wtns + circuit.zkey in, proof.json + public.json out), warm cache."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, 1_600_000)
d = "/tmp/isnark_files"; os.makedirs(d, exist_ok=True)
open(d + "/c.zkey", "wb").write(zkey); open(d + "/w.wtns", "wb").write(wtns)
cm = K.CacheManager()
for i in range(6):
    t = time.perf_counter()
    cm.prove(d + "/w.wtns", d + "/c.zkey", d + "/proof.json", d + "/public.json", "HIP")
    print("groth16_prove(files) #%d: %.2f ms" % (i, (time.perf_counter() - t) * 1e3), flush=True)
t = time.perf_counter(); K.groth16_verify(d + "/proof.json", d + "/public.json", d + "/vk.json") if os.path.exists(d + "/vk.json") else None
