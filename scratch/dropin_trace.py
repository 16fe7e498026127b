"""step timings of the drop-in sequence at benchmark/1600k: DROPIN_TRACE=1 lib/dropin_host … (stderr)"""
import importlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, int(sys.argv[1]) if len(sys.argv) > 1 else 1_600_000)
d = tempfile.mkdtemp()
open(d + "/c.zkey", "wb").write(zkey); open(d + "/w.wtns", "wb").write(wtns)
r = subprocess.run([ROOT + "/icicle-snark_amd/lib/dropin_host", d + "/c.zkey", d + "/w.wtns", d + "/p.json", d + "/q.json", "--iters", "5", "--keys-dir", d],
                   env=dict(os.environ, DROPIN_TRACE="1"), capture_output=True, text=True)
print(r.stdout[-600:]); print(r.stderr[-3000:])
