"""workload of the PMC passes (scratch/pmc_r05.sh): cache build + four proves from a host buffer (PMC_RESIDENT=1: the last three with the witness resident)"""
import importlib, os, sys
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
cm = K.CacheManager(); cm.load("k", zkey)
cm.prove_mem("k", wtns, 1, 1)
# PMC_RESIDENT=1: the witness stays in HBM (one witness sort per prove, as in round 4's table); default: every prove takes the witness
# from the host buffer like bench.py's own PMC child, so the witness is split into a head and a tail (three digit sorts per prove)
res = os.environ.get("PMC_RESIDENT", "0") != "0"
for _ in range(3):
    cm.prove_mem("k", wtns, 1, 1, resident=res)
cm.close()
