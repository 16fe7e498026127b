"""QAP front end alone (spmv + NTTs) via a prove's timings with MSM excluded is not separable; this script times
only the spmv kernel through rocprof.  Run: rocprofv3 --kernel-trace --stats -- python3 scratch/spmv_only.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
import bench
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, 1_600_000)
cm = K.CacheManager(); cm.load("k", zkey)
for i in range(4):
    cm.prove_mem("k", wtns, 1, 1)
