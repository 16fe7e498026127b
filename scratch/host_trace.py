"""host-side enqueue timeline of one prove (ICICLE_SNARK_TRACE_HOST=1): where the launching thread spends its time"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, N)
cm = K.CacheManager(); cm.load("k", zkey)
for i in range(4):
    print(f"--- prove {i}", file=sys.stderr)
    cm.prove_mem("k", wtns, resident=i > 0)
