import sys, importlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
grp = sys.argv[1]
rng = np.random.default_rng(1)
for n in [int(a) for a in sys.argv[2:]]:
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 61) - 1)
    pts = K.generator_mul(grp, sc[::-1].copy())
    print("n", n, flush=True)
    r = K.msm(grp, K.DeviceVec.from_host(sc), K.DeviceVec.from_host(pts))
    print(" ok", r.ravel()[:2], flush=True)
