import sys, importlib
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
grp = sys.argv[1] if len(sys.argv) > 1 else "g1"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 21
n = 1 << logn
rng = np.random.default_rng(0)
sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 61) - 1)
if len(sys.argv) > 3 and sys.argv[3] == "modr":   # uniform mod r (what a prover sees) instead of uniform below 2^253
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    top = rng.integers(0, R >> 192, size=n, dtype=np.uint64)
    sc[:, 3] = top
pts = K.generator_mul(grp, sc[::-1].copy())
d_s, d_b = K.DeviceVec.from_host(sc), K.DeviceVec.from_host(pts)
for rep in range(3):
    res = K.msm(grp, d_s, d_b)
    ms, geom = K.msm_profile(0)
    print(grp, geom, "sort %.2f acc %.2f rest %.2f total %.2f ms" % tuple(ms), flush=True)
