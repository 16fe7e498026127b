import sys, importlib
sys.path.insert(0, ".")
import numpy as np
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
rng = np.random.default_rng(0)
sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 61) - 1)
pts = K.generator_mul("g1", sc[::-1].copy())
kind = rng.integers(0, 100, size=n)
sk = sc.copy()
sk[kind < 40] = 0
sk[(kind >= 40) & (kind < 75)] = np.array([1, 0, 0, 0], dtype=np.uint64)
sk[(kind >= 75) & (kind < 85), 1:] = 0      # 64-bit values
d_b = K.DeviceVec.from_host(pts)
for name, s in (("uniform", sc), ("witness-like (40% 0, 35% 1, 10% u64)", sk)):
    d_s = K.DeviceVec.from_host(s)
    for rep in range(2):
        res = K.msm("g1", d_s, d_b)
        ms, geom = K.msm_profile(0)
    print(name, "sort %.2f acc %.2f rest %.2f total %.2f ms" % tuple(ms), flush=True)
    d_s.free()
