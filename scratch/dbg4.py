import sys, importlib, json
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import numpy as np
from conftest import load_golden, unhex
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
for grp, na in (("g1", 2), ("g2", 4)):
    for c in load_golden("msm.json")[grp]:
        n = c["n"]
        sc, bases = unhex(c["scalars"], n, 4), unhex(c["bases"], n, na, 4)
        want = unhex(c["result_affine"], na, 4)
        for cc in (5, 12, 16):
            bad = 0
            for rep in range(10):
                res = K.msm(grp, sc, bases, c=cc)
                if not np.array_equal(K.ec(grp, "to_affine", res), want): bad += 1
            # device resident variant
            d_s, d_b = K.DeviceVec.from_host(sc), K.DeviceVec.from_host(bases)
            bad_d = 0
            for rep in range(10):
                res = K.msm(grp, d_s, d_b, c=cc)
                if not np.array_equal(K.ec(grp, "to_affine", res), want): bad_d += 1
            d_s.free(); d_b.free()
            print(grp, n, c["kind"], "c=%d" % cc, "bad host-staged:", bad, "bad dev:", bad_d, flush=True)
