"""is an is_async bn254_msm call asynchronous?  host time inside the call vs time until the stream is idle (device-resident operands)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
n = 1 << 21
rng = np.random.default_rng(1)
sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)
bases = K.generator_mul("g1", sc[::-1].copy())
st = K.IcicleStream()
d_s, d_b, d_r = K.DeviceVec.from_host(sc, st), K.DeviceVec.from_host(bases, st), K.DeviceVec(96, st)
st.synchronize()
for rep in range(6):
    t0 = time.perf_counter()
    K.msm("g1", d_s, d_b, out=d_r, stream=st, is_async=True)
    t1 = time.perf_counter()
    st.synchronize()
    t2 = time.perf_counter()
    print(f"call {rep}: inside the call {1e3 * (t1 - t0):.3f} ms, until the stream is idle {1e3 * (t2 - t0):.3f} ms")
