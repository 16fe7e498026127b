import sys, importlib, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = 1 << logn
K.release_domain(); K.initialize_domain(K.get_root_of_unity(2 * n))
rng = np.random.default_rng(0)
x = rng.integers(0, 1 << 62, size=(3 * n, 4), dtype=np.uint64)
d = K.DeviceVec.from_host(x)
st = K.IcicleStream()
for inverse in (True, False):
    for rep in range(3):
        K.check(K.lib().icicle_device_synchronize())
        t0 = time.perf_counter()
        for _ in range(5):
            K.ntt(d, inverse, batch_size=3, stream=st, is_async=True)
        st.synchronize()
        dt = (time.perf_counter() - t0) / 5
    print("ntt logn=%d batch=3 inverse=%s: %.3f ms" % (logn, inverse, dt * 1e3), flush=True)
