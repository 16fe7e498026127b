"""Table-mode bn254_g2_msm alone on the GPU (L points, default 1 600 002 = the B2 MSM of benchmark/1600k): classic call, table
build, then `reps` table hits; prints the stage times of the last hit (msm_profile).  Under rocprofv3 --kernel-trace this is the
solo per-kernel record of the G2 accumulation (classic walk: ICICLE_SNARK_G2_TREE=0; pairwise tree: default)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_600_002
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(5)
def rand_fr(n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a
bases = K.generator_mul("g2", rand_fr(L))
d_b = K.DeviceVec.from_host(bases)
d_s = K.DeviceVec.from_host(rand_fr(L))
outs = []
for i in range(2 + reps):
    t = time.perf_counter()
    outs.append(K.ec("g2", "to_affine", K.msm("g2", d_s, d_b)))
    dt = (time.perf_counter() - t) * 1e3
    ms, g = K.msm_profile(0)
    print(f"call {i}: {dt:8.2f} ms wall | sort {ms[0]:.3f} acc {ms[1]:.3f} rest {ms[2]:.3f} total {ms[3]:.3f} | c={g['c']} W={g['W']} nb={g['nbuckets']}")
assert all(np.array_equal(o, outs[0]) for o in outs), "results differ between calls"
print("all results equal")
