"""thread sweep of the CPU oracle on this host: G1 MSM of 2^20 random scalars and a 3 x 2^20 NTT per OpenMP thread count
(what cpu_baseline's calibration picks from; `nproc`, the cgroup CPU quota and the load average are printed with it)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, oracle as O
print("logical CPUs", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/proc/loadavg"):
    try: print(f, open(f).read().strip())
    except OSError: pass
rng = np.random.default_rng(0)
n = 1 << 20
sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)
G = O.ec_to_affine("g1", O.ec_generator("g1"))
O.set_num_threads(64)
pts = O.fixed_base_mul("g1", G, sc[::-1].copy())
x = np.concatenate([sc] * 3)
for T in (8, 16, 32, 64, 128, 256):
    if T > (os.cpu_count() or 1): break
    O.set_num_threads(T)
    t = time.time(); O.msm("g1", sc, pts); t1 = time.time() - t
    t = time.time(); O.fr_ntt(x, False, batch=3); t3 = time.time() - t
    print("threads %3d: G1 msm 2^20 %.3f s (%.0f scalars/s per thread)   ntt 3x2^20 %.3f s" % (T, t1, n / t1 / T, t3), flush=True)
