"""groth16_prove_mem with the witness in PINNED host memory (hipHostMalloc) against a pageable buffer: same proof, ms per prove"""
import importlib, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
zkey, wtns = bench.make_inputs(K, S, 1_600_000)
cm = K.CacheManager(); cm.load("k", zkey)
hip = C.CDLL("libamdhip64.so")
p = C.c_void_p()
assert hip.hipHostMalloc(C.byref(p), C.c_size_t(len(wtns)), 0) == 0
C.memmove(p, wtns, len(wtns))
pinned = (C.c_char * len(wtns)).from_address(p.value)
ref = cm.prove_mem("k", wtns, 5, 7)[:2]
lib = K.lib()
pj, qj = C.create_string_buffer(1 << 14), C.create_string_buffer(1 << 20)
rb, sb = (5).to_bytes(32, "little"), (7).to_bytes(32, "little")
def prove_pinned():
    rc = lib.groth16_prove_mem(cm._h, b"k", pinned, C.c_size_t(len(wtns)), rb, sb, pj, C.c_size_t(len(pj)), qj, C.c_size_t(len(qj)), None)
    assert rc == 0, rc
    return pj.value.decode(), qj.value.decode()
assert prove_pinned() == ref
for name, f in (("pageable", lambda: cm.prove_mem("k", wtns, 5, 7)), ("pinned", prove_pinned), ("pageable", lambda: cm.prove_mem("k", wtns, 5, 7)), ("pinned", prove_pinned)):
    ts = []
    for _ in range(30):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    print(f"{name:9s} witness: median {ts[15]:.3f} ms, min {ts[0]:.3f} ms")
