"""Two proves in flight on ONE GPU (round-5 verdict item 5: measure, do not build).  benchmark/<N>:
  one     one thread, one manager: the headline loop (ms per prove)
  same    two host threads, same manager, same cached key — the existing entry (groth16_prove_mem holds the manager's mutex)
  twin    two host threads, two managers with the same zkey each: what a per-prove context (own streams, work buffers, bucket
          arrays) beside ONE set of key data would give — an upper bound measured with duplicated key data
wall / proves for each; proofs with fixed (r, s) must be equal everywhere."""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module("icicle-snark_amd"); S = importlib.import_module("icicle-snark_amd.synth")
K.set_device("HIP", 0)
N = int(os.environ.get("LOOP_CONSTRAINTS", "1600000"))
cache = f"/tmp/isnark_inputs_{N}"
if os.path.exists(cache + ".zkey"):
    zkey, wtns = open(cache + ".zkey", "rb").read(), open(cache + ".wtns", "rb").read()
else:
    zkey, wtns = bench.make_inputs(K, S, N)
    open(cache + ".zkey", "wb").write(zkey); open(cache + ".wtns", "wb").write(wtns)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cm1, cm2 = K.CacheManager(), K.CacheManager()
cm1.load("k", zkey); cm2.load("k", zkey)
ref = cm1.prove_mem("k", wtns, 3, 5)[0]
assert cm2.prove_mem("k", wtns, 3, 5)[0] == ref
for _ in range(3): cm1.prove_mem("k", wtns); cm2.prove_mem("k", wtns)

def loop(cm, n, out, fixed):
    K.set_device("HIP", 0)
    for i in range(n):
        pj = cm.prove_mem("k", wtns, 3, 5)[0] if fixed and i % 5 == 0 else cm.prove_mem("k", wtns)[0]
        if fixed and i % 5 == 0: out.append(pj == ref)

def run(cms, n):
    oks = []
    th = [threading.Thread(target=loop, args=(cm, n, oks, True)) for cm in cms]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = (time.perf_counter() - t) * 1e3
    assert all(oks) and oks, "a proof with fixed (r, s) differs"
    return dt / (n * len(cms))

for rnd in range(3):
    one = run([cm1], reps)
    same = run([cm1, cm1], reps)
    twin = run([cm1, cm2], reps)
    print(f"benchmark/{N}: ms per prove (wall / proves): one thread {one:.3f} | two threads, same key {same:.3f} | two threads, two managers {twin:.3f}")
