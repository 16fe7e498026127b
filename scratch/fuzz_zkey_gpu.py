import base64, importlib, json, os, random, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
g = json.load(open("tests/golden/groth16.json"))
zkey, wtns = base64.b64decode(g["zkey"]), base64.b64decode(g["wtns"])
rng = random.Random(int(sys.argv[1]))
built = refused = 0
c = K.CacheManager()
for it in range(int(sys.argv[2])):
    b = bytearray(zkey)
    for _ in range(rng.randrange(1, 3)):
        k = rng.randrange(6)
        if k == 0: del b[rng.randrange(len(b)):]
        elif k == 1:
            for _ in range(rng.randrange(1, 6)): b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif k == 2:
            i = rng.randrange(0, max(1, len(b) - 8)); b[i:i + 8] = rng.choice([b"\xff" * 8, b"\0" * 8, (2 ** 63).to_bytes(8, "little"), (len(b) * 3).to_bytes(8, "little"), (len(b) // 2).to_bytes(8, "little")])
        elif k == 3:
            i = rng.randrange(len(b) + 1); b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 200)))
        elif k == 4:
            i = rng.randrange(0, max(1, len(b) - 4)); b[i:i + 4] = rng.choice([b"\xff\xff\xff\xff", b"\0\0\0\0", b"\x01\0\0\x80", b"\x09\0\0\0", b"\x10\0\0\0", b"\x07\0\0\0"])
        else:
            i, j = sorted((rng.randrange(len(b)), rng.randrange(len(b)))); b[i:j] = b[i:j][::-1]
        if len(b) < 16: break
    key = f"m{it}"
    try:
        c.load(key, bytes(b)); c.prove_mem(key, wtns, 3, 5); built += 1
    except K.ProverError:
        refused += 1
    c.evict(key)
print("built", built, "refused", refused)
c.load("ok", zkey)
from conftest import unhex_int
case = g["cases"][0]
pj, qj, _ = c.prove_mem("ok", wtns, unhex_int(case["r"]), unhex_int(case["s"]))
assert json.loads(pj) == case["proof"]
print("golden proof still right")
