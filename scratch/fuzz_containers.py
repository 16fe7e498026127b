"""mutation fuzzing of the container and JSON parsers on a host WITHOUT a GPU (every call must return an error code or a
verdict, never crash): flips / truncations / length-field edits of the golden zkey and wtns through groth16_cache_load and
groth16_commitments' parsers, and of proof / public / vkey JSON through groth16_verify_json.  Meant to be run against a
library built with -fsanitize=address,undefined (see HISTORY.md §6b)."""
import base64, importlib, json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
K = importlib.import_module("icicle-snark_amd")
g = json.load(open(os.path.join(ROOT, "tests", "golden", "groth16.json")))
zkey, wtns = base64.b64decode(g["zkey"]), base64.b64decode(g["wtns"])
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000


def mutate(b: bytes) -> bytes:
    b = bytearray(b)
    k = rng.randrange(6)
    if k == 0 and len(b) > 1:
        del b[rng.randrange(len(b)):]
    elif k == 1:
        for _ in range(rng.randrange(1, 8)):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
    elif k == 2:
        i = rng.randrange(0, max(1, len(b) - 8)); b[i:i + 8] = rng.choice([b"\xff" * 8, b"\0" * 8, (2 ** 63).to_bytes(8, "little"), (len(b) * 2).to_bytes(8, "little")])
    elif k == 3:
        i = rng.randrange(len(b)); b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 64)))
    elif k == 4:
        i = rng.randrange(0, max(1, len(b) - 4)); b[i:i + 4] = rng.choice([b"\xff\xff\xff\xff", b"\0\0\0\0", b"\x01\0\0\x80"])
    else:
        i, j = sorted((rng.randrange(len(b)), rng.randrange(len(b)))); b[i:j] = b[i:j][::-1]
    return bytes(b)


errs = 0
for it in range(n):
    cm = K.CacheManager()
    try:
        cm.load("k", mutate(zkey))
    except K.ProverError:
        errs += 1
    cm.close()
print("zkey mutations:", n, "rejected or stopped at the device:", errs)
case = g["cases"][0]
S = importlib.import_module("icicle-snark_amd.synth")
from conftest import unhex
v = g["vk"]
vk = dict(vk_alpha_1=unhex(v["vk_alpha_1"], 2, 4), vk_beta_2=unhex(v["vk_beta_2"], 4, 4), vk_gamma_2=unhex(v["vk_gamma_2"], 4, 4),
          vk_delta_2=unhex(v["vk_delta_2"], 4, 4), IC=[unhex(p, 2, 4) for p in v["IC"]], n_public=len(v["IC"]) - 1)
texts = [json.dumps(case["proof"]), json.dumps(case["public"]), S.vk_to_json(vk)]
assert K.groth16_verify_json(*texts)
accepted = 0
bad = 0
for it in range(n):
    t = [x for x in texts]
    i = rng.randrange(len(t))
    t[i] = mutate(t[i].encode()).decode("latin-1")
    try:
        ok = bool(K.groth16_verify_json(t[0], t[1], t[2]))
        accepted += ok
        if ok and os.environ.get("FUZZ_SHOW"):
            import difflib
            a0, b0 = texts[i], t[i]
            sm = difflib.SequenceMatcher(None, a0, b0)
            print("ACCEPTED text", i, [(tag, a0[i1 - 12:i2 + 4], b0[j1:j2][:40]) for tag, i1, i2, j1, j2 in sm.get_opcodes() if tag != "equal"][:3])
    except (K.ProverError, UnicodeError, ValueError):
        bad += 1
print("JSON mutations:", n, "format errors:", bad, "accepted (mutation did not change a value):", accepted)
