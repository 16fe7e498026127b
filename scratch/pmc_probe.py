import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("icicle-snark_amd")
K.set_device("HIP", 0)
out = (C.c_double * 5)()
K.check(K.lib().icicle_snark_pmc_probes(out), "probes")
print("KNOWN", *[int(x) for x in out])
