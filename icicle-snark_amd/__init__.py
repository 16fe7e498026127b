"""icicle-snark_amd — MI355X-native Groth16/BN254 prover hot path.

The product is the C-ABI shared library ``lib/libicicle_snark_hip.so`` (hand-written HIP kernels for
gfx950 + the C++ prover host).  This Python package is a thin ctypes mirror of that ABI used by the
tests and bench.py; it never computes anything itself and raises if the library is missing.
"""
from .binding import *  # noqa: F401,F403
