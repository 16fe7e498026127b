"""pytest configuration: markers, import paths and shared fixtures.

`-m "not gpu"`  : oracle vs golden vectors, oracle vs oracle/_ref (when built), host logic, ABI checks — CPU only.
`-m gpu`        : parity tests proper — the HIP path (through the C ABI) against the oracle.
"""
import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def R():
    import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference): make -C oracle ref")
    return ref


@pytest.fixture(scope="session")
def K():
    """the product package (ctypes mirror of the C ABI)"""
    return importlib.import_module("icicle-snark_amd")


@pytest.fixture(scope="session")
def S():
    return importlib.import_module("icicle-snark_amd.synth")


@pytest.fixture(scope="session")
def gpu(K):
    """selects HIP device 0 through the C ABI; fails loudly when there is no device"""
    K.set_device("HIP", 0)
    return K


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unhex(h, *shape):
    a = np.frombuffer(bytes.fromhex(h), dtype=np.uint64).copy()
    return a.reshape(*shape) if shape else a


def unhex_int(h):
    return int.from_bytes(bytes.fromhex(h), "little")
