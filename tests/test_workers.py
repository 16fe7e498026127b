"""The prover's pool of persistent host worker threads (csrc/workers.h), compiled alone with g++ and driven on the CPU: tasks run once and
can be waited for, nested submissions work, exceptions are contained, and beyond the cap submit() refuses at once instead of blocking."""
import ctypes as C
import os
import subprocess

from conftest import ROOT


def test_worker_pool():
    out = os.path.join(ROOT, "build", "workers_check.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "icicle-snark_amd", "csrc"),
                    "-o", out, os.path.join(ROOT, "tests", "workers_check.cc")], check=True)
    assert C.CDLL(out).workers_check() == 0


def test_cold_feed_stages():
    """csrc/prover/cold_feed.h: the stage feed of the cold pipeline (uploader task -> prove thread), driven without a GPU"""
    out = os.path.join(ROOT, "build", "coldfeed_check.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-I" + os.path.join(ROOT, "icicle-snark_amd", "csrc"), "-o", out, os.path.join(ROOT, "tests", "coldfeed_check.cc")], check=True)
    assert C.CDLL(out).coldfeed_check() == 0
