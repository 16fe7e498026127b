"""csrc/fr29.h — the lazy radix-2^29 scalar field of the NTT passes — against the 8×32-bit Fr of csrc/ff.h on the host (no GPU):
products with twiddles in Montgomery-261 form, un-reduced sums up to 704·r, shrink, borrow-proof subtraction against a large
subtrahend, the standard-form product of the fused epilogue, canonicalisation.  The GPU tests check the passes themselves
against the oracle (tests/test_gpu_ops.py::test_ntt_vs_oracle, ::test_ntt_extreme_inputs_on_the_lazy_field)."""
import os
import subprocess

from conftest import ROOT


def test_fr29_host_check(tmp_path):
    exe = tmp_path / "fr29_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "icicle-snark_amd", "csrc"), "-o", str(exe), os.path.join(ROOT, "tests", "fr29_check.cc")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout + out.stderr
