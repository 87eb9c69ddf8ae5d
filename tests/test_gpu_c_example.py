"""The boundary is plain C: examples/line_fit.c is compiled with gcc against include/mhx.h and
libmhx.so and run (the reference's mcmc-fitting.lisp:1186 example from C)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "line_fit")
    lib = os.path.join(ROOT, "lisp-mcmc_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "line_fit.c"), "-L", lib, "-lmhx",
                           "-Wl,-rpath," + lib, "-lm", "-o", exe])
    return exe


def test_c_example_compiles_against_the_header(tmp_path):
    """CPU: the header is valid C99 and the library resolves every symbol the example uses"""
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    out = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "first step: prob -1821.547503103" in out.stdout
