"""The boundary is plain C: examples/line_fit.c is compiled with gcc against include/mhx.h and
libmhx.so and run (the reference's mcmc-fitting.lisp:1186 example from C)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path, name="line_fit"):
    exe = str(tmp_path / name)
    lib = os.path.join(ROOT, "lisp-mcmc_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-L", lib, "-lmhx",
                           "-Wl,-rpath," + lib, "-lm", "-o", exe])
    return exe


def test_c_example_compiles_against_the_header(tmp_path):
    """CPU: the header is valid C99 and the library resolves every symbol the example uses"""
    assert os.path.exists(build(tmp_path))
    assert os.path.exists(build(tmp_path, "walker_set"))


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    out = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "first step: prob -1821.547503103" in out.stdout


@pytest.mark.gpu
def test_c_walker_set_runs_through_the_group_entry_points(tmp_path):
    """examples/walker_set.c: mhx_group_* from plain C (one device here; `walker_set 8` on a
    node): 256 walkers of a two-peak fit, complete walker-adaptive-steps runs, parameters
    recovered"""
    out = subprocess.run([build(tmp_path, "walker_set"), "1", "256"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "1 device(s), 256 walkers" in out.stdout and "device 0 walks chains 0 .. 255" in out.stdout
