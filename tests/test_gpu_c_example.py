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
    assert os.path.exists(build(tmp_path, "closure_fit"))


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


@pytest.mark.gpu
def test_c_closure_fit_gets_the_peak_kernel_below_the_abi(tmp_path):
    """examples/closure_fit.c: a closure's body as text from plain C, the plist in another order
    than the enumerated model's: mhx_expr_classify names GAUSS_PEAKS {2, 2}, the engine runs
    gauss22_normal (as written: rtc[expr...]), both give the same log-posterior to rounding, and
    4096 walkers recover the generating parameters"""
    out = subprocess.run([build(tmp_path, "closure_fit")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "model 1 shape {2, 2}" in out.stdout and "b0 b1 a1 mu1 w1 a2 mu2 w2" in out.stdout
    assert "gauss22_normal" in out.stdout and "rtc[expr" in out.stdout
