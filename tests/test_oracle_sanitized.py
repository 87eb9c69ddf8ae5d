"""The checker under AddressSanitizer + UBSan: `make -C oracle asan` builds
oracle/libmhx_oracle_asan.so from the same source, and the oracle's own CPU test files run
against it in a child process with libasan preloaded.  A report (heap overflow, use after free,
signed overflow, misaligned access ...) aborts the child; leaks of the interpreter itself are
not looked for.  GPU sanitizers do not exist on the test pool; this covers the C the parity
claims rest on."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_oracle_tests_pass_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "libmhx_oracle_asan.so")
    env = dict(os.environ, MHX_ORACLE_LIBRARY=lib, LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    files = [os.path.join(HERE, f) for f in ("test_oracle_golden.py", "test_oracle_walker.py",
                                             "test_oracle_mirror.py", "test_device_math_cpu.py")]
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files,
                         capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
