import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver)")
    config.addinivalue_line("markers", "slow: tens of seconds of CPU (still part of the default run)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import oraclelib
    oraclelib.lib()
    return oraclelib


@pytest.fixture(autouse=True)
def _batch_kernels_unless_the_test_is_about_split_mode(request, monkeypatch):
    """Small batches on long datasets run in split mode by default (csrc/mhx_kernels.hpp), whose
    sums are grouped differently from the batch kernels'.  The parity suites were written about
    the batch kernels - several compare small engines with large ones bit for bit - so they pin
    MHX_SPLIT=0; tests/test_gpu_split.py manages the switch itself."""
    if request.module.__name__ != "test_gpu_split":
        monkeypatch.setenv("MHX_SPLIT", "0")
