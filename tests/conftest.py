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


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import oraclelib
    oraclelib.lib()
    return oraclelib
