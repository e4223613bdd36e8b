import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The parity checker: reference headers + eigen_shim (oracle/_ref) when present, else the restatement."""
    import oracle_lib as O
    if O.available("ref", "dd"):
        return O.Oracle("ref", "dd")
    if O.available("port", "dd"):
        return O.Oracle("port", "dd")
    pytest.skip("no oracle library built (make -C oracle)")
