import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly if it is missing or no device is visible."""
    from cqs_amd import _lib
    lib = _lib.load()
    assert lib.cqs_hip_device_count() > 0, "no HIP device visible: GPU tests need an MI355X"
    return lib
