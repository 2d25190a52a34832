import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ora():
    import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def hip():
    """libngp_hip.so on a GPU box. Fails loudly (no CPU fallback) if the library or the device is missing."""
    import ngp_abi
    lib = ngp_abi.load_hip()
    assert lib.ngp_device_available() == 1, "no HIP device visible"
    return lib
