import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-v3-rs_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (oracle/): the checker, never the thing under test."""
    from oracle_binding import oracle_binding
    return oracle_binding()


@pytest.fixture(scope="session")
def product():
    """libpbrt_hip.so through its C ABI.  No fallback: if it is not built the test errors."""
    import pbrt_hip
    return pbrt_hip.default_binding()


@pytest.fixture(scope="session")
def host(product):
    import pbrt_hip
    return pbrt_hip.Host(product)
