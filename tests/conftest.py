import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "droid-slam_reserch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def backends():
    """The product module; on the GPU box its HIP library must load (no fallback)."""
    import droid_backends
    droid_backends._lib.load()
    return droid_backends
