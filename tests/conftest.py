import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def debug_switch():
    """Sets development switches of the library (ifhip_debug_set) for one test and unsets them afterwards."""
    from imageflow_amd import _native
    used = []

    def set_(key, value):
        used.append(key)
        _native.debug_set(key, value)
    yield set_
    for k in used:
        _native.debug_set(k, None)
