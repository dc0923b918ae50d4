import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """libcmps.so built in-tree (cross-compiles without a GPU)."""
    from audio_mps_amd import build
    build.build()
    from audio_mps_amd import _capi
    return _capi.load()
