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
def family():
    from aprilslam_amd.families import get_family
    return get_family("tagStandard41h12")


@pytest.fixture(scope="session")
def gpu_detector():
    """One HIP detector for the whole GPU session (fails loudly if the library or GPU is missing)."""
    from aprilslam_amd import _lib
    det = _lib.Detector("tagStandard41h12", id_limit=0)
    yield det
    det.close()
