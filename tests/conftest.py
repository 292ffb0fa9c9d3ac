import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MATTEST = os.path.join(GOLDEN, "mattest.glaze")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    return pyoracle.lib()


@pytest.fixture(scope="session")
def instance():
    """RayTraceInstance on the GPU box.  GPU tests FAIL (not skip) when the HIP path is unavailable."""
    import glaze_amd
    inst = glaze_amd.RayTraceInstance.new()
    assert inst is not None, "no gfx950 device / libglaze_hip.so: the HIP path must be available for -m gpu tests"
    return inst
