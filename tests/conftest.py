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


def pytest_sessionstart(session):
    """A fresh checkout has no build products (they are git-ignored): build the library and the oracle once, the way
    __graft_entry__.build() does.  hipcc cross-compiles for gfx950 without a GPU; nothing here falls back to another backend."""
    import subprocess
    lib = os.path.join(ROOT, "glaze_amd", "csrc", "libglaze_hip.so")
    if not os.path.exists(lib) and not os.environ.get("GLAZE_HIP_LIB"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "glaze_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


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
